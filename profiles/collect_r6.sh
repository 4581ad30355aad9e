#!/bin/bash
# Collects the measurement set kept under profiles/r6 (run on the GPU box through gpurun; outputs under gpurun_out/r6p).
# usage: bash profiles/collect_r6.sh [part1|native|others|alltraffic|othertraffic|traffic <key> <bench args>]
# The default arithmetic of bench.py is f32x3 (fp32 storage / results, GEMMs on the bf16 matrix cores through the exact three-term
# split, operands arriving split where the plane kernels serve the launch); --dtype f32 = the native fp32 matrix cores.
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6p
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
part=${1:-part1}
if [ "$part" = part1 ]; then
  python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  echo "bench done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python3 $R/bench.py --no-alt --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_under_rocprof.json 2> $O/rocprof_graph.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -- python3 $R/bench.py --no-alt --steps 10 --warmup 3 --no-cpu-baseline --single-lane --no-graph > $O/bench_n1_single_lane_under_rocprof.json 2> $O/rocprof_single.err
  python3 $R/bench.py --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_n1_event_breakdown.txt
  cp $(find $O/prof_graph -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats.csv
  cp $(find $O/prof_single -name "*kernel_stats.csv" | head -1) $O/bench_n1_single_lane_kernel_stats.csv
  f=$(find $O/prof_graph -name "*kernel_trace.csv" | head -1)
  python3 $R/tests/microbench/trace_overlap.py $f 3 > $O/two_lane_trace_overlap.txt || true
  python3 $R/tests/microbench/trace_kernels.py $f 3 60 > $O/two_lane_trace_kernels.txt || true
  rm -rf $O/prof_graph $O/prof_single
elif [ "$part" = native ]; then
  # the native fp32 matrix cores on the same workload: kernel stats of the single-lane step
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32_single -- python3 $R/bench.py --dtype f32 --no-alt --steps 10 --warmup 3 --no-cpu-baseline --single-lane --no-graph > $O/bench_f32_single_lane_under_rocprof.json 2> $O/rocprof_f32_single.err
  cp $(find $O/prof_f32_single -name "*kernel_stats.csv" | head -1) $O/bench_f32_single_lane_kernel_stats.csv
  rm -rf $O/prof_f32_single
  python3 $R/bench.py --dtype f32 --no-alt --no-cpu-baseline > $O/bench_f32.json 2>> $O/others.err
elif [ "$part" = others ]; then
  python3 $R/bench.py --batch 128 --no-cpu-baseline > $O/bench_x3_b128.json 2>> $O/others.err
  python3 $R/bench.py --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c3_dyn128_b128_x3.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_b64_x3.json 2>> $O/others.err
  python3 $R/bench.py --dtype bf16s --batch 128 --no-cpu-baseline > $O/bench_c2_bf16s_b128.json 2>> $O/others.err
  python3 $R/bench.py --dtype bf16s --no-cpu-baseline > $O/bench_c2_bf16s_b256.json 2>> $O/others.err
  python3 $R/bench.py --image-size 128 --problem dyn_modeling --batch 128 --dtype bf16s --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c3_dyn128_b128_bf16s.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --dtype fp16 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_fp16_b256.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --dtype bf16s --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_bf16s_b256.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --dtype fp16s --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_fp16s_b256.json 2>> $O/others.err
  python3 $R/bench.py --infer --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_infer_x3.json 2>> $O/others.err
elif [ "$part" = alltraffic ]; then
  bash $R/profiles/collect_r6.sh traffic s64_f32x3_b256_seq_modeling
  bash $R/profiles/collect_r6.sh traffic s64_f32_b256_seq_modeling --dtype f32
elif [ "$part" = othertraffic ]; then
  # whole-step PMC traffic of the other workloads of BASELINE.md (so that none of their lines says step_bound: "unknown")
  bash $R/profiles/collect_r6.sh traffic s64_bf16s_b128_seq_modeling --dtype bf16s --batch 128
  bash $R/profiles/collect_r6.sh traffic s128_f32x3_b128_dyn_modeling --image-size 128 --problem dyn_modeling --batch 128
  bash $R/profiles/collect_r6.sh traffic s256_bf16s_b256_seq_modeling --image-size 256 --dtype bf16s --batch 256
  bash $R/profiles/collect_r6.sh traffic s256_fp16s_b256_seq_modeling --image-size 256 --dtype fp16s --batch 256
elif [ "$part" = traffic ]; then
  # whole-step HBM-side traffic of one bench.py workload: collect_r6.sh traffic <key> <bench args...>
  # separate PMC passes, --kernel-trace only (MI355X_MICROARCH.md, HBM section)
  key=$2; shift 2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$key -- python3 $R/bench.py --no-alt --steps 2 --warmup 1 --no-graph --no-cpu-baseline "$@" > /dev/null 2> $O/pmc_fetch_$key.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$key -- python3 $R/bench.py --no-alt --steps 2 --warmup 1 --no-graph --no-cpu-baseline "$@" > /dev/null 2> $O/pmc_write_$key.err
  python3 $R/tests/microbench/pmc_by_kernel.py $O/pmc_fetch_$key $O/pmc_write_$key 0 $O/hbm_traffic_by_kernel_$key.json > $O/hbm_traffic_by_kernel_$key.txt
  python3 $R/profiles/make_traffic_json.py $O/hbm_traffic_by_kernel_$key.json $O/traffic_$key.json $key "$@" > /dev/null
  rm -rf $O/pmc_fetch_$key $O/pmc_write_$key
  tail -1 $O/hbm_traffic_by_kernel_$key.txt
fi
echo "collect $part done"
