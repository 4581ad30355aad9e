#!/bin/bash
# Collects the measurement set kept under profiles/r4 (run on the GPU box through gpurun; outputs under gpurun_out/r4p).
# usage: bash profiles/collect_r4.sh [cache|part1|x3|others|alltraffic|traffic <key> <bench args>]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
part=${1:-part1}
if [ "$part" = cache ]; then
  # Where the operand fills of the fp32 step are served from: L2 hit rate and the average L1 -> L2 read
  # latency per launch shape.  Separate passes (TCC: four slots), --kernel-trace only, eager single-lane step.
  tag=${2:-f32}; shift 2 || true
  for pass in "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum"; do
    n=$(echo $pass | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_${tag}_$n -- python3 $R/bench.py --no-alt --steps 2 --warmup 1 --no-graph --single-lane --no-cpu-baseline "$@" > /dev/null 2> $O/pmc_${tag}_$n.err
    echo "pass $n done"
  done
  python3 $R/tests/microbench/pmc_cache_by_launch.py 4 $O/cache_by_launch_$tag.json $O/pmc_${tag}_* > $O/cache_by_launch_$tag.txt
  rm -rf $O/pmc_${tag}_*
  head -30 $O/cache_by_launch_$tag.txt
elif [ "$part" = part1 ]; then
  python3 $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  echo "bench done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python3 $R/bench.py --no-alt --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_under_rocprof.json 2> $O/rocprof_graph.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -- python3 $R/bench.py --no-alt --steps 10 --warmup 3 --no-cpu-baseline --single-lane --no-graph > $O/bench_n1_single_lane_under_rocprof.json 2> $O/rocprof_single.err
  python3 $R/bench.py --no-alt --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_n1_event_breakdown.txt
  cp $(find $O/prof_graph -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats.csv
  cp $(find $O/prof_single -name "*kernel_stats.csv" | head -1) $O/bench_n1_single_lane_kernel_stats.csv
  rm -rf $O/prof_graph $O/prof_single
elif [ "$part" = x3 ]; then
  # the "fp32x3" arithmetic (fp32 GEMMs on the bf16 matrix cores, exact three-term operand split): bench line, kernel stats, events
  python3 $R/bench.py --dtype f32x3 > $O/bench_x3.json 2> $O/bench_x3.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3_graph -- python3 $R/bench.py --dtype f32x3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_x3_under_rocprof.json 2> $O/rocprof_x3_graph.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_x3_single -- python3 $R/bench.py --dtype f32x3 --steps 10 --warmup 3 --no-cpu-baseline --single-lane --no-graph > $O/bench_x3_single_lane_under_rocprof.json 2> $O/rocprof_x3_single.err
  python3 $R/bench.py --dtype f32x3 --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_x3_event_breakdown.txt
  cp $(find $O/prof_x3_graph -name "*kernel_stats.csv" | head -1) $O/bench_x3_kernel_stats.csv
  cp $(find $O/prof_x3_single -name "*kernel_stats.csv" | head -1) $O/bench_x3_single_lane_kernel_stats.csv
  rm -rf $O/prof_x3_graph $O/prof_x3_single
  python3 $R/bench.py --dtype f32x3 --batch 128 --no-cpu-baseline > $O/bench_x3_b128.json 2>> $O/others.err
  python3 $R/bench.py --dtype f32x3 --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_x3_c3_dyn128_b128.json 2>> $O/others.err
  python3 $R/bench.py --dtype f32x3 --image-size 256 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_x3_c4_256_b64.json 2>> $O/others.err
  python3 $R/bench.py --dtype f32x3 --infer --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_x3_infer.json 2>> $O/others.err
elif [ "$part" = others ]; then
  python3 $R/bench.py --dtype bf16s --batch 128 --no-cpu-baseline > $O/bench_c2_bf16s_b128.json 2>> $O/others.err
  python3 $R/bench.py --dtype bf16s --no-cpu-baseline > $O/bench_c2_bf16s_b256.json 2>> $O/others.err
  python3 $R/bench.py --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c3_dyn128_b128.json 2>> $O/others.err
  python3 $R/bench.py --image-size 128 --problem dyn_modeling --batch 128 --dtype bf16s --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c3_dyn128_b128_bf16s.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --dtype fp16 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_fp16_b256.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --dtype bf16s --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_bf16s_b256.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --dtype fp16s --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_fp16s_b256.json 2>> $O/others.err
  python3 $R/bench.py --dtype fp16s --batch 128 --no-cpu-baseline > $O/bench_c2_fp16s_b128.json 2>> $O/others.err
  python3 $R/bench.py --image-size 256 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_f32_b64.json 2>> $O/others.err
  python3 $R/bench.py --dtype fp16 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_c1_fp16.json 2>> $O/others.err
  python3 $R/bench.py --infer --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_infer.json 2>> $O/others.err
  python3 $R/bench.py --dtype bf16s --batch 128 --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_c2_event_breakdown.txt
elif [ "$part" = alltraffic ]; then
  bash $R/profiles/collect_r4.sh traffic s64_f32_b256_seq_modeling
  bash $R/profiles/collect_r4.sh traffic s64_f32x3_b256_seq_modeling --dtype f32x3
  bash $R/profiles/collect_r4.sh traffic s64_bf16s_b128_seq_modeling --dtype bf16s --batch 128
  bash $R/profiles/collect_r4.sh traffic s128_f32_b128_dyn_modeling --image-size 128 --problem dyn_modeling --batch 128
  bash $R/profiles/collect_r4.sh traffic s256_bf16s_b256_seq_modeling --image-size 256 --dtype bf16s --batch 256
  bash $R/profiles/collect_r4.sh traffic s256_fp16s_b256_seq_modeling --image-size 256 --dtype fp16s --batch 256
  bash $R/profiles/collect_r4.sh traffic s256_fp16_b256_seq_modeling --image-size 256 --dtype fp16 --batch 256
elif [ "$part" = traffic ]; then
  # whole-step HBM-side traffic of one bench.py workload: collect_r4.sh traffic <key> <bench args...>
  # separate PMC passes, --kernel-trace only (MI355X_MICROARCH.md, HBM section)
  key=$2; shift 2
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$key -- python3 $R/bench.py --no-alt --steps 2 --warmup 1 --no-graph --no-cpu-baseline "$@" > /dev/null 2> $O/pmc_fetch_$key.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$key -- python3 $R/bench.py --no-alt --steps 2 --warmup 1 --no-graph --no-cpu-baseline "$@" > /dev/null 2> $O/pmc_write_$key.err
  python3 $R/tests/microbench/pmc_by_kernel.py $O/pmc_fetch_$key $O/pmc_write_$key 4 $O/hbm_traffic_by_kernel_$key.json > $O/hbm_traffic_by_kernel_$key.txt
  python3 $R/profiles/make_traffic_json.py $O/hbm_traffic_by_kernel_$key.json $O/traffic_$key.json $key "$@" > /dev/null
  rm -rf $O/pmc_fetch_$key $O/pmc_write_$key
  tail -1 $O/hbm_traffic_by_kernel_$key.txt
fi
echo "collect $part done"
