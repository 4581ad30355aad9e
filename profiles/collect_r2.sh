#!/bin/bash
# Collects the measurement set kept under profiles/r2 (run on the GPU box through gpurun; outputs under gpurun_out/r2p).
# usage: bash profiles/collect_r2.sh [part1|part2|part3]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2p
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
part=${1:-part1}
if [ "$part" = part1 ]; then
  python3 $R/bench.py --steps 30 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
  echo "bench done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_under_rocprof.json 2> $O/rocprof_graph.err
  echo "rocprof graph done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --single-lane --no-graph > $O/bench_n1_single_lane_under_rocprof.json 2> $O/rocprof_single.err
  echo "rocprof single done"
  python3 $R/bench.py --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_n1_event_breakdown.txt
  cp $(find $O/prof_graph -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats.csv
  cp $(find $O/prof_single -name "*kernel_stats.csv" | head -1) $O/bench_n1_single_lane_kernel_stats.csv
  rm -rf $O/prof_graph $O/prof_single
elif [ "$part" = part2 ]; then
  # HBM-side traffic per kernel: separate PMC passes, --kernel-trace only (MI355X_MICROARCH.md, HBM section)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
  python3 $R/tests/microbench/pmc_by_kernel.py $O/pmc_fetch $O/pmc_write 4 $O/hbm_traffic_by_kernel.json > $O/hbm_traffic_by_kernel.txt
  python3 $R/profiles/make_traffic_json.py $O/hbm_traffic_by_kernel.json $O/igemm_traffic_pmc.json
  rm -rf $O/pmc_fetch $O/pmc_write
else
  # the other BASELINE configs (same JSON contract; config.workload names each)
  python3 $R/bench.py --dtype bf16s --batch 128 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c2_bf16s_b128.json 2>> $O/part3.err
  python3 $R/bench.py --dtype bf16s --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c2_bf16s_b256.json 2>> $O/part3.err
  python3 $R/bench.py --image-size 128 --problem dyn_modeling --batch 128 --steps 30 --warmup 5 > $O/bench_c3_dyn128_b128.json 2>> $O/part3.err
  python3 $R/bench.py --image-size 128 --problem dyn_modeling --batch 128 --dtype bf16s --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c3_dyn128_b128_bf16s.json 2>> $O/part3.err
  python3 $R/bench.py --image-size 256 --dtype fp16 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_fp16_b256.json 2>> $O/part3.err
  python3 $R/bench.py --image-size 256 --batch 64 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_f32_b64.json 2>> $O/part3.err
  python3 $R/bench.py --image-size 256 --dtype bf16s --batch 256 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_256_bf16s_b256.json 2>> $O/part3.err
  python3 $R/bench.py --dtype fp16 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_c1_fp16.json 2>> $O/part3.err
  python3 $R/bench.py --infer --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_infer.json 2>> $O/part3.err
  python3 $R/bench.py --image-size 128 --problem dyn_modeling --batch 128 --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_c3_event_breakdown.txt
  python3 $R/bench.py --image-size 256 --dtype fp16 --batch 64 --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_c4_event_breakdown.txt
fi
echo "collect $part done"
