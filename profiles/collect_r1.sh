#!/bin/bash
# Collects the measurement set kept under profiles/r1 (run on the GPU box through gpurun; outputs under gpurun_out/v5).
# usage: bash profiles/collect_r1.sh [part1|part2]
set -eo pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/v5
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
part=${1:-part1}
if [ "$part" = part1 ]; then
  python3 $R/bench.py --steps 30 --warmup 5 > $O/bench_n1_v5.json 2> $O/bench_n1_v5.err
  echo "bench done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_graph -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_n1_under_rocprof_v5.json 2> $O/rocprof_graph.err
  echo "rocprof graph done"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --single-lane --no-graph > $O/bench_n1_single_lane_under_rocprof_v5.json 2> $O/rocprof_single.err
  echo "rocprof single done"
  python3 $R/bench.py --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_n1_event_breakdown_v5.txt
  cp $(find $O/prof_graph -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats_v5.csv
  cp $(find $O/prof_single -name "*kernel_stats.csv" | head -1) $O/bench_n1_single_lane_kernel_stats_v5.csv
  rm -rf $O/prof_graph $O/prof_single
else
  for dt in bf16s bf16; do
    python3 $R/bench.py --dtype $dt --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_n1_${dt}_v5.json 2>> $O/part2.err
    python3 $R/bench.py --dtype $dt --batch 128 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_n1_${dt}_b128_v5.json 2>> $O/part2.err
    python3 $R/bench.py --dtype $dt --breakdown --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2> $O/bench_n1_${dt}_event_breakdown_v5.txt
  done
  echo "bf16 done"
  python3 $R/bench.py --infer --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_n1_infer_v5.json 2>> $O/part2.err
  python3 $R/bench.py --infer --dtype bf16s --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_n1_infer_bf16s_v5.json 2>> $O/part2.err
  echo "infer done"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
  python3 $R/tests/microbench/pmc_by_kernel.py $O/pmc_fetch $O/pmc_write 4 $O/hbm_traffic_by_kernel_v5.json > $O/hbm_traffic_by_kernel_v5.txt
  rm -rf $O/pmc_fetch $O/pmc_write
  python3 $R/tests/microbench/bench_conv3.py > $O/conv3_microbench.txt 2>&1
  python3 $R/tests/microbench/bench_decode.py > $O/frame_decode_microbench.txt 2>&1 || true
fi
echo "collect $part done"
