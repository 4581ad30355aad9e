#!/bin/bash
# rocprofv3 kernel trace of the two-lane graph-replayed bench step -> overlap summary + per-kernel totals in $1
out=${1:-gpurun_out/trace}
shift
mkdir -p $GRAFT_REPO_ROOT/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt "$@" > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/err.txt
cd $GRAFT_REPO_ROOT
f=$(find $out/trace -name "*kernel_trace.csv" | head -1)
python tests/microbench/trace_overlap.py $f 6 > $out/overlap.txt
python tests/microbench/trace_kernels.py $f 6 60 > $out/kernels.txt
rm -rf $out/trace
head -3 $out/overlap.txt; head -30 $out/kernels.txt
