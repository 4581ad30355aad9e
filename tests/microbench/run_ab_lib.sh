#!/bin/bash
# same-box step A/B of two builds of the library: the in-tree one against tests/microbench/_prev/libmmdyn_hip_prev.so (a build of
# an earlier revision, linked by hand; not tracked).  usage: run_ab_lib.sh [outdir] [extra bench args]
out=${1:-gpurun_out/ab_lib}
shift || true
mkdir -p $out
prev=$PWD/tests/microbench/_prev/libmmdyn_hip_prev.so
for r in 1 2 3; do
  for v in new prev; do
    lib=""; [ $v = prev ] && lib=$prev
    MMDYN_HIP_LIB=$lib python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms', round(d['value']), 'samples/s', 'loss', d['config']['final_loss'])"
  done
done | tee $out/step_ab.txt
