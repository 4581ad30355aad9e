#!/bin/bash
# Same-box A/B of the fp32 arithmetic (native / three-term split) on the other fp32 configurations.
R=${GRAFT_REPO_ROOT:-/root/repo}
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
ab() {
  name=$1; shift
  for i in 1 2; do
    echo -n "$name  f32:   "; python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
    echo -n "$name  f32x3: "; python3 $R/bench.py --no-cpu-baseline --dtype f32x3 "$@" 2>/dev/null | ms
  done
}
ab "64px bs128          " --batch 128
ab "128px dyn bs128     " --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5
ab "256px bs64          " --image-size 256 --batch 64 --steps 20 --warmup 5
ab "64px inference bs256" --infer
