#!/bin/bash
# same-box step A/B: fp32x3 with the plane-ring kernel on operands that arrive split against the in-kernel split (round 4)
out=${1:-gpurun_out/ab_planes}
mkdir -p $out
for r in 1 2 3; do
  for v in planes noplanes; do
    flag=""; [ $v = noplanes ] && flag="--no-planes"
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt $flag 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms', round(d['value']), 'samples/s')"
  done
done | tee $out/step_ab.txt
