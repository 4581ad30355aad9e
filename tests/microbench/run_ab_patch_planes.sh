#!/bin/bash
# same-box step A/B (fp32x3): the 32-channel up-sampling layers on operands that arrive split (tconv_patch_kernel, P3) against the
# same kernel on fp32 operands.  usage: run_ab_patch_planes.sh [outdir] [extra bench args]
out=${1:-gpurun_out/ab_patch}
shift || true
mkdir -p $out
for r in 1 2 3; do
  for v in planes fp32; do
    flag=""; [ $v = fp32 ] && flag="--no-patch-planes"
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt $flag "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms', round(d['value']), 'samples/s', 'loss', d['config']['final_loss'])"
  done
done | tee $out/step_ab.txt
