#!/bin/bash
out=gpurun_out/ab_x3ws
mkdir -p $out
export MMDYN_HIP_LIB=$PWD/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
for r in 1 2 3; do
  for v in 0 1; do
    MMDYN_X3_WS=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3_ws=$v', round(d['ms_per_step'],3), 'ms', round(d['value']), 'samples/s', 'loss', d['config']['final_loss'])"
  done
done | tee $out/step_ab.txt
