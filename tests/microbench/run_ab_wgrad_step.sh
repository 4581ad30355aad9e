#!/bin/bash
# Step-level A/B on one box (LAB build): 128x64 weight-gradient tiles on / off, and the partial-slab target (blocks in flight)
mkdir -p gpurun_out/abw
set -o pipefail
export MMDYN_HIP_LIB=$PWD/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/abw/$name.json 2> gpurun_out/abw/$name.err || { tail -5 gpurun_out/abw/$name.err; exit 1; }
}
for i in 1 2 3; do
  run t0_$i MMDYN_WGRAD_128x64=0
  run t1_$i MMDYN_WGRAD_128x64=1
  run t1_b1024_$i MMDYN_WGRAD_128x64=1 MMDYN_WGRAD_BLOCKS=1024
  run t1_b1536_$i MMDYN_WGRAD_128x64=1 MMDYN_WGRAD_BLOCKS=1536
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/abw/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],3))
PY
