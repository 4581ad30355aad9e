#!/bin/bash
# A/B of the wave-specialised igemm kernels inside the step (same box, alternating runs; LAB build reads MMDYN_IGEMM_WS)
mkdir -p gpurun_out
set -o pipefail
export MMDYN_HIP_LIB=$PWD/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
for i in 1 2 3; do
  MMDYN_IGEMM_WS=0 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_ws0_$i.json 2> gpurun_out/bench_ws0_$i.err || exit 1
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_ws1_$i.json 2> gpurun_out/bench_ws1_$i.err || exit 1
done
timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --breakdown > gpurun_out/bench_ws1_bd.json 2> gpurun_out/bench_ws1_bd.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_ws*_[123].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))
PY
