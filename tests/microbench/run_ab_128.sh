#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "extended or dyn_modeling_128 or fp16 or igemm" > gpurun_out/pytest_ext.txt 2>&1; rc=$?
tail -3 gpurun_out/pytest_ext.txt
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/pytest_ext.txt | head; exit $rc; }
export MMDYN_HIP_LIB=$PWD/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
for i in 1 2; do
  for ws in 0 1; do
    MMDYN_IGEMM_WS=$ws timeout -k 10 200 python bench.py --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/ab128_ws${ws}_$i.json 2> gpurun_out/ab128_ws${ws}_$i.err || exit 1
    MMDYN_IGEMM_WS=$ws timeout -k 10 200 python bench.py --image-size 256 --batch 64 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/ab256_ws${ws}_$i.json 2> gpurun_out/ab256_ws${ws}_$i.err || exit 1
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab128_ws*.json')+glob.glob('gpurun_out/ab256_ws*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3))
PY
