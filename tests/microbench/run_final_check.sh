#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; rc=$?
tail -2 gpurun_out/pytest_gpu.txt
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/pytest_gpu.txt | head; exit $rc; }
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/fin_f32_$i.json 2>/dev/null || exit 1
  timeout -k 10 200 python bench.py --dtype bf16s --batch 128 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/fin_b16_$i.json 2>/dev/null || exit 1
  timeout -k 10 200 python bench.py --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/fin_c3_$i.json 2>/dev/null || exit 1
done
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown > /dev/null 2> gpurun_out/fin_bd.txt
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/fin_*_[12].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))
PY
grep -E "tconv_patch" gpurun_out/fin_bd.txt
