#!/bin/bash
# GPU tests, then the bench line for fp32 bs 256 and bf16s bs 128 (product library)
mkdir -p gpurun_out
set -o pipefail
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; rc=$?
tail -4 gpurun_out/pytest_gpu.txt
[ $rc -ne 0 ] && { grep -E "Error|assert|FAILED" gpurun_out/pytest_gpu.txt | head -20; exit $rc; }
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_f32_$i.json 2> gpurun_out/bench_f32_$i.err || exit 1
  timeout -k 10 200 python bench.py --dtype bf16s --batch 128 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_b16_$i.json 2> gpurun_out/bench_b16_$i.err || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_f32_[12].json')+glob.glob('gpurun_out/bench_b16_[12].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))
PY
