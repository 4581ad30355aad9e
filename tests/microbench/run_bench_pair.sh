#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
  timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_f32_$i.json 2> gpurun_out/bench_f32_$i.err || exit 1
  timeout -k 10 200 python bench.py --dtype bf16s --batch 128 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_b16_$i.json 2> gpurun_out/bench_b16_$i.err || exit 1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_f32_[12].json')+glob.glob('gpurun_out/bench_b16_[12].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))
PY
