#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do
  for t in 0 1; do
    tk=""; [ $t = 1 ] && tk="MMDYN_TICKET=1"
    env $tk timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/abt_f32_t${t}_$i.json 2>/dev/null || exit 1
    env $tk timeout -k 10 200 python bench.py --dtype bf16s --batch 128 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/abt_b16_t${t}_$i.json 2>/dev/null || exit 1
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/abt_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3))
PY
