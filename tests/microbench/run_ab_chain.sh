#!/bin/bash
mkdir -p gpurun_out
run() { # name, env...
  local name=$1; shift
  for i in 1 2; do
    env "$@" timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab_${name}_f32_$i.json 2> gpurun_out/ab_${name}_f32_$i.err || exit 1
    env "$@" timeout -k 10 200 python bench.py --dtype bf16s --batch 128 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/ab_${name}_b16_$i.json 2> gpurun_out/ab_${name}_b16_$i.err || exit 1
  done
}
run all X=1
run noticket MMDYN_AB_NOTICKET=1
run oldsplitk MMDYN_AB_OLDSPLITK=1
run nodgradact MMDYN_AB_NODGRADACT=1
run none MMDYN_AB_NOTICKET=1 MMDYN_AB_OLDSPLITK=1 MMDYN_AB_NODGRADACT=1
run all2 X=1
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*_[12].json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value']), round(d['ms_per_step'],3))
PY
