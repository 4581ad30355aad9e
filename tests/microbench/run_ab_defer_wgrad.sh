#!/bin/bash
# Step-level A/B on one box: decoder weight gradients on two extra streams (--defer-wgrad on; the fp32 default), in-chain (off),
# (a third variant, the encoders' weight gradients deferred too into a last four-stream row, measured 6.79 ms against 6.66 / 6.77 for
# on / off and was removed)
mkdir -p gpurun_out/dw
rm -f gpurun_out/dw/*.json
set -o pipefail
run() {  # name, args...
  name=$1; shift
  timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/dw/$name.json 2> gpurun_out/dw/$name.err || { tail -8 gpurun_out/dw/$name.err; exit 1; }
}
for i in 1 2 3; do
  for v in off on; do
    run f32_${v}_$i --steps 100 --warmup 10 --defer-wgrad $v
  done
done
for v in off on; do
  run dyn128_${v} --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5 --defer-wgrad $v
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/dw/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d['config']['final_loss'])
PY
