#!/bin/bash
# fp16s (IEEE-half activation storage): kernel + model tests, then the bench lines next to bf16s on the same box
set -o pipefail
mkdir -p gpurun_out/fp16s
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "storage or packed_weights or mixed_output" > gpurun_out/fp16s/kernels.log 2>&1 || { tail -40 gpurun_out/fp16s/kernels.log; exit 1; }
tail -3 gpurun_out/fp16s/kernels.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "fp16_storage or (full_batch_properties and fp16s)" > gpurun_out/fp16s/model.log 2>&1 || { tail -60 gpurun_out/fp16s/model.log; exit 1; }
grep "fp16s size" gpurun_out/fp16s/model.log; tail -3 gpurun_out/fp16s/model.log
for cfg in "bf16s 128 64" "fp16s 128 64" "bf16s 256 256" "fp16s 256 256" "fp16 256 256"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --dtype $1 --batch $2 --image-size $3 --no-cpu-baseline > gpurun_out/fp16s/bench_$1_b$2_s$3.json 2> gpurun_out/fp16s/bench_$1_b$2_s$3.err || { tail -20 gpurun_out/fp16s/bench_$1_b$2_s$3.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/fp16s/bench_$1_b$2_s$3.json").read().strip().splitlines()[-1])
print("$1 b$2 s$3:", d["value"], d["unit"], d["ms_per_step"], "ms")
PY
done
