#!/bin/bash
# Same-box A/B of the persistent ring kernel (LAB library: MMDYN_WSP=0 switches it off) on the other BASELINE configs' shares.
R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
ab() {
  name=$1; shift
  for i in 1 2; do
    echo -n "$name  one tile per block: "; MMDYN_WSP=0 python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
    echo -n "$name  persistent (rule):  "; python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
  done
}
ab "bf16s bs128        " --dtype bf16s --batch 128
ab "bf16s bs256        " --dtype bf16s
ab "fp16s bs128        " --dtype fp16s --batch 128
ab "128px fp32 bs128   " --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5
ab "128px bf16s bs128  " --image-size 128 --problem dyn_modeling --batch 128 --dtype bf16s --steps 50 --warmup 5
ab "256px bf16s bs256  " --image-size 256 --dtype bf16s --batch 256 --steps 20 --warmup 5
ab "256px fp32 bs64    " --image-size 256 --batch 64 --steps 20 --warmup 5
