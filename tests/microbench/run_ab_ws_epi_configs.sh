#!/bin/bash
# Same-box A/B of the one-tile ring kernel's epilogue on the other BASELINE configs (see run_ab_ws_epi.sh).
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$R/multimodal-dynamics_amd/mmdyn_hip
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
ab() {
  name=$1; shift
  for i in 1 2; do
    echo -n "$name  old epilogue: "; MMDYN_HIP_LIB=$L/libmmdyn_hip_lab_old.so python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
    echo -n "$name  new epilogue: "; MMDYN_HIP_LIB=$L/libmmdyn_hip_lab.so python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
  done
}
ab "bf16s bs128        " --dtype bf16s --batch 128
ab "bf16s bs256        " --dtype bf16s
ab "128px fp32 bs128   " --image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5
ab "256px bf16s bs256  " --image-size 256 --dtype bf16s --batch 256 --steps 20 --warmup 5
