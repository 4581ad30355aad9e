#!/bin/bash
# Same-box A/B of the three-term split's launch rule on the bench line (LAB library: the rule's switches are read per launch).
R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
run() { name=$1; shift; echo -n "$name "; env "$@" python3 $R/bench.py --no-cpu-baseline --dtype f32x3 2>/dev/null | ms; }
for i in 1 2; do
  run "rule (all modes, >= 512 blocks)          " A=1
  run "without the k4 s1 p0 layer               " MMDYN_X3_MODES=7
  run "convolutions only (no dense)             " MMDYN_X3_MODES=22
  run ">= 256 blocks                            " MMDYN_X3_MIN_BLOCKS=256
  run ">= 1024 blocks                           " MMDYN_X3_MIN_BLOCKS=1024
  run "igemm split only (wgrad native)          " MMDYN_X3_WGRAD=0
  run "wgrad split only (igemm native)          " MMDYN_X3=0
done
