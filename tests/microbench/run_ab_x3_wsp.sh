#!/bin/bash
# Same-box A/B (LAB library): the launches the persistent ring kernel serves, in the fp32x3 arithmetic, on that kernel (its MFMA waves
# split their fragments in registers) against the register-staged split kernels (MMDYN_X3_WSP=0); native fp32 next to them.
R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do
  echo -n "native fp32:                                  "; python3 $R/bench.py --no-cpu-baseline --no-alt "$@" 2>/dev/null | ms
  echo -n "fp32x3, register-staged split kernels only:   "; MMDYN_X3_WSP=0 python3 $R/bench.py --no-cpu-baseline --dtype f32x3 "$@" 2>/dev/null | ms
  echo -n "fp32x3, persistent ring kernel where it serves: "; python3 $R/bench.py --no-cpu-baseline --dtype f32x3 "$@" 2>/dev/null | ms
done
