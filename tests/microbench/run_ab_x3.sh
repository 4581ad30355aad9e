#!/bin/bash
# Same-box A/B of the three-term-split fp32 kernels (LAB library, MMDYN_X3=1) on the bench line: alternating runs.
# usage: bash tests/microbench/run_ab_x3.sh [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s  loss %s' % (d['ms_per_step'], d['value'], d.get('final_loss', d.get('loss'))))"; }
for i in 1 2 3; do
  echo -n "native fp32 matrix cores:   "; python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
  echo -n "three-term split (X3=1):    "; MMDYN_X3=1 MMDYN_X3_WGRAD=1 python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
done
