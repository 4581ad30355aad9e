#!/bin/bash
# Same-box A/B of the one-tile ring kernel's epilogue (dword stores vs the LDS-transposed 16-byte form): two LAB libraries,
# alternating bench runs.  libmmdyn_hip_lab_old.so = the LAB objects with the committed igemm_ws.hip, libmmdyn_hip_lab.so = with
# prototypes/igemm_ws_vec_epilogue.patch applied (docs/LAB_NOTES.md E(f)).  usage: bash tests/microbench/run_ab_ws_epi.sh [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$R/multimodal-dynamics_amd/mmdyn_hip
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do
  echo -n "old epilogue: "; MMDYN_HIP_LIB=$L/libmmdyn_hip_lab_old.so python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
  echo -n "new epilogue: "; MMDYN_HIP_LIB=$L/libmmdyn_hip_lab.so python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
done
