#!/bin/bash
# Same-box A/B of the round-4 changes on the bench line (LAB library reads the switches per launch): alternating runs.
# usage: bash tests/microbench/run_ab_r4.sh [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do
  echo -n "all off (WSP=0, heads one by one):        "; MMDYN_WSP=0 python3 $R/bench.py --no-cpu-baseline --no-grouped-heads "$@" 2>/dev/null | ms
  echo -n "persistent kernel only:                    "; python3 $R/bench.py --no-cpu-baseline --no-grouped-heads "$@" 2>/dev/null | ms
  echo -n "persistent without the s1p0 layer:         "; MMDYN_WSP_S1P0=0 python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
  echo -n "grouped heads only (WSP=0):                "; MMDYN_WSP=0 python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
  echo -n "all on (product rule):                     "; python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
done
