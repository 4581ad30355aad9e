#!/bin/bash
# Same-box A/B of the fp32 arithmetic on the bench line, PRODUCT library: native fp32 matrix cores (--dtype f32) against the
# three-term split on the bf16 matrix cores (--dtype f32x3); alternating runs.   usage: run_ab_x3.sh [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s  final loss %.6f' % (d['ms_per_step'], d['value'], d['config']['final_loss']))"; }
for i in 1 2 3; do
  echo -n "--dtype f32   (native fp32 matrix cores): "; python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | ms
  echo -n "--dtype f32x3 (three-term split):         "; python3 $R/bench.py --no-cpu-baseline --dtype f32x3 "$@" 2>/dev/null | ms
done
