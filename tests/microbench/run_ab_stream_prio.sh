#!/bin/bash
# same-box A/B: stream priorities of the two lanes (MMDYN_LANE_PRIO) and of the deferred weight-gradient streams (MMDYN_WGRAD_PRIO);
# -1 = high, 0 = default
for r in 1 2; do for v in "0 0" "-1 0" "0 -1" "-1 -1"; do set -- $v; MMDYN_LANE_PRIO=$1 MMDYN_WGRAD_PRIO=$2 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes $1 wgrad $2:', round(d['ms_per_step'],3), 'ms', round(d['value']))"; done; done
