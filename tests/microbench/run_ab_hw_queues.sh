#!/bin/bash
# same-box sweep of GPU_MAX_HW_QUEUES (ROCclr hardware queues per process; the step uses five streams) on the bench line
for r in 1 2; do for q in 4 2 3 5 6; do GPU_MAX_HW_QUEUES=$q python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queues $q', round(d['ms_per_step'],3), 'ms', round(d['value']))"; done; done
