#!/bin/bash
# same-box A/B: the last k entries of each decoder's deferred weight-gradient queue at the end of the lane's own encoder backward
# (MMDYN_WGRAD_TAIL=k) instead of behind the main stream's work
for r in 1 2 3; do for k in ${TAILS:-0 1 2}; do MMDYN_WGRAD_TAIL=$k python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tail $k:', round(d['ms_per_step'],3), 'ms', round(d['value']), d['config']['final_loss'])"; done; done
