#!/bin/bash
# same-box A/B: the deferred weight-gradient queues in front of (1) or behind (0, default) the main stream's own work of the encoder-backward row
for r in 1 2 3; do for v in 0 1; do MMDYN_WGRAD_FIRST=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad first $v:', round(d['ms_per_step'],3), 'ms', round(d['value']))"; done; done
