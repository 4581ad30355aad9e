#!/bin/bash
# same-box A/B: the decoder row cut after k steps of the decoders' backward (MMDYN_DEC_SPLIT=k), the weight gradients queued by then on
# the main stream next to the rest of that backward
for r in 1 2 3; do for k in ${SPLITS:-0 2 3}; do MMDYN_DEC_SPLIT=$k python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $k:', round(d['ms_per_step'],3), 'ms', round(d['value']), d['config']['final_loss'])"; done; done
