#!/bin/bash
# same-box A/B: the pose encoder's backward on the main stream (default) or at the end of a lane's encoder-backward graph
for r in 1 2 3; do for v in main 1 0; do e=""; [ $v != main ] && e=$v; MMDYN_POSE_BWD_LANE=$e python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pose bwd on $v:', round(d['ms_per_step'],3), 'ms', round(d['value']), d['config']['final_loss'])"; done; done
