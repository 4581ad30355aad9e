#!/bin/bash
# same-box A/B: deferred decoder weight gradients in the 16-bit storage modes (off = default there)
for r in 1 2; do for v in off on; do python bench.py --dtype bf16s --batch 128 --defer-wgrad $v --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16s b128 defer $v', round(d['ms_per_step'],3), 'ms', round(d['value']))"; done; done
for r in 1 2; do for v in off on; do python bench.py --dtype bf16s --image-size 256 --batch 256 --defer-wgrad $v --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16s 256px defer $v', round(d['ms_per_step'],3), 'ms', round(d['value']))"; done; done
