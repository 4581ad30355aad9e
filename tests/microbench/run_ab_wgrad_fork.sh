#!/bin/bash
# same-box step A/B: the decoders' deferred weight-gradient graphs forked behind their own decoder lane ("dec") against forked next
# to the encoder backward ("enc", rounds 3-4)
out=${1:-gpurun_out/ab_fork}
mkdir -p $out
for r in 1 2 3; do
  for v in ${FORKS:-dec enc}; do
    MMDYN_WGRAD_FORK=$v python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms', round(d['value']), 'samples/s')"
  done
done | tee $out/step_ab.txt
