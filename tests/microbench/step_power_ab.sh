#!/bin/bash
# Shader clock and package power (rocm-smi) while the DEFAULT bench step (fp32x3) replays, for two trees on the same box:
# usage: step_power_ab.sh <tree A> <tree B>   (e.g. the round-6 tree and a worktree of round 5's final commit)
for R in "$@"; do
  for rep in 1 2; do
    (cd $R && python3 bench.py --no-cpu-baseline --no-alt --steps 2000 --warmup 10 > /tmp/step_power_ab.json 2>/dev/null) &
    pid=$!
    sleep 8
    for i in 1 2 3; do
      /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' ' | sed 's/  */ /g'; echo
      sleep 1
    done
    wait $pid
    python3 -c "import json; d=json.load(open('/tmp/step_power_ab.json')); print('$R: %.3f ms/step  %.0f samples/s' % (d['ms_per_step'], d['value']))"
  done
done
