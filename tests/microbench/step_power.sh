#!/bin/bash
# Shader clock and package power (rocm-smi) while the bench step replays, native fp32 arithmetic and fp32x3.
R=${GRAFT_REPO_ROOT:-/root/repo}
for dt in f32 f32x3; do
  python3 $R/bench.py --no-cpu-baseline --no-alt --dtype $dt --steps 1500 --warmup 10 > /tmp/step_power_$dt.json 2>/dev/null &
  pid=$!
  sleep 7
  for i in 1 2 3; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' ' | sed 's/  */ /g'; echo
    sleep 1
  done
  wait $pid
  python3 -c "import json; d=json.load(open('/tmp/step_power_$dt.json')); print('$dt: %.3f ms/step  %.0f samples/s' % (d['ms_per_step'], d['value']))"
done
