R=${GRAFT_REPO_ROOT:-/root/repo}
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do
  echo -n "product lib f32x3: "; python3 $R/bench.py --no-cpu-baseline --dtype f32x3 2>/dev/null | ms
  echo -n "lab lib f32x3:     "; MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so python3 $R/bench.py --no-cpu-baseline --dtype f32x3 2>/dev/null | ms
  echo -n "product lib f32:   "; python3 $R/bench.py --no-cpu-baseline 2>/dev/null | ms
done
