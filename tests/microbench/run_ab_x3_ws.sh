R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2 3; do
  echo -n "fp32x3 (persistent ring + register-staged split kernels):      "; python3 $R/bench.py --no-cpu-baseline --dtype f32x3 "$@" 2>/dev/null | ms
  echo -n "fp32x3, one-tile ring kernels in the split arithmetic as well:  "; MMDYN_X3_WS=1 python3 $R/bench.py --no-cpu-baseline --dtype f32x3 "$@" 2>/dev/null | ms
done
