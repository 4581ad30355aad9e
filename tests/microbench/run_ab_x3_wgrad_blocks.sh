R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
for i in 1 2; do
  for wb in 768 512 1024 384 1536; do
    echo -n "fp32x3, weight-gradient blocks in flight $wb: "; MMDYN_WGRAD_BLOCKS=$wb python3 $R/bench.py --no-cpu-baseline --dtype f32x3 2>/dev/null | ms
  done
done
