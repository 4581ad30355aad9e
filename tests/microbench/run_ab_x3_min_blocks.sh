R=${GRAFT_REPO_ROOT:-/root/repo}
export MMDYN_HIP_LIB=$R/multimodal-dynamics_amd/mmdyn_hip/libmmdyn_hip_lab.so
ms() { python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.0f samples/s' % (d['ms_per_step'], d['value']))"; }
for cfg in "--batch 128" "--batch 256" "--image-size 128 --problem dyn_modeling --batch 128 --steps 50 --warmup 5"; do
 for i in 1 2; do
  for mb in 512 256 128 64; do
    echo -n "[$cfg] min blocks $mb: "; MMDYN_X3_MIN_BLOCKS=$mb python3 $R/bench.py --no-cpu-baseline --dtype f32x3 $cfg 2>/dev/null | ms
  done
 done
done
