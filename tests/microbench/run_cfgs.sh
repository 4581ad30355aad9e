mkdir -p gpurun_out/r2
for cfg in "X=1" "MMDYN_IGEMM_M16=1" "MMDYN_IGEMM_TILE=128,128 MMDYN_IGEMM_M16_128=1 MMDYN_IGEMM_M16=1" "MMDYN_IGEMM_TILE=128,128 MMDYN_IGEMM_M32=1" "MMDYN_IGEMM_TILE=128,64 MMDYN_IGEMM_M16=1" "MMDYN_IGEMM_TILE=128,64 MMDYN_IGEMM_M32=1" "MMDYN_IGEMM_TILE=64,128 MMDYN_IGEMM_M16=1"; do
  echo "== $cfg"
  env $cfg python tests/microbench/bench_igemm.py 2>&1 | grep igemm | awk '{print $(NF-1)}' | tr '\n' ' '
  echo
done
