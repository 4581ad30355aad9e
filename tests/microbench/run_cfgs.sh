mkdir -p gpurun_out/r2
run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],3))"; }
run m32 MMDYN_IGEMM_M32=1
run m16 X=1
run m32 MMDYN_IGEMM_M32=1
run m16 X=1
