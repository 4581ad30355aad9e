run() { tag=$1; shift; env "$@" python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],3))"; }
run lds X=1
run big2048_prio MMDYN_D16=1 MMDYN_D16_ONLY_BIG=1 MMDYN_D16_MIN_TILES=2048 MMDYN_D16_PRIO=1
run big2048_b256_prio MMDYN_D16=1 MMDYN_D16_ONLY_BIG=1 MMDYN_D16_MIN_TILES=2048 MMDYN_D16_BLOCKS=256 MMDYN_D16_PRIO=1
run big1024_b256_prio MMDYN_D16=1 MMDYN_D16_ONLY_BIG=1 MMDYN_D16_MIN_TILES=1024 MMDYN_D16_BLOCKS=256 MMDYN_D16_PRIO=1
run big2048_b512_prio MMDYN_D16=1 MMDYN_D16_ONLY_BIG=1 MMDYN_D16_MIN_TILES=2048 MMDYN_D16_BLOCKS=512 MMDYN_D16_PRIO=1
run lds X=1
