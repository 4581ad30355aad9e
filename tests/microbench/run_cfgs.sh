run() { tag=$1; shift; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --dtype bf16s --batch 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],3))"; }
run base X=1
run splitk1 MMDYN_SPLITK16=0
run splitk2 MMDYN_SPLITK16=2
run splitk4 MMDYN_SPLITK16=4
run base X=1
run splitk1 MMDYN_SPLITK16=0
