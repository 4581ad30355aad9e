run() { tag=$1; shift; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', round(d['value']), round(d['ms_per_step'],3))"; }
run contig X=1
run rr MMDYN_IGEMM_XCD_RR=1
run contig X=1
run rr MMDYN_IGEMM_XCD_RR=1
python tests/microbench/bench_igemm.py 2>&1 | grep igemm | awk '{print $(NF-1)}' | tr '\n' ' '; echo
MMDYN_IGEMM_XCD_RR=1 python tests/microbench/bench_igemm.py 2>&1 | grep igemm | awk '{print $(NF-1)}' | tr '\n' ' '; echo
