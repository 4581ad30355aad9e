mkdir -p gpurun_out/r2
run() { tag=$1; shift; python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>gpurun_out/r2/bench_$tag.err | tee gpurun_out/r2/bench_$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$tag', round(d['value']), 'samples/s', round(d['ms_per_step'],3), 'ms', 'step TF/s', round(r['step_algorithmic_tflops'],1), 'frac', round(r['step_frac_of_peak'],3))"; }
run c1_f32
run c2_bf16s_b128 --dtype bf16s --batch 128
run c3_dyn128_b128 --image-size 128 --problem dyn_modeling --batch 128
run c3_dyn128_b128_bf16s --image-size 128 --problem dyn_modeling --batch 128 --dtype bf16s
run c4_256_fp16_b256 --image-size 256 --dtype fp16 --batch 256
run c4_256_fp16_b64 --image-size 256 --dtype fp16 --batch 64
run c4_256_f32_b64 --image-size 256 --batch 64
run c1_fp16 --dtype fp16
