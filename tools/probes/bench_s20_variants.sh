B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
f() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['config']['host_enqueue_ms_per_step'],4), [round(k['avg_us'],1) for k in d['roofline']['kernels']])"; }
for i in 1 2; do
$B 2>/dev/null | f base
RK_BENCH_SAMPLE=post $B 2>/dev/null | f post
RK_BENCH_PREWARM=1 $B 2>/dev/null | f prewarm
RK_BENCH_PREWARM=1 RK_BENCH_SAMPLE=post $B 2>/dev/null | f prewarm_post
RK_GRAPH=0 RK_BENCH_PREWARM=1 $B 2>/dev/null | f prewarm_eager
done
python bench.py --no-cpu-baseline 2>/dev/null | f default200
