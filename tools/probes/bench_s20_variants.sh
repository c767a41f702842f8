B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
f() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['config']['host_enqueue_ms_per_step'],4), [round(k['avg_us'],1) for k in d['roofline']['kernels']])"; }
for i in 1 2; do
$B 2>/dev/null | f base
$B --sample timed 2>/dev/null | f sample_timed
$B --prewarm 1 2>/dev/null | f prewarm
$B --prewarm 1 --sample timed 2>/dev/null | f prewarm_sample_timed
RK_GRAPH=0 $B --prewarm 1 2>/dev/null | f prewarm_eager
done
python bench.py --no-cpu-baseline 2>/dev/null | f default200
