# steps per replayed graph (RK_GRAPH_GROUP) against the default bench and the driver's 20-step run
f() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d['config']['host_enqueue_ms_per_step'],4), [round(k['avg_us'],1) for k in d['roofline']['kernels']])"; }
for g in ${GROUPS_:-4 8 12 16}; do
RK_GRAPH_GROUP=$g python bench.py --no-cpu-baseline 2>/dev/null | f "G=$g k200"
RK_GRAPH_GROUP=$g python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | f "G=$g k20 "
done
