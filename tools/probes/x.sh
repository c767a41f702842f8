R=$GRAFT_REPO_ROOT
cd $R
run() { python bench.py $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', round(d['ms_per_step'],4), [(k['name'], round(k['avg_us'],1)) for k in d['roofline']['kernels']])"; }
for i in 1 2 3; do
run base
RK_HACK_SUM=1 run sum
done
