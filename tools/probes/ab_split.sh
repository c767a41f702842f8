line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], [ (k['name'],round(k['avg_us'],1)) for k in d['roofline']['kernels']])"; }
B="python bench.py --steps 200 --warmup 24 --no-cpu-baseline --no-recall"
RK_ADAM_SPLIT=0 $B 2>/dev/null | line split0
for rb in 128 256 512 1024; do
RK_ADAM_SPLIT=1 RK_ADAM_REST_BLOCKS=$rb $B 2>/dev/null | line split1_rb$rb
done
RK_ADAM_SPLIT=0 $B 2>/dev/null | line split0
