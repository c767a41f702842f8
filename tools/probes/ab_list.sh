# A/B of the lazy sweeps' need lists (rk_lazy_need_lists; RK_ADAM_LAZY=16,list / 16,scan / 16 = by shape), C2 and C3, one box
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],5), [round(k['avg_us'],1) for k in d['roofline']['kernels']][:6])"; }
for rep in 1 2; do for l in 16,scan 16,list 16; do
RK_ADAM_LAZY=$l python bench.py --steps 200 --warmup 24 --no-cpu-baseline --no-recall 2>/dev/null | line "c2 $l"
RK_ADAM_LAZY=$l python bench.py --config c3 --no-cpu-baseline --no-recall 2>/dev/null | line "c3 $l"
done; done
