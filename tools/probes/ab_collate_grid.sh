# A/B of the look-ahead collation's grid cap (RK_COLLATE_GRID: workgroups per block of the two row-parallel launches), C2, one box
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],5))"; }
for rep in 1 2; do
for g in ${GRIDS:-0 16 32 64}; do
RK_COLLATE_GRID=$g python bench.py --steps 200 --warmup 24 --no-cpu-baseline --no-recall 2>/dev/null | line "grid$g steps200"
RK_COLLATE_GRID=$g python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall 2>/dev/null | line "grid$g steps20"
done; done
