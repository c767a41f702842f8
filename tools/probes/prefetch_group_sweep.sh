for g in 1 4 16 32; do
  echo "G=$g"
  RK_PREFETCH_GROUP=$g python bench.py --steps 640 --warmup 32 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'])"
done
