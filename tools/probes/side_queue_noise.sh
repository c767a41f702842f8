# what a second busy HIP queue costs the training chain (bench --diag-reuse-block = no collation)
for n in 0 1 3 5 10; do
  echo "side kernels per step: $n"
  python bench.py --steps 640 --warmup 32 --no-cpu-baseline --diag-reuse-block --diag-side-noise $n 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['host_enqueue_ms_per_step'])"
done
