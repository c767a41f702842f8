#!/bin/bash
# A/B of the 20-step run's start: --pretouch-reps x repetitions on ONE box
for rep in 1 2 3; do
 for v in 1 8 16 32 64; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --pretouch-reps $v 2>/dev/null | tail -1 | \
   python -c "import sys,json; d=json.loads(sys.stdin.read()); print('reps=$v', round(d['ms_per_step'],4))"
 done
done
python bench.py --no-cpu-baseline --no-recall 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200 steps', round(d['ms_per_step'],4))"
