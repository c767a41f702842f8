#!/bin/bash
# A/B of the 20-step run (bench.py --no-tails switches them off: the last 4 of its 8 + 8 + 4 steps replayed as a captured tail graph instead
# of enqueued launch by launch) x repetitions on ONE box, and the 200-step figure beside them
for rep in 1 2 3 4; do
 for v in 1 0; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall $([ $v = 0 ] && echo --no-tails) 2>/dev/null | tail -1 | \
   python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph_tails=$v', round(d['ms_per_step'],4))"
 done
done
python bench.py --no-cpu-baseline --no-recall 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('200 steps', round(d['ms_per_step'],4))"
