#!/bin/bash
# C2 at B = 4000: decode tile variants of the pipelined family (rk_tune RK_TUNE_PG_TILE: 256 = 256 x 256, 1282 = 128 x 256, 128 = 128 x 128)
for v in 0 1282 128 0; do
  python bench.py --config c2b4k --steps 60 --warmup 16 --no-cpu-baseline --no-recall --tune 6=$v 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('pg_tile=$v', round(d['ms_per_step'],4), [(k['name'][3:], round(k['avg_us'],1)) for k in d['roofline']['kernels']])"
done
