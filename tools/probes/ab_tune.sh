#!/bin/bash
# usage: tools/probes/ab_tune.sh "<bench args A>" "<bench args B>" [reps] -- the default bench with two flag sets, alternating, on ONE box
A="$1"; B="$2"; reps=${3:-3}
for rep in $(seq $reps); do
 for v in "$A" "$B"; do
  python bench.py --no-cpu-baseline --no-recall $v 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('[%s]' % '$v', round(d['ms_per_step'],4), [(k['name'][3:], round(k['avg_us'],1)) for k in d['roofline']['kernels']])"
 done
done
