#!/bin/bash
# one forced RCCL rank through the users-DP paths of the SparseAdam configurations (C4, C5-shaped):
# replicated update (graph replay) vs owned-row Adam (host-sequenced), and the single-process step beside them
export MASTER_ADDR=127.0.0.1
o=gpurun_out/${1:-dp1}; mkdir -p $o
if [ "$2" = "tests" ]; then timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "one_rank_equals_single_process" > $o/pytest.log 2>&1; tail -3 $o/pytest.log; fi
for mode in 0 force; do
  for c in c4 c5u; do
    RK_FORCE_DP=1 RK_DP_OWNED=$mode MASTER_PORT=$((29000 + RANDOM % 900)) timeout 300 python bench.py --config $c --no-cpu-baseline --no-recall > $o/bench_${c}_dp1_owned$mode.json 2> $o/bench_${c}_dp1_owned$mode.err
    tail -1 $o/bench_${c}_dp1_owned$mode.json | python -c "
import sys,json
try:
  d=json.loads(sys.stdin.read()); print('$c owned=$mode', round(d['ms_per_step'],4), d['config'].get('graph_replay'), d['config']['parallelism'])
except Exception as e: print('$c owned=$mode FAILED', e)"
    tail -3 $o/bench_${c}_dp1_owned$mode.err
  done
done
