#!/bin/bash
# C5-shaped configurations: parity tests that cover them + the bench lines + kernel stats of both batch sizes
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
o=gpurun_out/${1:-c5}; mkdir -p $o
timeout 1200 python -m pytest tests/test_full_size.py tests/test_hip_parity.py -q -m gpu -k "c5 or step_case or big_then_ragged or ragged_1024 or fuzz" > $o/pytest.log 2>&1; grep -n "passed\|failed" $o/pytest.log | tail -2; grep -n "^FAILED\|^E  " $o/pytest.log | head
for c in c5u c5u4k; do
  rocprofv3 --kernel-trace --stats --output-format rocpd -d $o -o st_$c -- python bench.py --config $c --no-cpu-baseline --no-recall > $o/st_$c.log 2>&1
  python tools/rocpd_stats.py $(find $o -name "st_${c}_results.db") > $o/kernel_stats_$c.md; rm -f $(find $o -name "*.db")
  head -12 $o/kernel_stats_$c.md | cut -c1-150
  python bench.py --config $c --no-cpu-baseline --no-recall 2>/dev/null | tail -1 > $o/bench_$c.json
  python -c "import json; d=json.loads(open('$o/bench_$c.json').read()); print('$c', d['ms_per_step'], d['value'])"
done
