#!/bin/bash
# the streaming fused decode: parity tests of the large-batch domain + C2 at B = 4000 with it (default) and without
o=gpurun_out/${1:-fds}; mkdir -p $o
timeout 1200 python -m pytest tests/test_pgemm.py tests/test_hip_parity.py tests/test_gemm_precision.py -q -m gpu -k "fdec or step or ragged or fuzz or precision" > $o/pytest.log 2>&1; grep -n "passed\|failed" $o/pytest.log | tail -2; grep -n "^FAILED\|^E  " $o/pytest.log | head
for v in 1 0; do
  python bench.py --config c2b4k --steps 60 --warmup 16 --no-cpu-baseline --tune 11=$v 2>/dev/null | tail -1 > $o/c2b4k_$v.json
  python -c "
import json
d=json.loads(open('$o/c2b4k_$v.json').read()); print('c2b4k stream=$v', round(d['ms_per_step'],4), round(d['value']), d.get('recall_match_4dp'), [(k['name'][3:], round(k['avg_us'],1)) for k in d['roofline']['kernels']])"
done
