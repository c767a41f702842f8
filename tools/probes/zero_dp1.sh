export MASTER_ADDR=127.0.0.1
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_dp_two_process.py -x -q -m gpu -k "one_rank_equals_single_process or virtual_ranks or two_process or graph_replay_is_bitwise_equal_to_host" > gpurun_out/zero_pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/zero_pytest.log | tail -3; grep -n "Error\|assert" gpurun_out/zero_pytest.log | head -20
for z in 0 force; do
RK_FORCE_DP=1 RK_DP_ZERO=$z MASTER_PORT=$((29000 + RANDOM % 900)) timeout 300 python bench.py --no-cpu-baseline --no-recall 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('c2 dp1 zero=$z', round(d['ms_per_step'],4), d['config'].get('graph_replay'), [(k['name'][3:], round(k['avg_us'],1)) for k in d['roofline']['kernels']])"
done
