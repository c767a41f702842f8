export MASTER_ADDR=127.0.0.1 MASTER_PORT=29655
bash tools/profile_round.sh r03j > gpurun_out/r03j_console.txt 2>&1
o=gpurun_out/r03j
python bench.py > $o/bench_default.json 2> $o/bench_default.err
for c in c3 c4 c5u; do python bench.py --config $c --no-cpu-baseline > $o/bench_$c.json 2>> $o/bench.err; done
RK_GEMM_PREC=bf16 python bench.py --no-cpu-baseline > $o/bench_c2_bf16.json 2>> $o/bench.err
RK_FORCE_DP=1 python bench.py --no-cpu-baseline --no-recall > $o/bench_c2_dp1.json 2>> $o/bench.err
bash tools/prof_steps20.sh r03j_s20 > $o/steps20_console.txt 2>&1
cp gpurun_out/r03j_s20/dump.txt $o/steps20_timeline.txt
tail -3 gpurun_out/r03j_console.txt
for f in $o/bench_default.json $o/bench_c3.json $o/bench_c4.json $o/bench_c5u.json $o/bench_c2_bf16.json $o/bench_c2_dp1.json $o/bench_steps20.json; do python - "$f" <<'P'
import sys, json
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"), d.get("recall_at_20"), d.get("recall_match_4dp"))
except Exception as e:
  print(sys.argv[1], "ERR", e)
P
done
