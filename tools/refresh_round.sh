#!/bin/bash
# usage (on the GPU box): bash tools/refresh_round.sh <tag>
# every committed number of a round in one call: tools/profile_round.sh (C2 kernel stats, PMC traffic,
# SQ counters, bench lines), the other configurations' bench lines and kernel stats, the bf16 and
# one-rank RCCL data points, the 20-step timeline -> gpurun_out/<tag>/
tag=${1:-r06g}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 TMPDIR=/tmp
timeout 900 bash tools/profile_round.sh $tag > gpurun_out/${tag}_console.txt 2>&1
o=gpurun_out/$tag
timeout 300 python bench.py > $o/bench_default.json 2> $o/bench_default.err
# the driver's exact command
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver_cmd.json 2>> $o/bench_default.err
for c in c3 c3mse c4 c5u c5u4k; do timeout 300 python bench.py --config $c --no-cpu-baseline > $o/bench_$c.json 2>> $o/bench.err; done
for c in c3 c4 c5u c5u4k; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd -d $o -o st_$c -- python bench.py --config $c --no-cpu-baseline --no-recall > $o/st_$c.log 2>&1
  python tools/rocpd_stats.py $(find $o -name "st_${c}_results.db") > $o/kernel_stats_$c.md 2>> $o/bench.err
done
RK_GEMM_PREC=bf16 timeout 300 python bench.py --no-cpu-baseline > $o/bench_c2_bf16.json 2>> $o/bench.err
RK_GEMM_PREC=bf16 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_c2_bf16_steps20.json 2>> $o/bench.err
RK_GEMM_PREC=bf16 timeout 300 rocprofv3 --kernel-trace --stats --output-format rocpd -d $o -o st_bf16 -- python bench.py --no-cpu-baseline --no-recall > $o/st_bf16.log 2>&1
python tools/rocpd_stats.py $(find $o -name "st_bf16_results.db") > $o/kernel_stats_c2_bf16.md 2>> $o/bench.err
# round 6: every row swept every step (the lazy Adam off), the same command otherwise
RK_ADAM_LAZY=0 timeout 300 python bench.py --no-cpu-baseline --no-recall > $o/bench_c2_dense_adam.json 2>> $o/bench.err
RK_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --no-recall > $o/bench_c2_dp1.json 2>> $o/bench.err
# round 5: sharded dense Adam forced on one rank (the dense layout + ncclReduceScatter + the row-range sweep + the publish)
RK_FORCE_DP=1 RK_DP_ZERO=force MASTER_PORT=29656 timeout 300 python bench.py --no-cpu-baseline --no-recall > $o/bench_c2_dp1_zero.json 2>> $o/bench.err
# the SparseAdam configurations under one forced RCCL rank: replicated update (graph replay) vs owned rows (host-sequenced)
for c in c4 c5u; do for m in 0 force; do
  RK_FORCE_DP=1 RK_DP_OWNED=$m MASTER_PORT=$((29660 + RANDOM % 200)) timeout 300 python bench.py --config $c --no-cpu-baseline --no-recall > $o/bench_${c}_dp1_owned$m.json 2>> $o/bench.err
done; done
# C2 at B = 4000 (config.alt_large_batch of a multi-GPU run)
timeout 300 python bench.py --config c2b4k --steps 60 --warmup 16 --no-cpu-baseline > $o/bench_c2b4k.json 2>> $o/bench.err
# ... and 24 steps inside ONE epoch of 29 (no epoch boundary in the clock)
timeout 300 python bench.py --config c2b4k --steps 24 --warmup 4 --no-cpu-baseline > $o/bench_c2b4k_one_epoch.json 2>> $o/bench.err
# the driver's 20-step run by the round-3 methodology (first group's collation, cold state and the
# bracketed group all inside the clock) next to the default one
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-recall --no-precollate --no-pretouch --sample timed > $o/bench_steps20_r3method.json 2>> $o/bench.err
timeout 300 bash tools/prof_steps20.sh ${tag}_s20 > $o/steps20_console.txt 2>&1
cp gpurun_out/${tag}_s20/dump.txt $o/steps20_timeline.txt
timeout 120 python tools/probes/fdec_probe.py > $o/fdec_probe.txt 2>&1
rm -f $(find $o -name '*.db')
tail -3 gpurun_out/${tag}_console.txt
for f in $o/bench_default.json $o/bench_driver_cmd.json $o/bench.json $o/bench_steps20.json $o/bench_steps20_r3method.json $o/bench_c3.json $o/bench_c3mse.json $o/bench_c4.json $o/bench_c5u.json $o/bench_c5u4k.json $o/bench_c2b4k.json $o/bench_c2b4k_one_epoch.json $o/bench_c2_bf16.json $o/bench_c2_bf16_steps20.json $o/bench_c2_dense_adam.json $o/bench_c2_dp1.json $o/bench_c2_dp1_zero.json $o/bench_c4_dp1_owned0.json $o/bench_c4_dp1_ownedforce.json $o/bench_c5u_dp1_owned0.json $o/bench_c5u_dp1_ownedforce.json; do python - "$f" <<'P'
import sys, json
try:
  d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
  print(sys.argv[1].split('/')[-1], d["ms_per_step"], d["value"], d.get("roofline", {}).get("frac"), d.get("recall_at_20"), d.get("recall_match_4dp"))
except Exception as e:
  print(sys.argv[1], "ERR", e)
P
done
