#!/bin/bash
# usage (here, after the gpurun call of tools/refresh_round.sh <tag> came back): bash tools/collect_profiles.sh <tag> <round prefix>
# copies what the judge reads from gpurun_out/<tag>/ (scratch) to profiles/<prefix>_* (tracked)
tag=${1:-r06b}; r=${2:-r06}; o=gpurun_out/$tag; p=profiles
cp $o/bench_default.json $p/${r}_bench_c2_with_cpu_baseline.json
cp $o/bench_driver_cmd.json $p/${r}_bench_c2_driver_cmd.json
cp $o/bench.json $p/${r}_bench_c2.json
cp $o/bench_profiled.json $p/${r}_bench_c2_profiled.json
cp $o/bench_steps20.json $p/${r}_bench_c2_steps20.json
cp $o/bench_steps20_r3method.json $p/${r}_bench_c2_steps20_r3method.json
for c in c3 c3mse c4 c5u c5u4k c2b4k c2b4k_one_epoch c2_bf16 c2_bf16_steps20 c2_dense_adam c2_dp1 c2_dp1_zero; do cp $o/bench_$c.json $p/${r}_bench_$c.json; done
cp $o/bench_c4_dp1_ownedforce.json $p/${r}_bench_c4_dp1_owned.json
cp $o/bench_c5u_dp1_ownedforce.json $p/${r}_bench_c5u_dp1_owned.json
cp $o/bench_c4_dp1_owned0.json $p/${r}_bench_c4_dp1.json
cp $o/bench_c5u_dp1_owned0.json $p/${r}_bench_c5u_dp1.json
cp $o/kernel_stats.md $p/${r}_kernel_stats.md
for c in c3 c4 c5u c5u4k c2_bf16; do cp $o/kernel_stats_$c.md $p/${r}_kernel_stats_$c.md; done
cp $o/pmc_traffic.json $p/${r}_pmc_traffic.json
cp $o/sq_counters.txt $p/${r}_sq_counters.txt
cp $o/timeline.txt $p/${r}_step_timeline.txt
cp $o/steps20_timeline.txt $p/${r}_steps20_timeline.txt
cp $o/fdec_probe.txt $p/${r}_fdec_probe.txt
