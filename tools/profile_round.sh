#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>
# kernel-trace stats of the default bench (graph replay) + the two PMC traffic passes of the SAME command (round 6:
# under graph replay too -- the lazy Adam sweep only exists there; rocprofv3 counts replayed kernels) -> gpurun_out/<tag>/
tag=${1:-prof}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf $out; mkdir -p $out
B="python bench.py --steps 200 --warmup 24 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format rocpd -d $out -o stats -- $B > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format rocpd -d $out -o fetch -- $B > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format rocpd -d $out -o write -- $B > $out/write.log 2>&1
python tools/rocpd_stats.py $(find $out -name 'stats_results.db') > $out/kernel_stats.md
python tools/pmc_traffic.py $(find $out -name 'fetch_results.db') $(find $out -name 'write_results.db') > $out/pmc_traffic.json
python tools/rocpd_gaps.py $(find $out -name 'stats_results.db') 60 120 > $out/timeline.txt
grep metric $out/stats.log > $out/bench_profiled.json
# SQ counters of the contraction kernels (one more pass, its own run)
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format rocpd -d $out -o sq -- $B > $out/sq.log 2>&1
python tools/rocpd_pmc.py $(find $out -name 'sq_results.db') > $out/sq_counters.txt 2>&1
$B > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_steps20.json 2>> $out/bench.err
rm -f $(find $out -name '*.db')
cat $out/kernel_stats.md | head -20; cat $out/pmc_traffic.json; cat $out/timeline.txt
