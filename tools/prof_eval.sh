#!/bin/bash
# usage (on the GPU box): tools/prof_eval.sh <tag>  -> kernel stats of Recoder.recommend / evaluate at 1 M items
tag=${1:-evalprof}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format rocpd -d $out -o ev -- python tools/eval_bench.py > $out/eval.log 2>&1
python tools/rocpd_stats.py $(find $out -name '*.db') > $out/kernel_stats.md
rm -f $(find $out -name '*.db')
grep -v "Warn\|amdgpu.ids" $out/eval.log | tail -12; head -16 $out/kernel_stats.md | cut -c1-170
