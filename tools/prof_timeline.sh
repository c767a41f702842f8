#!/bin/bash
# usage (on the GPU box): tools/prof_timeline.sh <tag> [bench args]  -> gpurun_out/<tag>/c2_results.db + timeline
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format rocpd -d $out -o c2 -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $out/bench.log 2>&1
db=$(find $out -name '*.db' | head -1)
grep metric $out/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
python tools/rocpd_timeline.py $db 150 > $out/timeline.txt
cat $out/timeline.txt
