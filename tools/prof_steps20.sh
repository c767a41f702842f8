#!/bin/bash
# usage (on the GPU box): tools/prof_steps20.sh <tag>  -> kernel timeline of the driver's short run
tag=$1; shift
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format rocpd -d $out -o s20 -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/bench.log 2>&1
db=$(find $out -name '*.db' | head -1)
grep metric $out/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'
python tools/rocpd_dump.py $db 0 40 > $out/dump.txt
rm -f $db
