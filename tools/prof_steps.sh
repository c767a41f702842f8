#!/bin/bash
# usage (on the GPU box): tools/prof_steps.sh <tag> <steps> <first> <n> [bench flags]  -> kernel timeline (n steps from the first-th
# encoder forward) of a bench run of <steps> steps under graph replay
tag=$1; steps=$2; first=$3; n=$4; shift 4
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
rocprofv3 --kernel-trace --output-format rocpd -d $out -o st -- python bench.py --steps $steps --warmup 24 --no-cpu-baseline --no-recall "$@" > $out/bench.log 2>&1
db=$(find $out -name '*.db' | head -1)
python tools/rocpd_dump.py $db $first $n > $out/dump.txt
rm -f $db
