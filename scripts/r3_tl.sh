#!/bin/bash
# usage (through gpurun): bash scripts/r3_tl.sh <tag> <name> <bench args...> -- kernel timeline of one steady-state step of a bench configuration
TAG=$1; NAME=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$NAME -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --steps 32 "$@" > $OUT/tl_$NAME.log 2>&1
f=$(find $OUT/t_$NAME -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/timeline.py $f > $OUT/timeline_$NAME.txt 2>&1
rm -rf $OUT/t_$NAME
cat $OUT/timeline_$NAME.txt
