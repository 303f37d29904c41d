#!/bin/bash
# usage (through gpurun): bash scripts/timeline_1m.sh [extra bench flags]  -- kernel timeline of one steady-state 1 M step
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl1m
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t1m -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 "$@" > $OUT/b1m.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/t1m -name "*kernel_trace.csv" | head -1)
python3 scripts/timeline.py $f > $OUT/t1m.txt
rm -rf $OUT/t1m
cat $OUT/t1m.txt; cut -c1-200 $OUT/b1m.log | tail -1
