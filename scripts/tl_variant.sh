#!/bin/bash
# usage (through gpurun): bash scripts/tl_variant.sh <tag> [bench flags]  -- kernel timeline of one steady-state step under the
# caller's environment (MISPLAT_* knobs), written to gpurun_out/tl_<tag>.txt
set -e
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/tlv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --steps 30 "$@" > $OUT/b.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 scripts/timeline.py $f > gpurun_out/tl_$TAG.txt
rm -rf $OUT
