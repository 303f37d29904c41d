#!/bin/bash
# usage (through gpurun): bash scripts/r3_tl5m.sh <tag> -- kernel timeline of the 5 M model step (fixed view, dn loss)
TAG=${1:-tl}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t_5M -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --steps 32 --dn-loss --gaussians 5000000 --fixed-view > $OUT/tl_5M.log 2>&1
f=$(find $OUT/t_5M -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/scripts/timeline.py $f > $OUT/timeline_5M_dnloss_fixed.txt 2>&1
rm -rf $OUT/t_5M
cat $OUT/timeline_5M_dnloss_fixed.txt
