#!/bin/bash
# usage (through gpurun): bash scripts/small_timeline.sh  -- kernel timelines of the small configurations
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/small
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t10k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 --steps 40 > $OUT/b10k.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t100k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --gaussians 100000 --steps 40 > $OUT/b100k.log 2>&1
cd $GRAFT_REPO_ROOT
for t in t10k t100k; do
  f=$(find $OUT/$t -name "*kernel_trace.csv" | head -1)
  echo "== $t"; python3 scripts/timeline.py $f > $OUT/$t.txt; head -60 $OUT/$t.txt
  rm -rf $OUT/$t
done
cut -c1-200 $OUT/b10k.log | tail -1; cut -c1-200 $OUT/b100k.log | tail -1
