#!/bin/bash
# usage (through gpurun): bash scripts/r3_timelines.sh <tag> -- kernel timelines of one steady-state step of several configurations
TAG=${1:-tl}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
tl() { name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --steps 32 "$@" > $OUT/tl_$name.log 2>&1
  f=$(find $OUT/t_$name -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/scripts/timeline.py $f > $OUT/timeline_$name.txt 2>&1
  rm -rf $OUT/t_$name
  echo "== $name"; cat $OUT/timeline_$name.txt
}
tl 1M_fixed_ext --fixed-view --ext-activations
tl 1M_cycling_torch
tl 5M_dnloss_fixed --dn-loss --gaussians 5000000 --fixed-view
tl 100k_fixed_ext --gaussians 100000 --fixed-view --ext-activations
tl 10k_fixed_ext --gaussians 10000 --width 256 --height 256 --fixed-view --ext-activations
