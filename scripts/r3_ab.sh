#!/bin/bash
# usage (through gpurun): bash scripts/r3_ab.sh <tag> "<bench flags>" "ENV1=.. ENV2=.." "ENV=.." ...
# the headline bench (no CPU leg, no variants) under alternative environment settings, two alternating rounds
TAG=${1:-ab}; shift
FLAGS=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  i=0
  for cfg in "" "$@"; do
    env $cfg python bench.py --no-cpu-baseline --no-variants $FLAGS > $OUT/ab_${i}_r$round.json 2> $OUT/ab_${i}_r$round.err || { echo "bench failed: $cfg"; tail -5 $OUT/ab_${i}_r$round.err; }
    python - <<PY
import json
try:
    d = json.load(open("$OUT/ab_${i}_r$round.json"))
    k = d["roofline"]["kernel_ms"]
    print(f"round $round [{'$cfg' or 'default':44s}] {d['ms_per_step']:.4f} ms  dev {d['device_ms_median']:.4f}  fwd {k.get('blend_fwd', 0):.3f} bwd {k.get('blend_bwd', 0):.3f}  graphs {d['config']['graph_cache']}")
except Exception as e:
    print("round $round [$cfg] unreadable:", e)
PY
    i=$((i+1))
  done
done
