#!/bin/bash
# usage: scripts/prof_run.sh <tag> [bench args]   (on the GPU box through gpurun)
TAG=${1:-prof}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = max(int(r['Calls']) for r in rows if 'blend_bwd' in r['Name'])
print("total GPU ms/step", tot / calls / 1e6)
for r in rows[:22]:
    print(f"{r['Name'][:70]:70s} calls/step={int(r['Calls'])/calls:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/calls/1e6:6.3f}")
PY
grep '"metric"' $OUT/bench.log | cut -c1-200
