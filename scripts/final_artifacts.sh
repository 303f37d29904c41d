#!/bin/bash
# usage (through gpurun): bash scripts/final_artifacts.sh <tag>
# Produces under gpurun_out/<tag>/: bench.json (default bench.py run), kernel stats of the same command
# (rocprofv3 --kernel-trace --stats), PMC passes (scripts/pmc_run.sh) and their summary.
set -e
TAG=${1:-final}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/stats_bench.log 2>&1
python3 - <<PY > $OUT/kernel_summary.txt
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = max(int(r['Calls']) for r in rows if 'blend_bwd' in r['Name'])
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline  (1M Gaussians, 1920x1080)")
print("total GPU ms/step", round(tot / calls / 1e6, 4), " (steps incl. warmup:", calls, ")")
for r in rows[:32]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls'])/calls:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/calls/1e6:6.3f}")
PY
cp $f_dummy /dev/null 2>/dev/null || true
cd $GRAFT_REPO_ROOT
bash scripts/pmc_run.sh $TAG/pmc > $OUT/pmc.log 2>&1
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1
cat $OUT/bench.json | cut -c1-600
head -12 $OUT/kernel_summary.txt
