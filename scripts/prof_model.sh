#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-profm}
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/model_step.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
calls = max(int(r['Calls']) for r in rows if 'blend_bwd' in r['Name'])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("total GPU ms/step", tot / calls / 1e6)
for r in rows[:30]:
    print(f"{r['Name'][:80]:80s} calls/step={int(r['Calls'])/calls:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/calls/1e6:6.3f}")
PY
tail -1 $OUT/log.txt
