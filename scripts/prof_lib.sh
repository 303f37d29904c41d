#!/bin/bash
# usage: scripts/prof_lib.sh <tag> [lib.so]  -- kernel stats of bench.py with an alternative library build
TAG=$1; LIB=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
if [ -n "$LIB" ]; then export MISPLAT_LIB=$GRAFT_REPO_ROOT/$LIB; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
calls = max(int(r['Calls']) for r in rows if 'blend_bwd' in r['Name'])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print("$TAG total", round(tot / calls / 1e6, 4), " ".join(f"{k}={float(r['AverageNs'])/1e3:.1f}" for r in rows for k in (("ppb" if 'project_pack_bwd' in r['Name'] else "shb" if 'color_sh_kernel<true' in r['Name'] else ""),) if k))
PY
