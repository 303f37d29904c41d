#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/planab; mkdir -p $O
B="--no-cpu-baseline --no-variants --no-live-pmc --fixed-view --ext-activations"
for rep in 1 2 3; do
  python bench.py $B > $O/on_$rep.json 2>/dev/null
  python -c "
import sys, runpy
sys.path.insert(0, '.')
import collab_splats_amd.ops as o
o.PLAN_CACHE = False
sys.argv = ['bench.py'] + '$B'.split()
runpy.run_path('bench.py', run_name='__main__')" > $O/off_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/planab/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["ms_per_step"], d["device_ms_median"], d["config"]["graph_hit_rate"], d["config"]["graph_timed"])
PY
