#!/bin/bash
# same box: round 4's tree (collab_splats_amd/_exp/r04tree) against the current one at the small configurations
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/small_ab; mkdir -p $O
for rep in 1 2; do
for t in r04 cur; do
  D=$GRAFT_REPO_ROOT; [ $t = r04 ] && D=$GRAFT_REPO_ROOT/collab_splats_amd/_exp/r04tree
  ( cd $D && python bench.py --no-cpu-baseline --no-live-pmc --gaussians 10000 --width 256 --height 256 > $O/${t}_10k_$rep.json 2>/dev/null
    python bench.py --no-cpu-baseline --no-live-pmc --gaussians 100000 > $O/${t}_100k_$rep.json 2>/dev/null )
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/small_ab/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    v=d["variants"]
    print(f.split("/")[-1], d["ms_per_step"], {k:(x[0] if isinstance(x,list) else x["ms_per_step"]) for k,x in v.items()})
PY
