#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/small; mkdir -p $O
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-live-pmc --steps 200 --warmup 40 --gaussians 10000 --width 256 --height 256 > $O/10k_$rep.json 2>/dev/null
python bench.py --no-cpu-baseline --no-live-pmc --steps 200 --warmup 40 --gaussians 100000 > $O/100k_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/small/*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["ms_per_step"], {k:v[0] for k,v in d["variants"].items()}, d["config"]["graph_hit_rate"], d["roofline"]["kernel_ms"])
PY
