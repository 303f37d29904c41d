#!/bin/bash
# usage (through gpurun): bash scripts/ab_env.sh VAR valA valB [bench args]  -- alternate two settings of an environment knob
cd $GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; shift 3
for i in 1 2 3; do for x in $A $B; do
  env $V=$x timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/ab.json")); print("$V=$x", d["ms_per_step"], d["device_ms_median"])
PY
done; done
