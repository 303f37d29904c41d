#!/bin/bash
# usage (through gpurun): bash scripts/ab_multi.sh "ENV1=a ENV2=b" "ENV1=c" ... -- [bench args]: alternate several environment settings, three rounds
cd $GRAFT_REPO_ROOT
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done; shift
for i in 1 2 3; do for x in "${SETS[@]}"; do
  env $x timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/ab.json")); print("$x", d["ms_per_step"], d["device_ms_median"])
PY
done; done
