#!/bin/bash
# usage (through gpurun): bash scripts/ab_lib.sh <libA.so> <libB.so> [bench args]  -- alternate two builds of the library
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for i in 1 2; do for L in $A $B; do
  MISPLAT_LIB=$GRAFT_REPO_ROOT/collab_splats_amd/$L timeout -k 10 200 python bench.py --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/ab.json")); print("$L", d["ms_per_step"], d["device_ms_median"], d["roofline"]["kernel_ms"])
PY
done; done
