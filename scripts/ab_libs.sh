#!/bin/bash
# usage (through gpurun): bash scripts/ab_libs.sh "<bench args>" <lib1.so|default> <lib2.so> ...  -- alternate builds of the library (paths relative to collab_splats_amd/)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for i in 1 2; do for L in "$@"; do
  if [ "$L" = "default" ]; then unset MISPLAT_LIB; else export MISPLAT_LIB=$GRAFT_REPO_ROOT/collab_splats_amd/$L; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants $ARGS > gpurun_out/ab.json 2> gpurun_out/ab.err || tail -3 gpurun_out/ab.err
  python3 - <<PY
import json
d=json.load(open("gpurun_out/ab.json")); print("round $i %-36s" % "$L", d["ms_per_step"], d["device_ms_median"], d["roofline"]["kernel_ms"])
PY
done; done
