#!/bin/bash
# usage (through gpurun): bash scripts/r3_model.sh -- the model step (--dn-loss) at 1 M / 5 M, fixed and cycling views, with and without the SSIM half
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants --dn-loss "$@" > gpurun_out/m.json 2> gpurun_out/m.err || tail -3 gpurun_out/m.err
  python3 - "$@" <<PY
import json, sys
d = json.load(open("gpurun_out/m.json")); print("%-60s" % " ".join(sys.argv[1:]), d["ms_per_step"], d["device_ms_median"])
PY
}
for i in 1 2; do
run --gaussians 5000000 --fixed-view
run --gaussians 5000000 --fixed-view --no-ssim
run --gaussians 5000000
run --fixed-view
run --fixed-view --no-ssim
run
done
