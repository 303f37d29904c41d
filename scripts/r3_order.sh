#!/bin/bash
# usage (through gpurun): bash scripts/r3_order.sh -- what the forward's launch order is worth: previous-step order on/off, cycling and fixed views
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-variants "$@" > gpurun_out/o.json 2> gpurun_out/o.err || tail -3 gpurun_out/o.err
  python3 - "$MISPLAT_UNIT_ORDER_FWD" "$@" <<PY
import json, sys
d = json.load(open("gpurun_out/o.json")); print("fwd_order=%s %-30s" % (sys.argv[1], " ".join(sys.argv[2:])), d["ms_per_step"], d["device_ms_median"], d["roofline"]["kernel_ms"])
PY
}
for i in 1 2; do
for o in 1 0; do
export MISPLAT_UNIT_ORDER_FWD=$o
run
run --fixed-view
done; done
