#!/bin/bash
# GPU: the default bench line (driver's protocol and the default one) + the dense 5 M model step, with details files
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-bench}; mkdir -p $O
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --details $O/driver_protocol_details.json > $O/driver_protocol.json 2> $O/driver_protocol.err
timeout -k 10 300 python bench.py --dn-loss --gaussians 5000000 --fixed-view --details $O/5M_dnloss_fixed_details.json > $O/5M_dnloss_fixed.json 2> $O/5M_dnloss_fixed.err
timeout -k 10 300 python bench.py --dn-loss --gaussians 5000000 --details $O/5M_dnloss_details.json > $O/5M_dnloss.json 2> $O/5M_dnloss.err
python - "$O" <<'PY'
import json, sys
O=sys.argv[1]
for t in ("driver_protocol","5M_dnloss_fixed","5M_dnloss"):
    raw=open(f"{O}/{t}.json").read().strip().splitlines()[-1]
    d=json.loads(raw)
    r=d["roofline"]
    print(t, len(raw), "bytes;", d["ms_per_step"], d["value"], "frac", r["frac"], r["bytes_basis"], "nominal", r["frac_nominal"], "step", r["step_frac"], r["step_frac_nominal"], "roof", r["copy_roof_GBs"]/8000, r["units"])
PY
