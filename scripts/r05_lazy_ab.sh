#!/bin/bash
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/lazyab; mkdir -p $O
B="--no-cpu-baseline --no-variants --no-live-pmc"
for m in auto 0; do
  MISPLAT_LAZY_SH=$m timeout -k 10 300 python bench.py $B > $O/lazy_$m.json 2>$O/lazy_$m.err
done
python - <<'PY'
import json
for m in ("auto","0"):
    d=json.loads(open(f"gpurun_out/lazyab/lazy_{m}.json").read().strip().splitlines()[-1])
    print(m, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["units"])
PY
