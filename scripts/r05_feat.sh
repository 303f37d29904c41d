#!/bin/bash
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/feat; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "features or channel or steady or model" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for r in 1 2; do
timeout -k 10 300 python bench.py --features 13 --fixed-view --no-cpu-baseline --no-variants --no-live-pmc > $O/f_$r.json 2> $O/f_$r.err
python - <<PY
import json
d=json.loads(open("$O/f_$r.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
done
