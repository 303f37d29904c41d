#!/bin/bash
# GPU: full -m gpu suite + headline bench (no cpu baseline) into gpurun_out/$1
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05b}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1 || { tail -60 $O/gpu_tests.txt; exit 1; }
tail -3 $O/gpu_tests.txt
B="--no-cpu-baseline --no-variants --no-live-pmc"
timeout -k 10 300 python bench.py $B > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python bench.py $B --fixed-view --ext-activations > $O/bench_fixed.json 2> $O/bench_fixed.err
python - <<PY
import json
for t in ("bench","bench_fixed"):
    d=json.loads(open("$O/%s.json"%t).read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["value"], d["roofline"]["kernel_ms"])
PY
