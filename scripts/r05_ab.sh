#!/bin/bash
# GPU: a test selection on the shipped library, then kernel times of variant libraries beside it (scripts/r05_perf.sh)
set -e -o pipefail
cd $GRAFT_REPO_ROOT
T=$1; K=$2; shift 2
mkdir -p gpurun_out/$T
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "$K" > gpurun_out/$T/pytest.log 2>&1 || { tail -40 gpurun_out/$T/pytest.log; exit 1; }
tail -2 gpurun_out/$T/pytest.log
bash scripts/r05_perf.sh $T "$@"
