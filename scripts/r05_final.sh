#!/bin/bash
# GPU: what the driver runs at round end -- the gpu-marked tests, smoke(), the default bench line
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1 || { tail -60 $O/gpu_tests.txt; exit 1; }
tail -3 $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
tail -c 2600 $O/bench_driver.json
