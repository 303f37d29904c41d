#!/bin/bash
# usage (through gpurun): bash scripts/r3_check.sh <tag> [pytest -k expression]
# GPU tests + the default bench line of the current build -> gpurun_out/<tag>/
TAG=${1:-r3}
KEXPR=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export FLIPPROOF_LOG=$OUT/flipproof.log
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/test.log 2>&1
else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/test.log 2>&1
fi
RC=$?
tail -3 $OUT/test.log
if [ -f $FLIPPROOF_LOG ]; then python - <<PY
import re
L = open("$FLIPPROOF_LOG").read().splitlines()
tot = sum(int(re.search(r": (\d+) out-of", l).group(1)) for l in L)
strict = sum(int(re.search(r", (\d+) cover a pixel that differs", l).group(1)) for l in L)
loose = sum(int(re.search(r", (\d+) cover only", l).group(1)) for l in L)
print(f"FlipProof: {len(L)} tensor(s) with out-of-tolerance rows, {tot} rows, {strict} explained by a differing pixel, {loose} only by a low-margin pixel")
PY
fi
if [ $RC -ne 0 ]; then echo "TESTS FAILED rc=$RC"; grep -n "Error\|assert\|FAILED" $OUT/test.log | head -40; exit $RC; fi
timeout -k 10 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "BENCH FAILED"; tail -20 $OUT/bench.err; exit 1; }
python scripts/bench_brief.py $OUT/bench.json
