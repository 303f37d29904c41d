#!/bin/bash
# usage (through gpurun): bash scripts/pmc_kernel.sh <tag> <kernel substring> [bench flags] -- FETCH_SIZE / WRITE_SIZE / duration
# counters of ONE kernel under the caller's MISPLAT_* environment (counters only, separate passes), summarised to
# gpurun_out/pmck_<tag>.txt
set -e
TAG=$1; KER=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-variants $*"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/write.log 2>&1
n=0; for grp in ${PMCK_EXTRA:-}; do n=$((n+1))       # PMCK_EXTRA="A,B C,D": one more counters-only pass per group
  rocprofv3 --pmc ${grp//,/ } --output-format csv -d $OUT/extra$n -- python3 $ARGS > $OUT/extra$n.log 2>&1 || tail -3 $OUT/extra$n.log
done
cd $GRAFT_REPO_ROOT
python3 - "$OUT" "$KER" > gpurun_out/pmck_$TAG.txt <<'PY'
import csv, glob, sys, collections
root, ker = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ker in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    v = v[len(v) // 2:]
    print(f"{ker} {k}: mean {sum(v) / len(v):.6g} over {len(v)} launches")
PY
rm -rf $OUT
cat gpurun_out/pmck_$TAG.txt
