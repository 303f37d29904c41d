#!/bin/bash
# usage (through gpurun): bash scripts/ssim_pmc.sh <tag> -- SQ counters of the two SSIM kernels (scripts/ssim_time.py)
TAG=${1:-ssimpmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $GRAFT_REPO_ROOT/scripts/ssim_time.py > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 $GRAFT_REPO_ROOT/scripts/ssim_time.py > $OUT/sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("sq", "sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            if "ssim" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, v in acc.items():
            print(k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
