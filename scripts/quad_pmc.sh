#!/bin/bash
# usage (through gpurun): bash scripts/quad_pmc.sh   -- band vs four-sub-block backward: time and SQ counters
set -e
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/quadpmc
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -q -k "four_sub_blocks" > $OUT/test.log 2>&1 || (tail -40 $OUT/test.log; exit 1)
tail -1 $OUT/test.log
for sb in 1 4; do
  MISPLAT_SUB_BLOCKS=$sb timeout -k 10 200 python bench.py --no-cpu-baseline > $OUT/b$sb.json 2>$OUT/b$sb.err
  python - <<PY
import json
d=json.load(open("$OUT/b$sb.json"))
print("sub_blocks=$sb", d["ms_per_step"], d["device_ms_median"], d["roofline"]["kernel_ms"])
PY
done
cd /tmp && export TMPDIR=/tmp
export MISPLAT_SUB_BLOCKS=4
ARGS="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python3 $ARGS > $OUT/sq2.log 2>&1 || true
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_FLAT_LDS_ONLY --output-format csv -d $OUT/sq3 -- python3 $ARGS > $OUT/sq3.log 2>&1 || true
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_summary.py $OUT 2>&1 | grep "blend_bwd"
