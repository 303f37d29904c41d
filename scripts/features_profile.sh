#!/bin/bash
# usage (through gpurun): bash scripts/features_profile.sh <tag>  -- the features-model call: bench line, kernel trace summary, PMC of its compositing kernels
TAG=${1:-feat}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python scripts/features_step.py --json --both > $OUT/features_bench.json 2> $OUT/features_bench.err || tail -5 $OUT/features_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/scripts/features_step.py --steps 10 > $OUT/stats.log 2>&1
python3 - <<PY > $OUT/features_kernel_summary.txt
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
calls = max(int(r['Calls']) for r in rows if 'blend_bwd' in r['Name'])
tot = sum(float(r['TotalDurationNs']) for r in rows if 'distribution' not in r['Name'])
print("# rocprofv3 --kernel-trace --stats -- python3 scripts/features_step.py --steps 10   (features-model call: 16 fused channels + ED, 1 M / 1080p)")
print("total GPU ms/step", round(tot / calls / 1e6, 4), " (steps:", calls, ")")
for r in rows[:24]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls'])/calls:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/calls/1e6:6.3f}")
PY
rm -rf $OUT/stats
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc/sq -- python3 $GRAFT_REPO_ROOT/scripts/features_step.py --steps 2 > $OUT/pmc_sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/fetch -- python3 $GRAFT_REPO_ROOT/scripts/features_step.py --steps 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -- python3 $GRAFT_REPO_ROOT/scripts/features_step.py --steps 2 > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc/sq2 -- python3 $GRAFT_REPO_ROOT/scripts/features_step.py --steps 2 > $OUT/pmc_sq2.log 2>&1 || true
cd $GRAFT_REPO_ROOT
python3 scripts/pmc_traffic.py $OUT/pmc $(cat collab_splats_amd/_build_rev.txt 2>/dev/null || echo unknown) > $OUT/features_pmc.json 2> $OUT/pmc_traffic.err
cat $OUT/features_bench.json; head -14 $OUT/features_kernel_summary.txt; python3 - <<PY
import json
d = json.load(open("$OUT/features_pmc.json"))
print({k: d.get(k) for k in ("blend_fwd", "blend_bwd", "valu_issue_frac")})
for k in ("blend_fwd", "blend_bwd"):
    c = d["_counters"].get(k, {}); print(k, {x: c.get(x) for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT")})
PY
