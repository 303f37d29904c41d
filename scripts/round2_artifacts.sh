#!/bin/bash
# usage (through gpurun): bash scripts/round2_artifacts.sh <tag>
# bench line + kernel stats + PMC passes of the current build -> gpurun_out/<tag>/
set -e
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/stats_bench.log 2>&1
python3 - <<PY > $OUT/kernel_summary.txt
import csv, glob
f = glob.glob("$OUT/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'stream_copy' not in r['Name'] and 'distribution' not in r['Name'])
calls = max(int(r['Calls']) for r in rows if 'blend_bwd' in r['Name'])
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline  (1M Gaussians, 1920x1080; the copy-roof")
print("# probe and the input generation excluded from the total; steps = timed + warm-up + the instrumented pass)")
print("total GPU ms/step", round(tot / calls / 1e6, 4), " (steps:", calls, ")")
for r in rows[:34]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls'])/calls:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/calls/1e6:6.3f}")
PY
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash scripts/pmc_run.sh $TAG/pmc > $OUT/pmc.log 2>&1 || true
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1 || true
python3 scripts/pmc_traffic.py $OUT/pmc $(cat collab_splats_amd/_build_rev.txt 2>/dev/null || echo unknown) > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err || true
# the other configurations (bench lines only) and the small-scene timelines
python3 bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 > $OUT/bench_10k.json 2> $OUT/b10k.err
python3 bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 --graphed > $OUT/bench_10k_graphed.json 2>> $OUT/b10k.err
python3 bench.py --no-cpu-baseline --gaussians 100000 > $OUT/bench_100k.json 2> $OUT/b100k.err
python3 bench.py --no-cpu-baseline --gaussians 100000 --graphed > $OUT/bench_100k_graphed.json 2>> $OUT/b100k.err
python3 bench.py --no-cpu-baseline --dn-loss > $OUT/bench_1M_dnloss.json 2> $OUT/b1mdn.err
python3 bench.py --no-cpu-baseline --dn-loss --gaussians 5000000 > $OUT/bench_5M_dnloss.json 2> $OUT/b5m.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t10k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 --steps 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t100k -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --gaussians 100000 --steps 40 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t1M -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t1Mdn -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --dn-loss --steps 30 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/t5Mdn -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --dn-loss --gaussians 5000000 --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for t in t10k t100k t1M t1Mdn t5Mdn; do
  python3 scripts/timeline.py $(find $OUT/$t -name "*kernel_trace.csv" | head -1) > $OUT/timeline_$t.txt 2>&1 || true
  rm -rf $OUT/$t
done
rm -rf $OUT/stats
for f in bench_10k bench_10k_graphed bench_100k bench_100k_graphed bench_1M_dnloss bench_5M_dnloss; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', d['ms_per_step'], d['device_ms_median'], d['value'])"; done
cut -c1-700 $OUT/bench.json
head -16 $OUT/kernel_summary.txt
cat $OUT/pmc_traffic.json | head -12
