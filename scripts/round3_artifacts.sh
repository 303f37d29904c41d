#!/bin/bash
# usage (through gpurun): bash scripts/round3_artifacts.sh <tag>
# bench lines + kernel stats + PMC passes + timelines + the features-model profile of the current build -> gpurun_out/<tag>/
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
summary() { # $1 = stats dir, $2 = header text
python3 - <<PY
import csv, glob
f = glob.glob("$1/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'stream_copy' not in r['Name'] and 'distribution' not in r['Name'])
calls = max(int(r['Calls']) for r in rows if 'blend_bwd' in r['Name'])
print("# $2")
print("# (the copy-roof probe and the input generation excluded from the total; steps = timed + warm-up + the instrumented pass)")
print("total GPU ms/step", round(tot / calls / 1e6, 4), " (steps:", calls, ")")
for r in rows[:30]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls'])/calls:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/calls/1e6:6.3f}")
PY
}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants > $OUT/stats_bench.log 2>&1
summary $OUT/stats "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants  (the default headline: 1 M Gaussians, 1920x1080, torch activations, 8 cycling views; per-kernel averages are over the views)" > $OUT/kernel_summary.txt
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --fixed-view --ext-activations > $OUT/stats2_bench.log 2>&1
summary $OUT/stats2 "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants --fixed-view --ext-activations  (round 2's headline form: one fixed view, activations inside the kernels)" > $OUT/kernel_summary_fixed_ext.txt
rm -rf $OUT/stats2
cd $GRAFT_REPO_ROOT
bash scripts/pmc_run.sh $TAG/pmc --fixed-view --ext-activations > $OUT/pmc.log 2>&1 || true
python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1 || true
python3 scripts/pmc_traffic.py $OUT/pmc $(cat collab_splats_amd/_build_rev.txt 2>/dev/null || echo unknown) > $OUT/pmc_traffic.json 2> $OUT/pmc_traffic.err || true
bash scripts/pmc_run.sh $TAG/pmc_cyc > $OUT/pmc_cyc.log 2>&1 || true
python3 scripts/pmc_traffic.py $OUT/pmc_cyc $(cat collab_splats_amd/_build_rev.txt 2>/dev/null || echo unknown) > $OUT/pmc_traffic_cycling.json 2> /dev/null || true
rm -rf $OUT/pmc/*/runc/*agent_info.csv
bash scripts/r3_more.sh $TAG > $OUT/more.txt 2>&1
python3 bench.py --no-cpu-baseline --no-variants --dn-loss --fixed-view --buckets > $OUT/bench_1M_dnloss_fixed_buckets.json 2> $OUT/b1.err
python3 bench.py --no-cpu-baseline --no-variants --dn-loss --gaussians 5000000 --fixed-view --buckets > $OUT/bench_5M_dnloss_fixed_buckets.json 2> $OUT/b2.err
python3 bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 --graphed > $OUT/bench_10k_graphed.json 2> $OUT/b3.err
python3 bench.py --no-cpu-baseline --no-variants --gaussians 10000 --width 256 --height 256 --fixed-view --ext-activations > $OUT/bench_10k_fixed_ext.json 2> $OUT/b4.err
python3 bench.py --no-cpu-baseline --no-variants --gaussians 100000 --fixed-view --ext-activations > $OUT/bench_100k_fixed_ext.json 2> $OUT/b5.err
bash scripts/r3_timelines.sh $TAG > $OUT/timelines_all.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t_dn -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --dn-loss --steps 32 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/scripts/timeline.py $(find $OUT/t_dn -name "*kernel_trace.csv" | head -1) > $OUT/timeline_1M_dnloss_cycling.txt 2>&1; rm -rf $OUT/t_dn
cd $GRAFT_REPO_ROOT
bash scripts/features_profile.sh $TAG/features > $OUT/features_all.txt 2>&1
rm -rf $OUT/features/pmc/*/runc/*agent_info.csv
python3 scripts/bench_brief.py $OUT/bench.json
cat $OUT/more.txt
for f in bench_1M_dnloss_fixed_buckets bench_5M_dnloss_fixed_buckets bench_10k_graphed bench_10k_fixed_ext bench_100k_fixed_ext; do python3 -c "
import json; d=json.load(open('$OUT/$f.json')); print('$f', d['ms_per_step'], d['device_ms_median'], d['value'], d['config'].get('path', {}).get('backward_sink'))"; done
head -8 $OUT/kernel_summary.txt
