#!/bin/bash
# usage (through gpurun): bash scripts/artifacts.sh <tag> [part ...]      parts: bench stats pmc timelines models small features
# The judged evidence of the current build -> gpurun_out/<tag>/ (copy what is wanted into profiles/ with scripts/collect_profiles.sh).
# Every part appends a progress line to gpurun_out/<tag>/progress.txt (a long run must keep writing).
TAG=${1:-art}; shift
PARTS=${*:-bench stats pmc timelines models small features}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
say() { echo "$(date +%T) $*" | tee -a $OUT/progress.txt; }
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
for r in rows[:32]:
    print(f"{r['Name'][:84]:84s} calls/step={int(r['Calls'])/calls:5.1f} avg_us={float(r['AverageNs'])/1e3:8.1f} ms/step={float(r['TotalDurationNs'])/calls/1e6:6.3f}")
PY
}
timeline() { # $1 = name, rest = bench flags
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT/t_$name -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --steps 30 "$@" > /dev/null 2>&1 )
  python3 scripts/timeline.py $(find $OUT/t_$name -name "*kernel_trace.csv" | head -1) > $OUT/timeline_$name.txt 2>&1
  rm -rf $OUT/t_$name
  say "timeline $name: $(head -1 $OUT/timeline_$name.txt)"
}
for part in $PARTS; do case $part in
bench)
  python3 bench.py --details $OUT/bench_details.json > $OUT/bench.json 2> $OUT/bench.err || tail -5 $OUT/bench.err
  python3 bench.py --steps 20 --warmup 5 --details $OUT/bench_driver_protocol_details.json > $OUT/bench_driver_protocol.json 2> /dev/null
  say "bench: $(python3 scripts/bench_brief.py $OUT/bench.json 2>&1 | head -3 | tr '\n' ' ')";;
stats)
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants > $OUT/stats_bench.log 2>&1 )
  summary $OUT/stats "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants  (the default headline: 1 M Gaussians, 1920x1080, torch activations, 8 cycling views; per-kernel averages are over the views)" > $OUT/kernel_summary.txt
  cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv; rm -rf $OUT/stats
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2 -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-variants --fixed-view --ext-activations > $OUT/stats2_bench.log 2>&1 )
  summary $OUT/stats2 "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-variants --fixed-view --ext-activations  (one fixed view, activations inside the kernels)" > $OUT/kernel_summary_fixed_ext.txt
  rm -rf $OUT/stats2
  say "stats: $(sed -n 3p $OUT/kernel_summary.txt) | fixed: $(sed -n 3p $OUT/kernel_summary_fixed_ext.txt)";;
pmc)
  bash scripts/pmc_run.sh $TAG/pmc_cyc > $OUT/pmc_cyc.log 2>&1 || true
  python3 scripts/pmc_traffic.py $OUT/pmc_cyc $(cat collab_splats_amd/_build_rev.txt 2>/dev/null || echo unknown) > $OUT/pmc_traffic_cycling.json 2> /dev/null || true
  bash scripts/pmc_run.sh $TAG/pmc --fixed-view --ext-activations > $OUT/pmc.log 2>&1 || true
  python3 scripts/pmc_traffic.py $OUT/pmc $(cat collab_splats_amd/_build_rev.txt 2>/dev/null || echo unknown) > $OUT/pmc_traffic_fixed_ext.json 2> /dev/null || true
  python3 scripts/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.txt 2>&1 || true
  rm -rf $OUT/pmc $OUT/pmc_cyc
  say "pmc: $(python3 -c "import json; d=json.load(open('$OUT/pmc_traffic_cycling.json')); print(d.get('blend_bwd'), d.get('blend_fwd'), d.get('valu_issue_frac'))" 2>&1)";;
timelines)
  timeline 1M_cycling_torch
  timeline 1M_fixed_ext --fixed-view --ext-activations
  timeline 5M_dnloss_fixed --gaussians 5000000 --dn-loss --fixed-view
  timeline 5M_dnloss_cycling --gaussians 5000000 --dn-loss
  MISPLAT_FRONT_ONLY=0 timeline 5M_dnloss_fixed_full_sort --gaussians 5000000 --dn-loss --fixed-view
  timeline 1M_dnloss_cycling --dn-loss;;
models)
  for n in 1000000 5000000; do for v in "" "--fixed-view"; do
    f=bench_$((n/1000000))M_dnloss$( [ -n "$v" ] && echo _fixed ); python3 bench.py --no-cpu-baseline --no-variants --dn-loss --gaussians $n $v > $OUT/$f.json 2> $OUT/$f.err
    say "$f: $(python3 -c "import json; d=json.load(open('$OUT/$f.json')); print(d['ms_per_step'], d['device_ms_median'], d['config']['graph_hit_rate'])" 2>&1)"
  done; done
  MISPLAT_FRONT_ONLY=0 python3 bench.py --no-cpu-baseline --no-variants --dn-loss --gaussians 5000000 --fixed-view > $OUT/bench_5M_dnloss_fixed_full_sort.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --no-variants --dn-loss --gaussians 5000000 --fixed-view --buckets > $OUT/bench_5M_dnloss_fixed_buckets.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --no-variants --dn-loss --gaussians 5000000 --fixed-view --buckets --rehearse-sparse > $OUT/bench_5M_dnloss_fixed_buckets_sparse_rehearsal.json 2> /dev/null
  # the configs[4] step with every collective of the N > 1 path issued through RCCL by a group of ONE rank (no scaling figure: a smoke)
  MISPLAT_FORCE_COLLECTIVES=1 python3 bench.py --no-cpu-baseline --no-variants --shared-grads --dn-loss --gaussians 5000000 --buckets > $OUT/bench_5M_shared_rccl_world1.json 2> $OUT/bench_5M_shared_rccl_world1.err
  say "5M full sort / buckets / sparse rehearsal: $(for f in full_sort buckets buckets_sparse_rehearsal; do python3 -c "import json; d=json.load(open('$OUT/bench_5M_dnloss_fixed_$f.json')); print(d['ms_per_step'], end=' ')"; done 2>&1)";;
small)
  python3 bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 > $OUT/bench_10k.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --gaussians 100000 > $OUT/bench_100k.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 --graphed --fixed-view > $OUT/bench_10k_graphed.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --gaussians 10000 --width 256 --height 256 --graphed > $OUT/bench_10k_graphed_cycling.json 2> /dev/null
  python3 bench.py --no-cpu-baseline --gaussians 100000 --graphed > $OUT/bench_100k_graphed_cycling.json 2> /dev/null
  say "10k / 100k / 10k graphed: $(for f in 10k 100k 10k_graphed; do python3 -c "import json; d=json.load(open('$OUT/bench_$f.json')); print(d['ms_per_step'], {k: v[0] for k, v in d['variants'].items()}, d['config'].get('graph_hit_rate'), end=' | ')"; done 2>&1)";;
features)
  python3 bench.py --features 13 --fixed-view --no-cpu-baseline > $OUT/bench_features.json 2> $OUT/bench_features.err
  timeline 1M_features_fixed --features 13 --fixed-view
  say "features: $(python3 -c "import json; d=json.load(open('$OUT/bench_features.json')); print(d['ms_per_step'], {k: v[0] for k, v in d['variants'].items()})" 2>&1)";;
esac; done
cat $OUT/progress.txt
