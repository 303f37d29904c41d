#!/bin/bash
# usage (through gpurun): bash scripts/r3_more.sh <tag>   -- the other bench configurations (bench lines only)
TAG=${1:-more}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; return; }
  python - <<PY
import json
d = json.load(open("$OUT/$name.json")); k = d["roofline"]["kernel_ms"]
print(f"$name: {d['ms_per_step']:.4f} ms/step dev {d['device_ms_median']:.4f} value {d['value']:.1f} fwd {k.get('blend_fwd',0):.3f} bwd {k.get('blend_bwd',0):.3f} B {k.get('raster_fwd_B',0):.3f} variants", {a: b['ms_per_step'] for a, b in d.get('variants', {}).items()})
PY
}
run bench_1M_dnloss --dn-loss
run bench_1M_dnloss_fixed --dn-loss --fixed-view
run bench_5M_dnloss --dn-loss --gaussians 5000000
run bench_5M_dnloss_fixed --dn-loss --gaussians 5000000 --fixed-view
run bench_100k --gaussians 100000 --no-variants
run bench_10k --gaussians 10000 --width 256 --height 256 --no-variants
