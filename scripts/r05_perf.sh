#!/bin/bash
# GPU: kernel times of variant libraries (collab_splats_amd/_exp/libmisplat_<tag>.so; "main" = the shipped one), cycling + fixed
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-perf}; shift; mkdir -p $O
B="--no-cpu-baseline --no-variants --no-live-pmc"
for tag in "$@"; do
  L=$PWD/collab_splats_amd/_exp/libmisplat_$tag.so; [ "$tag" = main ] && L=$PWD/collab_splats_amd/libmisplat.so
  MISPLAT_LIB=$L timeout -k 10 300 python bench.py $B > $O/$tag.json 2> $O/$tag.err
  MISPLAT_LIB=$L timeout -k 10 300 python bench.py $B --fixed-view > $O/${tag}_fixed.json 2> $O/${tag}_fixed.err
done
python - "$O" "$@" <<'PY'
import json, sys
O=sys.argv[1]
for t in sys.argv[2:]:
    for sfx in ("", "_fixed"):
        d=json.loads(open(f"{O}/{t}{sfx}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernel_ms"]
        print(f"{t+sfx:24s} step {d['ms_per_step']:.4f}  fwd {k['blend_fwd']:.4f}  bwd {k['blend_bwd']:.4f}")
PY
