#!/bin/bash
# round 5, first GPU call: RCCL world-of-one test, atomic-side diagnostics of blend_bwd, forced-collective configs[4] bench
set -e -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "rccl_world_of_one or sparse_reduce_without or gradient_sink" > $O/rccl_test.txt 2>&1 || { tail -50 $O/rccl_test.txt; exit 1; }
tail -3 $O/rccl_test.txt
B="--no-cpu-baseline --no-variants --no-live-pmc"
timeout -k 10 300 python bench.py $B > $O/bwd_base.json 2> $O/bwd_base.err
MISPLAT_LIB=$PWD/collab_splats_amd/_exp/libmisplat_noatomic.so timeout -k 10 300 python bench.py $B > $O/bwd_noatomic.json 2> $O/bwd_noatomic.err
MISPLAT_LIB=$PWD/collab_splats_amd/_exp/libmisplat_rowstores.so timeout -k 10 300 python bench.py $B > $O/bwd_rowstores.json 2> $O/bwd_rowstores.err
timeout -k 10 300 python bench.py $B --fixed-view > $O/bwd_base_fixed.json 2> $O/bwd_base_fixed.err
MISPLAT_LIB=$PWD/collab_splats_amd/_exp/libmisplat_noatomic.so timeout -k 10 300 python bench.py $B --fixed-view > $O/bwd_noatomic_fixed.json 2> $O/bwd_noatomic_fixed.err
python - <<'PY'
import json
for t in ("base","noatomic","rowstores","base_fixed","noatomic_fixed"):
    d=json.loads(open(f"gpurun_out/r05a/bwd_{t}.json").read().strip().splitlines()[-1])
    print(t, d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
MISPLAT_FORCE_COLLECTIVES=1 NCCL_DEBUG=WARN timeout -k 10 400 python bench.py --shared-grads --dn-loss --gaussians 5000000 --buckets --no-cpu-baseline > $O/forced_5M.json 2> $O/forced_5M.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05a/forced_5M.json").read().strip().splitlines()[-1])
print("forced 5M", d["ms_per_step"], d.get("allreduce_ms"), d["config"]["parallelism"][:300])
PY
