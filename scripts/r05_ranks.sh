#!/bin/bash
# two ranks on the one card (gloo: a rehearsal of the N > 1 plumbing of bench.py, not a measurement)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ranks; mkdir -p $O
export MISPLAT_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 8 --warmup 4 --gaussians 300000 --no-cpu-baseline > $O/indep.json 2> $O/indep.err; echo "rc $?"; tail -1 $O/indep.json | cut -c1-400
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 8 --warmup 4 --gaussians 300000 --shared-grads --dn-loss --no-cpu-baseline > $O/shared.json 2> $O/shared.err; echo "rc $?"; tail -1 $O/shared.json | cut -c1-600; tail -3 $O/shared.err
