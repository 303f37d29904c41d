#!/bin/bash
# GPU: the soak runs of the round on the current build -> gpurun_out/soak5/
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak5; mkdir -p $O
timeout -k 10 400 python scripts/soak.py 1000000 400 > $O/soak.json 2> $O/soak.err && \
timeout -k 10 400 python scripts/soak_model.py 400000 300 100 > $O/soak_model.json 2> $O/soak_model.err && \
SOAK_FEATURES=13 timeout -k 10 300 python scripts/soak.py 1000000 250 > $O/soak_features.json 2> $O/soak_features.err
for f in soak soak_model soak_features; do echo "== $f"; tail -c 600 $O/$f.json; echo; tail -2 $O/$f.err; done
