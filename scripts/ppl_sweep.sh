#!/bin/bash
for f in 1 2 4; do for b in 1 2 4; do
  echo -n "ppl_fwd=$f ppl_bwd=$b: "
  MISPLAT_PPL_FWD=$f MISPLAT_PPL_BWD=$b python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
