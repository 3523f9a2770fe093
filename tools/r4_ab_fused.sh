#!/bin/bash
# round-4 A/B (GPU box): S-FB with the Adagrad update inside the step's launches (okge_train_step) against the two-call sequence
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for V in 1 0 1 0; do
  OKGE_FUSED_UPDATE=$V OKGE_BENCH_OLP=0 OKGE_BENCH_CONFIGS=0 OKGE_BENCH_DROPIN=0 timeout -k 10 300 python bench.py --steps 2000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused=$V', round(d['ms_per_step'],5), round(d['ms_per_step_min'],5), {k: round(v,1) for k,v in d['roofline']['kernels_us'].items()})"
done
