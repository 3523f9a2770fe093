#!/bin/bash
# round-4 helper (GPU box): parameter sweep of the deferred-decay kernels at configs[4]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/bench_configs.py S-OLP-tok-short 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_us']; print(d['ms_per_step'], 'catch', k.get('pool_catch_up'), 'adagrad', k['adagrad'], 'encode', k['pool_encode'])"; }
run OKGE_LAZY_BATCH=16
run OKGE_LAZY_BATCH=8
run OKGE_LAZY_BATCH=32
run OKGE_LAZY_BATCH=64
run OKGE_CATCH_PAIRS=8
run OKGE_CATCH_PAIRS=32
run OKGE_CATCH_ROWS=4
run OKGE_CATCH_ROWS=16 OKGE_CATCH_PAIRS=32
run OKGE_LAZY_DECAY=4
run OKGE_LAZY_DECAY=16
run OKGE_LAZY_DECAY=32
run OKGE_LAZY_BATCH=16
