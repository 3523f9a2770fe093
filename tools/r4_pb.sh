#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not token_pooled" > $O/pb_tests.log 2>&1; rc=$?
tail -5 $O/pb_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_configs.py S-DM S-FB-kl > $O/pb_cfg.json 2>&1; cat $O/pb_cfg.json
