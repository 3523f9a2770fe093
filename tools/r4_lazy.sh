#!/bin/bash
# round-4 helper (GPU box): the deferred-decay tests, then configs[4] at decay windows 1 / 4 / 8 / 16
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_token_pooled.py tests/test_token_pooled_full_size.py -m gpu -x -q > $O/lazy_tests.log 2>&1; rc=$?
tail -15 $O/lazy_tests.log
[ $rc -ne 0 ] && exit 1
for w in ${WINDOWS:-1 8 4 16}; do
  OKGE_LAZY_DECAY=$w timeout -k 10 300 python tools/bench_configs.py S-OLP-tok > $O/lazy_w$w.json 2> $O/lazy_w$w.err || { tail -5 $O/lazy_w$w.err; exit 1; }
  echo "== window $w"; cat $O/lazy_w$w.json
done
