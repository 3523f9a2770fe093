#!/bin/bash
# round-4 helper (GPU box): the partial-slab reduction beside the dQ kernel (OKGE_REDUCE_OVERLAP) -- tests, then A/B on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/ovl_tests.log 2>&1; rc=$?
tail -4 $O/ovl_tests.log
[ $rc -ne 0 ] && exit 1
for v in 1 0 1 0; do
  echo "== OKGE_REDUCE_OVERLAP=$v"
  OKGE_REDUCE_OVERLAP=$v timeout -k 10 300 python tools/bench_configs.py S-DM S-OLP-tok-short S-FB-rank8 2>/dev/null | cut -c1-420
done
