#!/bin/bash
# round-4 helper (GPU box): token-pooled tests, then S-OLP-tok with the scatter plan and with atomics, each under rocprofv3 --stats
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_token_pooled.py tests/test_token_pooled_full_size.py -x -q > $O/t.log 2>&1; echo "tests rc=$?" >> $O/t.log
tail -4 $O/t.log
cd /tmp && export TMPDIR=/tmp
for V in plan atomics; do
  rm -rf /tmp/tr_$V
  OKGE_POOL_SCATTER=$V timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$V -- python3 $R/tools/bench_configs.py S-OLP-tok > $O/tok_$V.txt 2> $O/tok_$V.err || exit 1
  find /tmp/tr_$V -name "*kernel_stats.csv" -exec cp {} $O/tok_${V}_kernel_stats.csv \;
  grep -v amdgpu.ids $O/tok_$V.txt
done
