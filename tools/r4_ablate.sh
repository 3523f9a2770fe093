#!/bin/bash
# round-4 diagnostics (GPU box): S-OLP-tok under rocprofv3 --stats with parts of the scatter plan left out (OKGE_SC_ABLATE bits)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for A in "$@"; do
  rm -rf /tmp/tr_a
  OKGE_SC_ABLATE=$A timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_a -- python3 $R/tools/bench_configs.py S-OLP-tok-short > /dev/null 2> $O/abl_$A.err || exit 1
  find /tmp/tr_a -name "*kernel_stats.csv" -exec cp {} $O/abl_${A}_kernel_stats.csv \;
  echo "ablate=$A"; python3 $R/tools/kstats.py $O/abl_${A}_kernel_stats.csv | grep "pool_\|bn_\|adagrad"
done
