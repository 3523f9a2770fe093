#!/bin/bash
# round-4 helper (GPU box): run the given pytest targets under -m gpu, print the tail and the [hip ...] lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest "$@" -m gpu -x -q -s > $O/some.log 2>&1; echo "tests rc=$?" >> $O/some.log
grep "^\[hip\|^\[oracle" $O/some.log
tail -8 $O/some.log
