#!/bin/bash
# round-4 helper (GPU box): configs[4] with the lazy update kernel under three register budgets, same box
# (the 6- and 8-wave instances it selected with OKGE_LAZY_WAVES were removed after this measurement: profiles/round4_ablation.md section 6;
#  what remains useful is the deferred-vs-eager A/B at the end)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
for w in 5 6 8; do
  OKGE_LAZY_WAVES=$w timeout -k 10 300 python tools/bench_configs.py S-OLP-tok > $O/lazy_occ$w.json 2> $O/lazy_occ$w.err || { tail -5 $O/lazy_occ$w.err; exit 1; }
  echo "== waves $w"; head -2 $O/lazy_occ$w.json
done
OKGE_LAZY_DECAY=1 timeout -k 10 300 python tools/bench_configs.py S-OLP-tok > $O/lazy_occ_eager.json 2>/dev/null; echo "== eager"; head -2 $O/lazy_occ_eager.json
