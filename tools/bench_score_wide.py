#!/usr/bin/env python3
"""ms per okge_score_prefixes call at the DistMult d = 512 evaluation shape (B = 512, N = 14 541): the register-tile score sweep
(default) against the 32 x 32 cut (OKGE_TILE_W=32)."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from open_knowledge_graph_embeddings_amd import hotpath as H
hp = H.HotPath("cuda:0")
n_ent, d, B = 14543, 512, 512
E = torch.randn((n_ent, d), device="cuda") * 0.1
R = torch.randn((239, d), device="cuda") * 0.1
t = lambda lo, hi, n: torch.randint(lo, hi, (n,), device="cuda", dtype=torch.int32)
batch = H.PrefixBatch(po_rel=t(2, 239, B // 2), po_obj=t(2, n_ent, B // 2), sp_subj=t(2, n_ent, B // 2), sp_rel=t(2, 239, B // 2), cand_first=2, n_cand=n_ent - 2)
out = torch.empty((B, (n_ent - 2 + 3) // 4 * 4), device="cuda")[:, :n_ent - 2]
for _ in range(5): hp.score(E, R, "distmult", batch, out=out)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): hp.score(E, R, "distmult", batch, out=out)
torch.cuda.synchronize(); print("OKGE_TILE_W", os.environ.get("OKGE_TILE_W"), "score ms", (time.perf_counter() - t0) / 50 * 1e3)
