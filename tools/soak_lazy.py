#!/usr/bin/env python3
"""GPU box: a long run of the token-pooled step with the deferred decay against every-row-every-step -- N steps of
non-repeating batches over a Zipf vocabulary, losses compared every step, tables after flush() at the end (bit for bit)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_knowledge_graph_embeddings_amd import hotpath as H  # noqa: E402
from open_knowledge_graph_embeddings_amd.synthetic import make_token_matrix  # noqa: E402
from open_knowledge_graph_embeddings_amd.token_pooled import TokenPooledTrainStep, TokenSlot  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
n_ent, n_rel, d, B, N, L, vt_e, vt_r = 200_000, 5_000, 256, 1024, 2048, 10, 40_000, 4_000
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)          # noqa: E731
te, tr = t(make_token_matrix(rng, n_ent, vt_e, L)), t(make_token_matrix(rng, n_rel, vt_r, L))
We, Wr = (rng.standard_normal((vt_e, d), dtype=np.float32) * 0.1), (rng.standard_normal((vt_r, d), dtype=np.float32) * 0.1)


def make(window):
    e, r = TokenSlot(t(We), te, "sum", True), TokenSlot(t(Wr), tr, "sum", True)
    e.bn[:d], r.bn[:d] = 0.5, 0.5
    return TokenPooledTrainStep(e, r, "complex", lr=0.1, dropout=0.1, seed=1, decay_window=window), e, r


a, b = make(8), make(1)
mismatch = 0
t_a = t_b = 0.0
for i in range(steps):
    r2 = np.random.default_rng(1000 + i)
    rows = np.arange(B, dtype=np.int32)
    cols = np.sort(r2.integers(0, N, B)).astype(np.int32)
    order = np.argsort(cols, kind="stable")
    batch = H.PrefixBatch(po_rel=t(r2.integers(2, n_rel, B // 2).astype(np.int32)), po_obj=t(r2.integers(2, n_ent, B // 2).astype(np.int32)),
                          sp_subj=t(r2.integers(2, n_ent, B // 2).astype(np.int32)), sp_rel=t(r2.integers(2, n_rel, B // 2).astype(np.int32)),
                          pos_row=t(rows[order]), pos_col=t(cols[order]), cand_ids=t((r2.choice(n_ent - 2, N, replace=False) + 2).astype(np.int32)))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    la = a[0].step(batch)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    lb = b[0].step(batch)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t_a, t_b = t_a + (t1 - t0), t_b + (t2 - t1)
    mismatch += int(float(la[0]) != float(lb[0]))
lag = int(a[0]._counters[0]) - a[1].row_steps
a[0].flush()
torch.cuda.synchronize()
same = all(torch.equal(x.W, y.W) and torch.equal(x.sumW, y.sumW) and torch.equal(x.bn, y.bn) for x, y in ((a[1], b[1]), (a[2], b[2])))
print(json.dumps({"steps": steps, "loss_mismatches": mismatch, "tables_bit_equal_after_flush": bool(same), "max_lag_before_flush": int(lag.max()),
                  "rows_owing_before_flush": int((lag > 0).sum()), "ms_per_step_deferred": round(1e3 * t_a / steps, 4),
                  "ms_per_step_eager": round(1e3 * t_b / steps, 4)}))
