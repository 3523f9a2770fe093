"""Seeded synthetic workloads of the shapes BASELINE.json / SURVEY.md section 8d name (no dataset download:
FB15k-237's train split is absent from the reference snapshot and there is no network)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class Workload:
    name: str
    scorer: str
    n_ent: int
    n_rel: int
    d: int
    n_po: int
    n_sp: int
    n_cand: int           # 0 => 1-vs-all (ids 2 .. n_ent-1)
    input_dropout: float
    init_std: float
    lr: float
    loss: str = "bce"

    @property
    def B(self):
        return self.n_po + self.n_sp

    @property
    def N(self):
        return self.n_cand if self.n_cand else self.n_ent - 2


WORKLOADS = {
    # configs[1]: FB15k-237 LookupComplexRelationModel d=200, 1-vs-all (config/fb15k237/fb15k237-complex-kge.yaml)
    "S-FB": Workload("S-FB", "complex", 14543, 239, 200, 256, 256, 0, 0.4, 0.1, 0.3),
    # configs[0]: same, d=64, batch 128 (CPU plumbing case)
    "S-FB-plumb": Workload("S-FB-plumb", "complex", 14543, 239, 64, 64, 64, 0, 0.4, 0.1, 0.3),
    # configs[3] scaled per GPU: OLPBENCH-shaped, ComplEx d=256, batch 4096
    "S-OLP": Workload("S-OLP", "complex", 2_500_000, 100_000, 256, 2048, 2048, 0, 0.0, 0.1, 0.1),
}


def make_tables(w: Workload, seed=1234):
    rng = np.random.default_rng(seed)
    E = (rng.standard_normal((w.n_ent, w.d), dtype=np.float32) * np.float32(w.init_std))
    R = (rng.standard_normal((w.n_rel, w.d), dtype=np.float32) * np.float32(w.init_std))
    return E, R


def make_batch(w: Workload, seed, zipf=False):
    """Host-side batch: prefix ids uniform over [2, n) (Zipf(1.1) for OLP shapes), positives per row
    1 + Geometric(0.55) capped at 64, unique columns (SURVEY.md section 8d 'S-FB').
    Returns dict(po_rel, po_obj, sp_subj, sp_rel, pos_row, pos_col [sorted by col], n_pos)."""
    rng = np.random.default_rng(seed)
    N = w.N

    def ent_ids(n):
        if zipf:
            return (2 + (rng.zipf(1.1, n) - 1) % (w.n_ent - 2)).astype(np.int32)
        return rng.integers(2, w.n_ent, n).astype(np.int32)

    out = {
        "po_rel": rng.integers(2, w.n_rel, w.n_po).astype(np.int32), "po_obj": ent_ids(w.n_po),
        "sp_subj": ent_ids(w.n_sp), "sp_rel": rng.integers(2, w.n_rel, w.n_sp).astype(np.int32),
    }
    counts = np.minimum(rng.geometric(0.55, w.B), 64)      # geometric >= 1  ==  1 + Geometric0
    rows = np.repeat(np.arange(w.B, dtype=np.int64), counts)
    cols = rng.integers(0, N, rows.shape[0]).astype(np.int64)
    key = np.unique(cols * w.B + rows)                       # dedupe (labels are sets), sorted by (col, row)
    out["pos_col"] = (key // w.B).astype(np.int32)
    out["pos_row"] = (key % w.B).astype(np.int32)
    out["n_pos"] = int(key.shape[0])
    return out


def dense_labels(batch, B, N):
    y = np.zeros((B, N), np.float32)
    y[batch["pos_row"], batch["pos_col"]] = 1.0
    return y


def make_eval_batch(w: Workload, seed):
    """Evaluation batch: like make_batch, plus answer groups (one mention group per positive; FB15k-237 has
    single-mention entities) and an all-splits filter = the positives plus ~2 answers from other splits per row
    (dataset.py:520-565, :921-927).  CSR arrays: row_ptr/grp_ptr/ids for groups, filt_ptr/filt_col for filters."""
    b = make_batch(w, seed)
    rng = np.random.default_rng(seed + 99)
    order = np.lexsort((b["pos_col"], b["pos_row"]))                   # by row, then column
    rows, cols = b["pos_row"][order].astype(np.int64), b["pos_col"][order].astype(np.int64)
    counts = np.bincount(rows, minlength=w.B)
    b["row_ptr"] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    b["grp_ptr"] = np.arange(len(cols) + 1, dtype=np.int64)            # one id per group
    b["ids"] = cols.astype(np.int32)
    extra_rows = np.repeat(np.arange(w.B, dtype=np.int64), 2)
    extra_cols = rng.integers(0, w.N, extra_rows.shape[0]).astype(np.int64)
    key = np.unique(np.concatenate([rows, extra_rows]) * w.N + np.concatenate([cols, extra_cols]))
    frow, fcol = key // w.N, key % w.N
    b["filt_ptr"] = np.concatenate([[0], np.cumsum(np.bincount(frow, minlength=w.B))]).astype(np.int64)
    b["filt_col"] = fcol.astype(np.int32)
    return b


def make_token_matrix(rng, n, vocab, max_len=10, zipf_a=1.2):
    """(n, max_len) int32 token-id rows of the S-OLP-tok workload (model.py:579-586 layout):
    BOS=2, 1 + Poisson(2) body tokens from a Zipf vocabulary (ids 4 .. vocab-1; the reference's token ids are
    frequency-ranked), EOS=3, right-padded with 0."""
    lens = np.minimum(1 + rng.poisson(2, n), max_len - 2)
    m = np.zeros((n, max_len), np.int32)
    m[:, 0] = 2
    body = (4 + (rng.zipf(zipf_a, (n, max_len)) - 1) % (vocab - 4)).astype(np.int32)
    for j in range(1, max_len):
        m[:, j] = np.where(j <= lens, body[:, j], np.where(j == lens + 1, 3, 0))
    return m
