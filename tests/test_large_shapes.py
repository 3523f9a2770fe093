"""BASELINE configs[3] at its real size: |E| = 2.5 M (and one of eight row shards, 312 500 candidates), |R| = 100 k,
d = 256, B = 4096 -- B x N = 1.0e10 score elements, beyond 2^32, so every size_t index of the tile / dQ / score
kernels and the candidate-range sweep of the training workspace (okge_api.hip: make_geometry) is exercised.

Nothing of that size can be recomputed densely on the host, so parity is SAMPLED against the NumPy oracle in float64:
  scores  : whole rows of `all_outputs` for a few batch rows + a random block of columns for all rows
  loss    : BCE over the (validated) HIP scores, summed in float64 with plain torch ops on the device
  dE      : rows of ~500 sampled candidates (needs only X[:, sample]) incl. the first / last rows and range borders
  dQ/dR/dE: for batch rows whose entity and relation occur once in the batch, the complete chain
            X[row, :] -> G[row, :] -> dQ[row] -> (de, dr), swept over all candidates in chunks
  ranks   : bit-exact against the oracle's rule on the same score rows
Dropout is ON (p = 0.4 on entities, 0.2 on relations): the Philox row keys are global candidate positions."""
import os

import numpy as np
import pytest
import torch

from oracle import kge_oracle as ko

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("production_config")]   # OKGE_VALIDATE off, like production

D, B, N_REL = 256, 4096, 100_000
P_ENT, P_REL, SEED, STEP = 0.4, 0.2, 77, 5


def _zipf_ids(rng, n, lo, hi):
    """Zipf(1.1)-distributed ids over [lo, hi) (S-OLP prefix entities)"""
    return (lo + (rng.zipf(1.1, n) - 1) % (hi - lo)).astype(np.int32)


def _case(n_ent, seed):
    from open_knowledge_graph_embeddings_amd import hotpath as H
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(seed)
    E = torch.randn((n_ent, D), device=dev, generator=gen) * 0.35          # scores of order 1: sigmoid off its linear part
    R = torch.randn((N_REL, D), device=dev, generator=gen) * 0.35
    rng = np.random.default_rng(seed)
    n_po = n_sp = B // 2
    N = n_ent - 2
    po_rel = rng.integers(2, N_REL, n_po).astype(np.int32)
    sp_rel = rng.integers(2, N_REL, n_sp).astype(np.int32)
    po_obj, sp_subj = _zipf_ids(rng, n_po, 2, n_ent), _zipf_ids(rng, n_sp, 2, n_ent)
    # one positive per row + extras on the first / last candidates and around multiples of 65536 (range borders)
    rows = np.arange(B)
    cols = rng.integers(0, N, B)
    extra_c = np.concatenate([[0, 1, N - 1, N - 2], (np.arange(1, 1 + N // 65536) * 65536)[:8] - 1,
                              (np.arange(1, 1 + N // 65536) * 65536)[:8]]).astype(np.int64)
    extra_c = extra_c[extra_c < N]
    extra_r = rng.integers(0, B, len(extra_c))
    coords = np.unique(np.stack([np.concatenate([cols, extra_c]), np.concatenate([rows, extra_r])], 1), axis=0)   # by col, row
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)       # noqa: E731
    batch = H.PrefixBatch(po_rel=t(po_rel), po_obj=t(po_obj), sp_subj=t(sp_subj), sp_rel=t(sp_rel),
                          pos_row=t(coords[:, 1].astype(np.int32)), pos_col=t(coords[:, 0].astype(np.int32)),
                          cand_first=2, n_cand=N)
    batch.drop_cand = H.DropoutSpec(P_ENT, SEED, H.STREAM_CAND, STEP)
    batch.drop_po_ent = H.DropoutSpec(P_ENT, SEED, H.STREAM_PO_ENT, STEP)
    batch.drop_sp_ent = H.DropoutSpec(P_ENT, SEED, H.STREAM_SP_ENT, STEP)
    batch.drop_po_rel = H.DropoutSpec(P_REL, SEED, H.STREAM_PO_REL, STEP)
    batch.drop_sp_rel = H.DropoutSpec(P_REL, SEED, H.STREAM_SP_REL, STEP)
    ids = dict(po_rel=po_rel, po_obj=po_obj, sp_subj=sp_subj, sp_rel=sp_rel, coords=coords)
    return E, R, batch, ids


def _queries(E, R, ids):
    """float64 folded queries of all B rows (oracle encode + prefix_query), and the pieces the chain rule needs"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    f64 = lambda x: x.cpu().numpy().astype(np.float64)                    # noqa: E731
    n_po, n_sp = len(ids["po_rel"]), len(ids["sp_rel"])
    Ed = lambda i: f64(E[torch.from_numpy(i.astype(np.int64)).to(E.device)])   # noqa: E731
    Rd = lambda i: f64(R[torch.from_numpy(i.astype(np.int64)).to(R.device)])   # noqa: E731
    km = lambda stream, n, p: ko.dropout_keep_mask(SEED, stream, STEP, n, D, p)   # noqa: E731
    k_po_e, k_sp_e = km(H.STREAM_PO_ENT, n_po, P_ENT), km(H.STREAM_SP_ENT, n_sp, P_ENT)
    k_po_r, k_sp_r = km(H.STREAM_PO_REL, n_po, P_REL), km(H.STREAM_SP_REL, n_sp, P_REL)
    o = Ed(ids["po_obj"]) * k_po_e / (1 - P_ENT)
    r_po = Rd(ids["po_rel"]) * k_po_r / (1 - P_REL)
    s = Ed(ids["sp_subj"]) * k_sp_e / (1 - P_ENT)
    r_sp = Rd(ids["sp_rel"]) * k_sp_r / (1 - P_REL)
    Q = np.concatenate([ko.prefix_query(ko.COMPLEX, ko.DIR_PO, o, r_po), ko.prefix_query(ko.COMPLEX, ko.DIR_SP, s, r_sp)], 0)
    ent = np.concatenate([o, s], 0)
    rel = np.concatenate([r_po, r_sp], 0)
    keep_e = np.concatenate([k_po_e, k_sp_e], 0) / (1 - P_ENT)
    keep_r = np.concatenate([k_po_r, k_sp_r], 0) / (1 - P_REL)
    return Q, ent, rel, keep_e, keep_r


def _cand_rows(E, cols):
    """float64 masked candidate rows C[cols] (candidate position = column; entity id = column + 2)"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    cols = np.asarray(cols, np.int64)
    rows = E[torch.from_numpy(cols + 2).to(E.device)].cpu().numpy().astype(np.float64)
    keep = ko.dropout_keep_mask(SEED, H.STREAM_CAND, STEP, len(cols), D, P_ENT, row_keys=cols.astype(np.uint32))
    return rows * keep / (1 - P_ENT), keep / (1 - P_ENT)


@pytest.mark.parametrize("label,n_ent", [("one of eight shards", 312_502), ("whole table", 2_500_002)])
def test_cfg4_size_sampled_parity(okge_lib, label, n_ent):
    from open_knowledge_graph_embeddings_amd import hotpath as H
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    hp = H.HotPath("cuda:0")
    E, R, batch, ids = _case(n_ent, seed=n_ent % 1000)
    N = n_ent - 2
    assert B * N > 2 ** 30
    ws_bytes = int(hp.lib.okge_train_workspace_bytes(B, N, D))
    assert ws_bytes < 4 << 30, f"training workspace {ws_bytes / 2**30:.1f} GiB: G^T is not chunked"
    dE, dR = torch.zeros_like(E), torch.zeros_like(R)
    scores = torch.empty((B, N), dtype=torch.float32, device=E.device)
    norm = float(B) * float(N)
    loss = hp.forward_backward(E, R, "complex", batch, dE, dR, normalizer=norm, scores=scores, grads_zero=True)
    torch.cuda.synchronize()
    rng = np.random.default_rng(1)
    Q, ent, rel, keep_e, keep_r = _queries(E, R, ids)
    coords = ids["coords"]

    # ---- scores: a block of random columns for ALL rows (incl. first / last columns) -----------------------------------
    cs = np.unique(np.concatenate([rng.integers(0, N, 500), [0, 1, N - 1, N - 2, 65535, 65536], ids["po_obj"][:8] - 2]))
    cs = cs[(cs >= 0) & (cs < N)]
    Cs, keep_cs = _cand_rows(E, cs)
    Xs = Q @ Cs.T                                                           # (B, len(cs)) float64
    got = scores[:, torch.from_numpy(cs).to(E.device)].cpu().numpy()
    assert np.abs(got - Xs).max() < 1e-4, np.abs(got - Xs).max()            # north-star bound (observed ~1e-5 at |x|~10)
    # ---- loss over all B x N elements from the HIP scores, float64 on the device --------------------------------------
    ref_loss = 0.0
    for lo in range(0, B, 256):
        x = scores[lo:lo + 256].double()
        ref_loss += float((torch.clamp(x, min=0) + torch.log1p(torch.exp(-x.abs()))).sum())
    pr, pc = torch.from_numpy(coords[:, 1]).to(E.device), torch.from_numpy(coords[:, 0]).to(E.device)
    ref_loss -= float(scores[pr, pc].double().sum())
    assert abs(float(loss[0]) - ref_loss) <= 2e-5 * abs(ref_loss), (float(loss[0]), ref_loss)
    # ---- dE of the sampled candidates: dC = G[:, cs]^T Q (masked), + prefix-row contributions checked further down -----
    Y = np.zeros_like(Xs)
    col_of = {c: j for j, c in enumerate(cs)}
    for c, r in coords:
        if c in col_of:
            Y[r, col_of[c]] = 1.0
    _, g = ko.loss_and_dscore(Xs, Y, ko.LOSS_BCE)
    dC = ((g / norm).T @ Q) * keep_cs
    prefix_ents = set(ids["po_obj"].tolist()) | set(ids["sp_subj"].tolist())
    plain = np.array([c + 2 not in prefix_ents for c in cs])               # candidates that are no batch prefix entity
    got_dE = dE[torch.from_numpy(cs + 2).to(E.device)].cpu().numpy()
    scale = np.abs(dC).max()
    assert np.abs(got_dE[plain] - dC[plain]).max() <= 3e-5 * scale, (np.abs(got_dE[plain] - dC[plain]).max(), scale)
    # ---- full chain for batch rows whose entity and relation are unique in the batch ------------------------------------
    ent_ids = np.concatenate([ids["po_obj"], ids["sp_subj"]])
    rel_ids = np.concatenate([ids["po_rel"], ids["sp_rel"]])
    uniq_e = np.isin(ent_ids, [e for e, n in zip(*np.unique(ent_ids, return_counts=True)) if n == 1])
    uniq_r = np.isin(rel_ids, [r for r, n in zip(*np.unique(rel_ids, return_counts=True)) if n == 1])
    cand_rows = np.flatnonzero(uniq_e & uniq_r)
    rs = np.unique(np.concatenate([cand_rows[:3], cand_rows[-3:], rng.choice(cand_rows, 6, replace=False)]))
    dq = np.zeros((len(rs), D))
    xrow_err = 0.0
    Yr = {r: coords[coords[:, 1] == r, 0] for r in rs}
    for lo in range(0, N, 131072):
        hi = min(N, lo + 131072)
        Cc, _ = _cand_rows(E, np.arange(lo, hi))
        X = Q[rs] @ Cc.T
        xrow_err = max(xrow_err, float(np.abs(scores[torch.from_numpy(rs).to(E.device), lo:hi].cpu().numpy() - X).max()))
        y = np.zeros_like(X)
        for i, r in enumerate(rs):
            m = Yr[r][(Yr[r] >= lo) & (Yr[r] < hi)] - lo
            y[i, m] = 1.0
        _, g = ko.loss_and_dscore(X, y, ko.LOSS_BCE)
        dq += (g / norm) @ Cc
    assert xrow_err < 1e-4, xrow_err                                        # whole score rows, every column
    got_dR = dR[torch.from_numpy(rel_ids[rs].astype(np.int64)).to(E.device)].cpu().numpy()
    got_dEp = dE[torch.from_numpy(ent_ids[rs].astype(np.int64)).to(E.device)].cpu().numpy()
    for i, r in enumerate(rs):
        direction = ko.DIR_PO if r < B // 2 else ko.DIR_SP
        de, dr = ko.prefix_query_backward(ko.COMPLEX, direction, ent[r:r + 1], rel[r:r + 1], dq[i:i + 1])
        de, dr = de[0] * keep_e[r], dr[0] * keep_r[r]
        assert np.abs(got_dR[i] - dr).max() <= 3e-5 * np.abs(dr).max() + 1e-30
        # the entity's row also holds its gradient as a CANDIDATE (column = id - 2)
        Cc, kc = _cand_rows(E, [ent_ids[r] - 2])
        xc = Q @ Cc.T
        yc = np.zeros_like(xc)
        yc[coords[coords[:, 0] == ent_ids[r] - 2, 1], 0] = 1.0
        _, gc = ko.loss_and_dscore(xc, yc, ko.LOSS_BCE)
        full = de + ((gc / norm).T @ Q)[0] * kc[0]
        assert np.abs(got_dEp[i] - full).max() <= 3e-5 * np.abs(full).max() + 1e-30
    # ---- ranks on sampled rows of the same score matrix: bit-exact rule -----------------------------------------------
    rr = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, 6)]))
    row_ptr, grp_ptr, gids, fptr, fcol = [0], [0], [], [0], []
    dense_f = np.zeros((len(rr), N), bool)
    k = 0
    for b in range(B):
        if k < len(rr) and rr[k] == b:
            for _ in range(int(rng.integers(1, 4))):
                gids.extend(rng.integers(0, N, int(rng.integers(1, 3))).tolist())
                grp_ptr.append(len(gids))
            f = np.unique(np.concatenate([rng.integers(0, N, 20), gids[grp_ptr[row_ptr[-1]]:]]))
            fcol.extend(f.tolist())
            dense_f[k, f] = True
            k += 1
        row_ptr.append(len(grp_ptr) - 1)
        fptr.append(len(fcol))
    t = lambda a, dt: torch.from_numpy(np.asarray(a, dt)).to(E.device)     # noqa: E731
    ranks = hp.filtered_ranks(scores, t(fptr, np.int64), t(fcol, np.int32), t(row_ptr, np.int64), t(grp_ptr, np.int64),
                              t(gids, np.int32)).cpu().numpy()
    sub = scores[torch.from_numpy(rr).to(E.device)].cpu().numpy()
    sub_ptr = np.concatenate([[0], np.cumsum(np.diff(row_ptr)[rr])])
    ref = ko.filtered_ranks(sub, dense_f, sub_ptr, np.asarray(grp_ptr, np.int64), np.asarray(gids, np.int32))
    np.testing.assert_array_equal(ranks, ref)


def test_candidate_ranges_match_single_range(okge_lib, monkeypatch):
    """the range sweep (forced small here) gives the single-launch result: same tiles, same order inside a tile; only
    the dQ slabs are summed range by range"""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    hp = H.HotPath("cuda:0")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    n_ent, n_rel, d, b = 3000, 40, 200, 160
    E = torch.from_numpy((rng.standard_normal((n_ent, d)) * 0.3).astype(np.float32)).to(dev)
    R = torch.from_numpy((rng.standard_normal((n_rel, d)) * 0.3).astype(np.float32)).to(dev)
    N = n_ent - 2
    coords = np.unique(np.stack([rng.integers(0, N, 3 * b), rng.integers(0, 2 * b, 3 * b)], 1), axis=0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)       # noqa: E731
    out = {}
    for kind in ("bce", "kl"):
        for mb in ("1024", "1"):
            monkeypatch.setenv("OKGE_GT_MBYTES", mb)                      # 1 MiB: 12 tiles = 768 candidates per range
            r2 = np.random.default_rng(4)
            batch = H.PrefixBatch(po_rel=t(r2.integers(2, n_rel, b).astype(np.int32)), po_obj=t(r2.integers(2, n_ent, b).astype(np.int32)),
                                  sp_subj=t(r2.integers(2, n_ent, b).astype(np.int32)), sp_rel=t(r2.integers(2, n_rel, b).astype(np.int32)),
                                  pos_row=t(coords[:, 1].astype(np.int32)), pos_col=t(coords[:, 0].astype(np.int32)),
                                  cand_first=2, n_cand=N)
            batch.drop_cand = H.DropoutSpec(0.3, 9, H.STREAM_CAND, 1)
            hp._ws, hp._ws_bytes = None, 0
            dE, dR = torch.zeros_like(E), torch.zeros_like(R)
            loss = hp.forward_backward(E, R, "complex", batch, dE, dR, loss=kind, grads_zero=True)
            torch.cuda.synchronize()
            out[(kind, mb)] = (float(loss[0]), dE.cpu().numpy(), dR.cpu().numpy(), hp._ws_bytes)
        one, many = out[(kind, "1024")], out[(kind, "1")]
        assert many[3] < one[3]
        assert abs(one[0] - many[0]) <= 1e-6 * abs(one[0])
        np.testing.assert_allclose(many[1], one[1], rtol=0, atol=3e-6 * np.abs(one[1]).max())
        np.testing.assert_allclose(many[2], one[2], rtol=0, atol=3e-6 * np.abs(one[2]).max())


@pytest.mark.parametrize("d,kind,ids", [(32, "bce", False), (200, "kl", False), (256, "bce", True), (200, "bce", True)])
def test_tail_split_matches_plain_launch(okge_lib, monkeypatch, d, kind, ids):
    """more candidate tiles than CUs: the tiles left over after whole rounds (275 tiles = 256 + 19 here) are launched apart
    with the batch rows split across workgroups (okge_api.hip, make_geometry "tail split").  Same loss and gradients as the
    plain launch (OKGE_TAIL_SPLIT=0), up to the order the tail's partial candidate gradients are added in; also with the
    candidate ranges forced to 256 tiles, where the last range is ALL tail, and with an explicit candidate id list."""
    from open_knowledge_graph_embeddings_amd import hotpath as H
    hp = H.HotPath("cuda:0")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(d + len(kind))
    N, n_rel, b = 64 * 274 + 37, 30, 128                                   # 275 tiles, the last one ragged; B = 256 rows
    n_ent = N + 2 + (500 if ids else 0)
    E = torch.from_numpy((rng.standard_normal((n_ent, d)) * 0.3).astype(np.float32)).to(dev)
    R = torch.from_numpy((rng.standard_normal((n_rel, d)) * 0.3).astype(np.float32)).to(dev)
    coords = np.unique(np.stack([rng.integers(0, N, 4 * b), rng.integers(0, 2 * b, 4 * b)], 1), axis=0)
    coords = np.concatenate([coords, [[N - 1, 0], [N - 1, 2 * b - 1], [64 * 256, 5]]]).astype(np.int64)   # positives inside the tail
    coords = np.unique(coords, axis=0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)       # noqa: E731
    cand_ids = t((2 + rng.permutation(n_ent - 2)[:N]).astype(np.int32)) if ids else None
    out = {}
    for split, mb in (("0", "1024"), ("1", "1024"), ("1", "16")):         # 16 MiB: ranges of 256 tiles at B = 256
        monkeypatch.setenv("OKGE_TAIL_SPLIT", split)
        monkeypatch.setenv("OKGE_GT_MBYTES", mb)
        r2 = np.random.default_rng(4)
        batch = H.PrefixBatch(po_rel=t(r2.integers(2, n_rel, b).astype(np.int32)), po_obj=t(r2.integers(2, n_ent, b).astype(np.int32)),
                              sp_subj=t(r2.integers(2, n_ent, b).astype(np.int32)), sp_rel=t(r2.integers(2, n_rel, b).astype(np.int32)),
                              pos_row=t(coords[:, 1].astype(np.int32)), pos_col=t(coords[:, 0].astype(np.int32)),
                              cand_first=2, n_cand=N, cand_ids=cand_ids)
        batch.drop_cand = H.DropoutSpec(0.3, 9, H.STREAM_CAND, 1)
        batch.drop_po_ent = H.DropoutSpec(0.2, 9, H.STREAM_PO_ENT, 1)
        hp._ws, hp._ws_bytes = None, 0
        dE, dR = torch.full_like(E, 7.0), torch.zeros_like(R)              # grads_zero on a contiguous range: rows are STORED
        if ids:
            dE.zero_()
        loss = hp.forward_backward(E, R, "complex", batch, dE, dR, loss=kind, grads_zero=not ids, label_smoothing=0.1 if kind == "bce" else 0.0)
        torch.cuda.synchronize()
        if not ids:
            assert float(dE[:2].min()) == 7.0                              # rows in front of the candidates: untouched
            dE[:2] = 0
        out[(split, mb)] = (float(loss[0]), dE.cpu().numpy(), dR.cpu().numpy())
    plain = out[("0", "1024")]
    for key in (("1", "1024"), ("1", "16")):
        got = out[key]
        assert abs(plain[0] - got[0]) <= 1e-6 * abs(plain[0]), key
        np.testing.assert_allclose(got[1], plain[1], rtol=0, atol=3e-6 * np.abs(plain[1]).max(), err_msg=str(key))
        np.testing.assert_allclose(got[2], plain[2], rtol=0, atol=3e-6 * np.abs(plain[2]).max(), err_msg=str(key))
    # candidate rows outside the tail come from the same tiles in both: bit-equal (rows that also receive a prefix gradient
    # through float atomics aside)
    if not ids:
        r2 = np.random.default_rng(4)
        _, po_obj, sp_subj, _ = (r2.integers(2, hi, b) for hi in (n_rel, n_ent, n_ent, n_rel))
        rows = np.setdiff1d(np.arange(2, 2 + 64 * 256), np.concatenate([po_obj, sp_subj]))
        np.testing.assert_array_equal(out[("1", "1024")][1][rows], plain[1][rows])
