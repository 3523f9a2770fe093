"""CPU ORACLE for the open-KGE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``open_knowledge_graph_embeddings_amd``) never does and fails loudly when its
HIP library is missing.

What this is: a NumPy restatement (explicit forward AND hand-written backward, no autograd) of the
reference's batched prefix-scoring training step and filtered-rank evaluation.  Every function cites
the reference lines it follows (paths relative to /root/reference).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against golden
vectors produced by running the reference itself in the build container
(``tests/golden/make_golden.py``; vectors G1, G2, G3, G5, G7 of SURVEY.md section 8c).
"""
from __future__ import annotations

import numpy as np

COMPLEX, DISTMULT = 0, 1
DIR_PO, DIR_SP = 0, 1
LOSS_BCE, LOSS_KL = 0, 1

KIND_NAMES = {"complex": COMPLEX, "distmult": DISTMULT}
LOSS_NAMES = {"bce": LOSS_BCE, "kl": LOSS_KL}


# ------------------------------------------------------------------------------------------------
# embedder: gather -> dropout            openkge/model.py:455-480 (_encode), :502-510 (encode_obj)
# ------------------------------------------------------------------------------------------------
def encode(table, ids, keep=None, p=0.0):
    """``embedding(ids)`` followed by ``dropout(p)``: rows * keep / (1 - p).

    model.py:457-462.  ``keep`` is the Bernoulli keep-mask (bool/uint8, same shape as the rows) or
    None for p == 0 / eval mode.  torch's CPU dropout multiplies by ``noise = bernoulli/(1-p)``.
    """
    rows = table[np.asarray(ids).reshape(-1).astype(np.int64)]
    if keep is not None and p > 0.0:
        scale = table.dtype.type(1.0) / table.dtype.type(1.0 - p)
        rows = rows * (keep.astype(table.dtype) * scale)
    return rows


# ------------------------------------------------------------------------------------------------
# scorers: prefix branch                 openkge/model.py:198-229 (ComplEx), :268-274 (DistMult)
# ------------------------------------------------------------------------------------------------
def prefix_query(kind, direction, ent, rel):
    """Fold the two prefix slots into ONE query row so that score = query . candidate.

    ComplEx sp (model.py:205-210):  (s1 r1) O1 + (s2 r1) O2 + (s1 r2) O2 - (s2 r2) O1
        = [s1 r1 - s2 r2 , s2 r1 + s1 r2] . [O1, O2]
    ComplEx po (model.py:211-216):  (o1 r1) S1 + (o2 r1) S2 + (o2 r2) S1 - (o1 r2) S2
        = [o1 r1 + o2 r2 , o2 r1 - o1 r2] . [S1, S2]
    DistMult (model.py:269-272): (s * r) . O   /  (r * o) . S
    First half of the embedding is the real part, second half the imaginary part (model.py:202-204).
    """
    if kind == DISTMULT:
        return ent * rel
    h = ent.shape[1] // 2
    e1, e2 = ent[:, :h], ent[:, h:]
    r1, r2 = rel[:, :h], rel[:, h:]
    if direction == DIR_SP:
        return np.concatenate([e1 * r1 - e2 * r2, e2 * r1 + e1 * r2], axis=1)
    return np.concatenate([e1 * r1 + e2 * r2, e2 * r1 - e1 * r2], axis=1)


def prefix_query_backward(kind, direction, ent, rel, dq):
    """Chain rule of :func:`prefix_query` (what autograd does for model.py:205-216 / :269-272)."""
    if kind == DISTMULT:
        return dq * rel, dq * ent
    h = ent.shape[1] // 2
    e1, e2 = ent[:, :h], ent[:, h:]
    r1, r2 = rel[:, :h], rel[:, h:]
    q1, q2 = dq[:, :h], dq[:, h:]
    if direction == DIR_SP:
        de = np.concatenate([q1 * r1 + q2 * r2, -q1 * r2 + q2 * r1], axis=1)
        dr = np.concatenate([q1 * e1 + q2 * e2, -q1 * e2 + q2 * e1], axis=1)
    else:
        de = np.concatenate([q1 * r1 - q2 * r2, q1 * r2 + q2 * r1], axis=1)
        dr = np.concatenate([q1 * e1 + q2 * e2, q1 * e2 - q2 * e1], axis=1)
    return de, dr


def score_prefix_4mm(kind, direction, ent, rel, cand):
    """The reference's literal op sequence (4 matrix products for ComplEx), model.py:205-216."""
    if kind == DISTMULT:
        return (ent * rel) @ cand.T
    h = ent.shape[1] // 2
    a1, a2 = ent[:, :h], ent[:, h:]
    r1, r2 = rel[:, :h], rel[:, h:]
    c1, c2 = cand[:, :h], cand[:, h:]
    if direction == DIR_SP:
        return (a1 * r1) @ c1.T + (a2 * r1) @ c2.T + (a1 * r2) @ c2.T - (a2 * r2) @ c1.T
    return (a1 * r1) @ c1.T + (a2 * r1) @ c2.T + (a2 * r2) @ c1.T - (a1 * r2) @ c2.T


def score_triples(kind, subj, rel, obj):
    """(b,) scores of b encoded triples, Hadamard form (model.py:231-238 ComplEx, :276 DistMult):
    ComplEx  sum s1*r1*o1 + s2*r1*o2 + s1*r2*o2 - s2*r2*o1,  DistMult  sum s*r*o."""
    if kind == DISTMULT:
        return (subj * obj * rel).sum(axis=1)
    h = subj.shape[1] // 2
    s1, s2, r1, r2, o1, o2 = subj[:, :h], subj[:, h:], rel[:, :h], rel[:, h:], obj[:, :h], obj[:, h:]
    return (s1 * o1 * r1 + s2 * o2 * r1 + s1 * o2 * r2 - s2 * o1 * r2).sum(axis=1)


def score_prefix(kind, direction, ent, rel, cand):
    """(b, N) scores of b prefixes against N candidate rows = query @ cand.T."""
    return prefix_query(kind, direction, ent, rel) @ cand.T


# ------------------------------------------------------------------------------------------------
# loss                                    openkge/trainer.py:93-106
# ------------------------------------------------------------------------------------------------
def smooth_labels(y, smoothing):
    """trainer.py:103-105: labels = (labels + 1/N) * (1 - eps)."""
    if smoothing > 0:
        y = (y + y.dtype.type(1.0 / y.shape[-1])) * y.dtype.type(1.0 - smoothing)
    return y


def log_softmax(x):
    m = x.max(axis=1, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


def loss_and_dscore(x, y, loss_kind, smoothing=0.0):
    """Summed loss and d(loss)/d(score).

    bce (trainer.py:102-106, BCEWithLogitsLoss(sum)): sum max(x,0) - x*y + log1p(exp(-|x|)),
        gradient sigmoid(x) - y, labels smoothed first.
    kl  (trainer.py:99-100,106, KLDivLoss(sum) on log_softmax): sum xlogy(y,y) - y*log_softmax(x)
        with UNnormalised y (the normalisation is commented out, trainer.py:101),
        gradient softmax(x) * sum_n y - y.
    """
    if loss_kind == LOSS_BCE:
        y = smooth_labels(y, smoothing)
        loss = np.maximum(x, 0) - x * y + np.log1p(np.exp(-np.abs(x)))
        sig = 1.0 / (1.0 + np.exp(-x))
        return loss.sum(dtype=np.float64), (sig - y).astype(x.dtype)
    lsm = log_softmax(x)
    with np.errstate(divide="ignore", invalid="ignore"):
        ylogy = np.where(y > 0, y * np.log(np.where(y > 0, y, 1.0)), 0.0)
    loss = (ylogy - y * lsm).sum(dtype=np.float64)
    g = np.exp(lsm) * y.sum(axis=1, keepdims=True) - y
    return loss, g.astype(x.dtype)


# ------------------------------------------------------------------------------------------------
# one training step: forward + loss + backward
#   openkge/trainer.py:48-113 (AddLossModule.forward) and :217-234 (normalise, backward)
# ------------------------------------------------------------------------------------------------
def step_forward_backward(kind, E, R, po, sp, cand_ids, labels, loss_kind=LOSS_BCE, smoothing=0.0,
                          normalizer=None, p_ent=0.0, p_rel=0.0, keep_cand=None, keep_po_ent=None,
                          keep_sp_ent=None, keep_po_rel=None, keep_sp_rel=None, want_grads=True):
    """Returns dict(loss, outputs[B,N], dE, dR).

    po = (rel_ids, obj_ids) or None, sp = (subj_ids, rel_ids) or None  (trainer.py:69-71);
    output/label rows are ordered po first, then sp (trainer.py:91).  Candidates are encoded once
    and shared by both directions (trainer.py:75-87), including their dropout mask.
    normalizer defaults to B*N (dataset.py:935); the gradient is d(loss/normalizer).
    """
    dt = E.dtype
    cand_ids = np.asarray(cand_ids).reshape(-1).astype(np.int64)
    C = encode(E, cand_ids, keep_cand, p_ent)
    parts = []
    if po is not None:
        rel_ids, obj_ids = po
        r = encode(R, rel_ids, keep_po_rel, p_rel)
        o = encode(E, obj_ids, keep_po_ent, p_ent)
        parts.append((DIR_PO, o, r, np.asarray(obj_ids).reshape(-1).astype(np.int64),
                      np.asarray(rel_ids).reshape(-1).astype(np.int64), keep_po_ent, keep_po_rel))
    if sp is not None:
        subj_ids, rel_ids = sp
        s = encode(E, subj_ids, keep_sp_ent, p_ent)
        r = encode(R, rel_ids, keep_sp_rel, p_rel)
        parts.append((DIR_SP, s, r, np.asarray(subj_ids).reshape(-1).astype(np.int64),
                      np.asarray(rel_ids).reshape(-1).astype(np.int64), keep_sp_ent, keep_sp_rel))
    Q = np.concatenate([prefix_query(kind, d, e, r) for (d, e, r, *_rest) in parts], axis=0)
    X = Q @ C.T
    B, N = X.shape
    if normalizer is None:
        normalizer = float(B * N)
    loss, g = loss_and_dscore(X, labels.astype(dt), loss_kind, smoothing)
    out = {"loss": float(loss), "outputs": X}
    if not want_grads:
        return out
    G = (g / dt.type(normalizer)).astype(dt)
    dQ = G @ C
    dC = G.T @ Q
    dE = np.zeros_like(E)
    dR = np.zeros_like(R)
    sc_e = dt.type(1.0) / dt.type(1.0 - p_ent) if p_ent > 0 else dt.type(1.0)
    sc_r = dt.type(1.0) / dt.type(1.0 - p_rel) if p_rel > 0 else dt.type(1.0)
    if keep_cand is not None and p_ent > 0:
        dC = dC * (keep_cand.astype(dt) * sc_e)
    np.add.at(dE, cand_ids, dC)                       # embedding_dense_backward
    row = 0
    for (d, e, r, e_ids, r_ids, keep_e, keep_r) in parts:
        n = e.shape[0]
        de, dr = prefix_query_backward(kind, d, e, r, dQ[row:row + n])
        if keep_e is not None and p_ent > 0:
            de = de * (keep_e.astype(dt) * sc_e)
        if keep_r is not None and p_rel > 0:
            dr = dr * (keep_r.astype(dt) * sc_r)
        np.add.at(dE, e_ids, de)
        np.add.at(dR, r_ids, dr)
        row += n
    # padding_idx=PAD=0 rows never receive gradient (model.py:390-391); ids are >= 2 on this path.
    out.update(dE=dE, dR=dR, dQ=dQ, G=G, Q=Q)
    return out


# ------------------------------------------------------------------------------------------------
# dense Adagrad as configured by OptimRegime    utils/optim.py:29,139-160 + torch.optim.Adagrad
# ------------------------------------------------------------------------------------------------
def adagrad_step(p, g, state_sum, lr, weight_decay=1e-10, eps=1e-8):
    """In place: g += wd*p; sum += g*g; p -= lr * g / (sqrt(sum) + eps).

    eps is 1e-8, not Adagrad's 1e-10: OptimRegime builds Adam(lr=0) first and re-wraps its
    param_groups (utils/optim.py:29,143-146), so Adam's eps leaks [confirmed by
    tests/golden/make_golden.py: 'effective Adagrad group'].  lr_decay = 0, initial accumulator 0.
    Applied to EVERY row (dense gradients, model_config.sparse False).
    """
    dt = p.dtype
    g = g + dt.type(weight_decay) * p
    state_sum += g * g
    std = np.sqrt(state_sum) + dt.type(eps)
    p -= dt.type(lr) * (g / std)
    return p, state_sum


# ------------------------------------------------------------------------------------------------
# filtered ranks                          openkge/dataset.py:423-453 (compute_metrics)
# ------------------------------------------------------------------------------------------------
def filtered_ranks(pred, filt, row_ptr, grp_ptr, ids):
    """Per (row, answer group): true = max_{j in group} pred[row, j]; pred[filter] = -1e8;
    rank = #(pred' > true) + (#(pred' == true)) // 2   (dataset.py:436-446).  int64, exact.

    Groups are CSR-of-CSR: row b owns groups row_ptr[b]..row_ptr[b+1]; group g owns
    ids[grp_ptr[g]..grp_ptr[g+1]] (candidate-relative ids, dataset.py:921-926).
    """
    pred = np.asarray(pred, dtype=np.float32)
    ranks = np.zeros(len(grp_ptr) - 1, dtype=np.int64)
    for b in range(pred.shape[0]):
        masked = np.where(np.asarray(filt[b]).astype(bool), np.float32(-1e8), pred[b])
        for g in range(row_ptr[b], row_ptr[b + 1]):
            true = pred[b, ids[grp_ptr[g]:grp_ptr[g + 1]]].max()
            gt = int((masked > true).sum())
            eq = int((masked == true).sum())
            ranks[g] = gt + eq // 2
    return ranks


def metrics_from_ranks(ranks, row_ptr):
    """MetricResult averages of compute_metrics (dataset.py:447-452, utils/metrics.py:20-23):
    each meter is a count-weighted running mean over rows -> equals the plain mean over groups."""
    ranks = np.asarray(ranks, dtype=np.int64)
    out = {"mrr": 0.0, "mr": 0.0, "h1": 0.0, "h3": 0.0, "h10": 0.0, "h50": 0.0}
    n = len(ranks)
    if n == 0:
        return out, 0
    out["mrr"] = float((1.0 / (ranks + 1).astype(np.float32)).astype(np.float64).sum() / n)
    out["mr"] = float(ranks.sum() / n)
    for k in (1, 3, 10, 50):
        out[f"h{k}"] = float((ranks < k).sum() / n)
    return out, n


# ------------------------------------------------------------------------------------------------
# token-pooled embedder                   openkge/model.py:716-796 (UnigramPoolingRelationEmbedder._encode)
#                                         torch.nn.BatchNorm1d(momentum=0.1, eps=1e-5) as built in model.py:611-616
# ------------------------------------------------------------------------------------------------
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def token_pool(W, token_ids, ids, pool):
    """rows of `ids` -> pooled token embeddings.  token_ids: (|vocab|, L) right-padded with 0; padded positions
    take part like any other (the embedding's row 0 is NOT zero: init normal_ runs over the whole weight,
    model.py:660-661; padding_idx only stops its gradient).  Returns (pooled (n,d), aux for backward)."""
    tok = token_ids[np.asarray(ids).reshape(-1).astype(np.int64)].astype(np.int64)          # (n, L)
    emb = W[tok]                                                                             # (n, L, d)
    if pool == "max":
        arg = emb.argmax(axis=1)                                                             # first maximum, like torch
        return np.take_along_axis(emb, arg[:, None, :], axis=1)[:, 0, :], (tok, arg)
    if pool == "mean":
        lengths = (tok > 0).sum(axis=1, keepdims=True).astype(W.dtype) + W.dtype.type(1e-12)
        return emb.sum(axis=1) / lengths, (tok, lengths)
    return emb.sum(axis=1), (tok, None)


def token_pool_backward(dW, d_pooled, aux, pool):
    """scatter-add d(pooled) into the token table gradient; row 0 (padding_idx) receives nothing"""
    tok, extra = aux
    n, L = tok.shape
    if pool == "max":
        rows = np.take_along_axis(tok, extra, axis=1)                                        # (n, d) token per column
        for i in range(n):
            for k in range(d_pooled.shape[1]):
                if rows[i, k] != 0:
                    dW[rows[i, k], k] += d_pooled[i, k]
        return
    g = d_pooled / extra if pool == "mean" else d_pooled
    for t in range(L):
        sel = tok[:, t] != 0
        np.add.at(dW, tok[sel, t], g[sel])


def batchnorm_train(x, weight, bias, running_mean=None, running_var=None):
    """training-mode BatchNorm1d over the rows of x; updates the running statistics in place (unbiased variance,
    momentum 0.1).  Returns (y, aux)."""
    n = x.shape[0]
    mean = x.mean(axis=0, dtype=np.float64)
    var = ((x - mean) ** 2).mean(axis=0, dtype=np.float64)
    rstd = 1.0 / np.sqrt(var + BN_EPS)
    xhat = ((x - mean) * rstd).astype(x.dtype)
    if running_mean is not None:
        running_mean += BN_MOMENTUM * (mean.astype(x.dtype) - running_mean)
        running_var += BN_MOMENTUM * ((var * n / max(n - 1, 1)).astype(x.dtype) - running_var)
    return xhat * weight + bias, (xhat, rstd.astype(x.dtype))


def batchnorm_train_backward(dy, weight, aux):
    xhat, rstd = aux
    n = dy.shape[0]
    dbias = dy.sum(axis=0, dtype=np.float64).astype(dy.dtype)
    dweight = (dy * xhat).sum(axis=0, dtype=np.float64).astype(dy.dtype)
    dx = weight * rstd * (dy - dbias / n - xhat * dweight / n)
    return dx.astype(dy.dtype), dweight, dbias


def batchnorm_eval(x, weight, bias, running_mean, running_var):
    return (x - running_mean) / np.sqrt(running_var + x.dtype.type(BN_EPS)) * weight + bias


def unigram_step_forward_backward(kind, We, Wr, ent_tokens, rel_tokens, po, sp, cand_ids, labels, pool="sum",
                                  bn_ent=None, bn_rel=None, normalizer=None, p_drop=0.0, keep=None, want_outputs=True):
    """AddLossModule.forward + backward for a token-pooled model (trainer.py:48-113 over model.py:762-796), bce loss.
    bn_* = dict(weight, bias, running_mean, running_var) or None.  Encoding calls happen in the
    reference's order -- candidates, po rows, sp rows -- each with ITS OWN batch statistics.
    Dropout (model.py:783-784: the LAST stage of _encode, after pooling and batch-norm): `keep` = dict of boolean masks
    cand / po_rel / po_ent / sp_ent / sp_rel (this build's Philox masks, or masks captured from the reference), rows
    scaled by 1/(1-p_drop).  `labels`: dense (B,N) array, or (rows, cols) coordinates of the positives.
    Returns dict(loss, outputs, dWe, dWr, d_bn_ent (w,b), d_bn_rel (w,b))."""
    keep = keep or {}
    dt = We.dtype
    scale = dt.type(1.0 / (1.0 - p_drop)) if p_drop > 0 else dt.type(1.0)

    def enc(W, tokens, ids, bn, mask):
        x, paux = token_pool(W, tokens, ids, pool)
        baux = None
        if bn is not None:
            x, baux = batchnorm_train(x, bn["weight"], bn["bias"], bn["running_mean"], bn["running_var"])
        m = None
        if mask is not None and p_drop > 0:
            m = mask.astype(dt) * scale
            x = x * m
        return x, (paux, baux, m)

    C, auxC = enc(We, ent_tokens, cand_ids, bn_ent, keep.get("cand"))
    parts = []
    if po is not None:
        r, auxr = enc(Wr, rel_tokens, po[0], bn_rel, keep.get("po_rel"))
        e, auxe = enc(We, ent_tokens, po[1], bn_ent, keep.get("po_ent"))
        parts.append((DIR_PO, e, r, auxe, auxr))
    if sp is not None:
        e, auxe = enc(We, ent_tokens, sp[0], bn_ent, keep.get("sp_ent"))
        r, auxr = enc(Wr, rel_tokens, sp[1], bn_rel, keep.get("sp_rel"))
        parts.append((DIR_SP, e, r, auxe, auxr))
    X = np.concatenate([score_prefix(kind, d_, e, r, C) for d_, e, r, _, _ in parts])
    B, N = X.shape
    normalizer = float(B * N) if normalizer is None else normalizer
    if isinstance(labels, tuple):
        y = np.zeros((B, N), X.dtype)
        y[np.asarray(labels[0], np.int64), np.asarray(labels[1], np.int64)] = 1
        labels = y
    loss, g = loss_and_dscore(X, labels, LOSS_BCE, 0.0)
    G = (g / X.dtype.type(normalizer)).astype(X.dtype)
    del g, labels
    dWe, dWr = np.zeros_like(We), np.zeros_like(Wr)
    dbe = [np.zeros(We.shape[1], We.dtype), np.zeros(We.shape[1], We.dtype)]
    dbr = [np.zeros(Wr.shape[1], Wr.dtype), np.zeros(Wr.shape[1], Wr.dtype)]

    def back(d_rows, aux, W_grad, bn, acc):
        paux, baux, m = aux
        if m is not None:
            d_rows = d_rows * m
        if bn is not None:
            d_rows, dw, db = batchnorm_train_backward(d_rows, bn["weight"], baux)
            acc[0] += dw
            acc[1] += db
        token_pool_backward(W_grad, d_rows, paux, pool)

    dC = np.zeros_like(C)
    row = 0
    for d_, e, r, auxe, auxr in parts:
        Gp = G[row:row + e.shape[0]]
        q = prefix_query(kind, d_, e, r)
        dC += Gp.T @ q
        de, dr = prefix_query_backward(kind, d_, e, r, Gp @ C)
        back(de, auxe, dWe, bn_ent, dbe)
        back(dr, auxr, dWr, bn_rel, dbr)
        row += e.shape[0]
    back(dC, auxC, dWe, bn_ent, dbe)
    return dict(loss=float(loss), outputs=X if want_outputs else None, dWe=dWe, dWr=dWr, d_bn_ent=tuple(dbe), d_bn_rel=tuple(dbr))


# ------------------------------------------------------------------------------------------------
# batch producer                          openkge/dataset.py:724-940 (OneToNMentionRelationDataset_collate_func)
#                                         utils/misc.py:56-89 (packed answer groups)
# ------------------------------------------------------------------------------------------------
def pack_groups(groups):
    """utils/misc.py:56-70.  k answer groups -> [b_0+L, ..., b_k+L, 0, ids...] with b_0 = 0, b_i = cumulative
    group lengths and L = k+2 the header length: header entries are slice-relative positions, 0 ends the header."""
    offsets, ids = [0], []
    for g in groups:
        ids.extend(g if isinstance(g, (list, tuple)) else [g])
        offsets.append(len(ids))
    L = len(offsets) + 1
    return [o + L for o in offsets] + [0] + ids


def unpack_groups(packed):
    """utils/misc.py:72-89 -> (list of groups, flat id list)."""
    packed = [int(x) for x in packed]
    header = []
    for off in packed:
        if off == 0:
            break
        header.append(off)
    groups = [packed[a:b] for a, b in zip(header[:-1], header[1:])]
    flat = packed[header[0]:header[-1]] if len(header) > 1 else []
    return groups, flat


def collate_batch(rows, seen_entities, all_splits_entities, n_entities, entity_offset, is_training,
                  use_batch_shared, min_size_batch_labels=0, negatives=None):
    """dataset.py:724-940 on one batch of (P,7) prefix rows [a, b, this_start, this_end, all_start, all_end, slot].

    Returns dict:
      po = (rel[b0], obj[b0]) or None   slot 0 rows, batch order      (dataset.py:884-891,932)
      sp = (subj[b1], rel[b1]) or None  slot 2 rows
      labels  : sorted unique (row, col) pairs, rows numbered po-first (the dense label tensor as coordinates)
      groups  : per row, list of groups of candidate-relative ids       (label_ids, evaluation only)
      filters : per row, sorted unique candidate positions              (filter_mask rows, evaluation only)
      cand    : candidate entity ids in column order
      normalizer_loss = B*N, normalizer_metric = number of labels       (dataset.py:934-935)
    `negatives`: the already sampled fill-up ids in the order they are appended (dataset.py:853-860 takes them
    from a Python set, whose order is an implementation detail; callers replay the reference's order or use
    their own sampler)."""
    items = {0: [], 2: []}
    index = {}                                     # batch-shared: entity id -> column, in first-seen order
    for row in rows:
        a, b, ts, te, as_, ae, slot = (int(x) for x in row)
        groups, flat = unpack_groups(seen_entities[ts:te])
        everything = [int(x) for x in all_splits_entities[as_:ae]]
        items[slot].append(((a, b), groups, flat, everything))
        if use_batch_shared:
            for e in (flat if is_training else everything):
                index.setdefault(e, len(index))
    if use_batch_shared:
        m = max(0, min_size_batch_labels or 0)
        if len(index) >= m:
            cand = list(index)
        else:
            extra = [int(x) for x in (negatives if negatives is not None else [])]
            cand = (list(index) + [e for e in extra if e not in index])[:m]
        col = index.__getitem__
    else:
        cand = list(range(entity_offset, n_entities))
        col = lambda e: e - entity_offset                  # noqa: E731
    labels, groups_out, filters, parts = set(), [], [], {}
    r = 0
    for slot in (0, 2):
        parts[slot] = None
        if not items[slot]:
            continue
        parts[slot] = (np.asarray([it[0][0] for it in items[slot]], np.int32),
                       np.asarray([it[0][1] for it in items[slot]], np.int32))
        for _, groups, flat, everything in items[slot]:
            labels.update((r, col(e)) for e in flat)
            if not is_training:
                groups_out.append([[col(e) for e in g] for g in groups])
                filters.append(sorted({col(e) for e in everything}))
            r += 1
    return dict(po=parts[0], sp=parts[2], labels=sorted(labels), groups=groups_out, filters=filters,
                cand=np.asarray(cand, np.int32), normalizer_loss=float(r * len(cand)),
                normalizer_metric=float(len(labels)))


# ------------------------------------------------------------------------------------------------
# on-disk format -> dataset tensors        openkge/dataset.py:480-710
# 5-column TSV  s \t p \t o \t subject-mention-ids \t object-mention-ids   (utils/map_dataset_to_ids.py:11-17)
# ------------------------------------------------------------------------------------------------
DIRECTIONS = {          # name: (key column 2, key column 1, prefix columns, slot, answer column)  dataset.py:489-492
    "sp_o": (1, 0, (0, 1), 2, 4),
    "po_s": (1, 2, (1, 2), 0, 3),
}


def collect_prefix_groups(lines, direction):
    """dataset.py:494-518.  Lines are sorted by the STRING of the relation column, then (stably) by the STRING of
    the prefix entity column; consecutive lines with the same prefix form one record whose answer groups are the
    mention-id lists of the answer column.  The record still open when the input ends is never written
    (there is no flush after the loop): the last prefix in sort order is dropped."""
    pref_p, pref_e, (c1, c2), slot, ans = DIRECTIONS[direction]
    rows = [ln.split("\t") for ln in lines]
    rows = sorted(sorted(rows, key=lambda f: f[pref_p]), key=lambda f: f[pref_e])
    out, cur = [], None
    for f in rows:
        prefix = (int(f[c1]), int(f[c2]))
        group = [int(i) for i in f[ans].split()]
        if cur is not None and cur["prefix"] == prefix:
            cur["entities"].append(group)
        else:
            if cur is not None:
                out.append(cur)
            cur = {"prefix": prefix, "entities": [group], "slot": slot}
    return out


def merge_all_splits(train, valid, test):
    """dataset.py:520-565: records of the three splits, stably sorted by prefix (numeric), answers united per
    prefix.  The reference keeps each union in a Python set's iteration order; here ascending."""
    merged = {}
    for rec in sorted(train + valid + test, key=lambda r: r["prefix"]):
        merged.setdefault(rec["prefix"], set()).update(e for g in rec["entities"] for e in g)
    return [(p, sorted(v)) for p, v in merged.items()]


def dataset_tensors(records, merged, is_training, max_size_prefix_label=-1):
    """dataset.py:567-710.  records / merged: {"sp_o": [...], "po_s": [...]} of one split / of all splits.
    -> (seen_prefixes (P,7) int32, seen_entities int32, all_splits_entities int32)."""
    all_ents, coords = [], {}
    for d in ("sp_o", "po_s"):
        for prefix, ents in merged[d]:
            coords[(d, prefix)] = (len(all_ents), len(all_ents) + len(ents))
            all_ents.extend(ents)
    prefixes, seen = [], []
    for d in ("sp_o", "po_s"):
        for rec in records[d]:
            groups = rec["entities"]
            chunked = is_training and max_size_prefix_label > 1 and len(groups) > max_size_prefix_label
            step = max_size_prefix_label if chunked else max(1, len(groups))
            for off in range(0, len(groups), step):
                packed = pack_groups(groups[off:off + step])
                a, b = (0, 0) if is_training else coords[(d, rec["prefix"])]
                prefixes.append([rec["prefix"][0], rec["prefix"][1], len(seen), len(seen) + len(packed), a, b, rec["slot"]])
                seen.extend(packed)
    return (np.asarray(prefixes, np.int32).reshape(-1, 7), np.asarray(seen, np.int32), np.asarray(all_ents, np.int32))


# ------------------------------------------------------------------------------------------------
# counter-based dropout masks (this build's replacement for torch's bernoulli_, SURVEY.md 'hard
# parts'): Philox4x32-10, key = seed, counter = (row, column/8, stream, step).  Integer work ->
# the HIP kernels reproduce these masks bit for bit.
# ------------------------------------------------------------------------------------------------
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011).  Inputs uint32 arrays (broadcastable)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c0.astype(np.uint64)
            p1 = _PHILOX_M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def dropout_threshold(p):
    """keep  <=>  u16 >= threshold, threshold = floor(p * 2^16) clipped to [0, 65535]."""
    return np.uint32(min(max(int(float(p) * 65536.0), 0), 65535))


def dropout_keep_mask(seed, stream, step, nrows, d, p, row_keys=None):
    """keep[row, k] for k < d.  One Philox4x32-10 call yields 8 uniform 16-bit numbers:
    counter = (row_key, k >> 3, stream, step); element k uses word (k & 7) >> 1, low half for even k,
    high half for odd k.  keep <=> that u16 >= floor(p * 65536).

    row_key defaults to the row's position in the list (candidate position / batch row).
    """
    rows = np.arange(nrows, dtype=np.uint32) if row_keys is None else np.asarray(row_keys, dtype=np.uint32)
    no = (d + 7) // 8
    o = np.arange(no, dtype=np.uint32)
    k0 = np.uint32(int(seed) & 0xFFFFFFFF)
    k1 = np.uint32((int(seed) >> 32) & 0xFFFFFFFF)
    w = philox4x32_10(rows[:, None], o[None, :], np.uint32(stream), np.uint32(int(step) & 0xFFFFFFFF), k0, k1)
    w = np.stack(w, axis=-1)                                   # [rows, no, 4] uint32
    u16 = np.stack([w & np.uint32(0xFFFF), w >> np.uint32(16)], axis=-1)   # [rows, no, 4, 2]
    u = u16.reshape(len(rows), no * 8)[:, :d]
    return u >= dropout_threshold(p)
