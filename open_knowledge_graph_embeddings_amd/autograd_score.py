"""Autograd graph for the plugin methods outside AddLossModule: `encode_subj/rel/obj`, `get_all_*`, `_score`, `triple_score`,
`sp_prefix_score` / `po_prefix_score` called with gradients enabled (a user's own loss on the scores).

The reference gets these gradients from ATen's autograd through its op sequence (openkge/model.py:198-240, :268-278,
:455-480).  Here the forward AND the backward stay on the HIP kernels: okge_encode_rows / okge_score_prefixes forward;
okge_prefix_score_backward (`G . C`, `G^T . Q` on a hand-written exact-fp32 MFMA GEMM + the transpose of the query fold) and
okge_scatter_rows (gradient rows summed per table row in sorted order, dropout masks replayed) backward.  The training path the
reference's Trainer uses (AddLossModule: fused loss + backward, no (B, N) gradient block) does not come through here.
"""
from __future__ import annotations

import torch

from . import hotpath as H


def fold_query(scorer: str, sp: bool, ent: torch.Tensor, rel: torch.Tensor) -> torch.Tensor:
    """Q with score = Q . cand^T (openkge/model.py:205-216 regrouped, :268-274): the prefix entity and relation rows folded
    into one (b, d) operand.  Differentiable torch ops."""
    if scorer == "distmult":
        return ent * rel
    h = ent.shape[-1] // 2
    e1, e2, r1, r2 = ent[:, :h], ent[:, h:], rel[:, :h], rel[:, h:]
    if sp:            # (s1 r1 - s2 r2) . o1 + (s2 r1 + s1 r2) . o2
        return torch.cat([e1 * r1 - e2 * r2, e2 * r1 + e1 * r2], 1)
    return torch.cat([e1 * r1 + e2 * r2, e2 * r1 - e1 * r2], 1)      # po: (o1 r1 + o2 r2) . s1 + (o2 r1 - o1 r2) . s2


def triple_score(scorer: str, subj, rel, obj):
    """(b, 1) Hadamard-form scores of encoded triples (openkge/model.py:231-238, :276), differentiable torch ops"""
    subj, rel, obj = (t.reshape(-1, t.shape[-1]) for t in (subj, rel, obj))
    if scorer == "distmult":
        return (subj * obj * rel).sum(1, keepdim=True)
    h = rel.shape[-1] // 2
    s1, s2, r1, r2, o1, o2 = subj[:, :h], subj[:, h:], rel[:, :h], rel[:, h:], obj[:, :h], obj[:, h:]
    return (s1 * r1 * o1 + s2 * r1 * o2 + s1 * r2 * o2 - s2 * r2 * o1).sum(1, keepdim=True)


class PrefixScoreFn(torch.autograd.Function):
    """(b, N) scores of encoded prefix rows against encoded candidate rows: forward okge_score_prefixes, backward
    dQ = G . C, dC = G^T . Q and the chain rule of `fold_query`"""

    @staticmethod
    def forward(ctx, ent, rel, cand, engine, scorer, sp):
        ent_c, rel_c, cand_c = ent.detach().contiguous(), rel.detach().contiguous(), cand.detach().contiguous()
        b = rel_c.shape[0]
        ar = torch.arange(b, dtype=torch.int32, device=rel_c.device)
        batch = H.PrefixBatch(sp_subj=ar, sp_rel=ar) if sp else H.PrefixBatch(po_rel=ar, po_obj=ar)
        batch.cand_table, batch.cand_first, batch.n_cand = cand_c, 0, cand_c.shape[0]
        ctx.save_for_backward(ent_c, rel_c, cand_c)
        ctx.scorer, ctx.sp, ctx.engine = scorer, sp, engine
        return engine.score(ent_c, rel_c, scorer, batch)

    @staticmethod
    def backward(ctx, g):
        ent, rel, cand = ctx.saved_tensors
        g = g.contiguous()
        need_e, need_r, need_c = ctx.needs_input_grad[:3]
        d_ent, d_rel, d_cand = ctx.engine.prefix_score_backward(ctx.scorer, ctx.sp, g, ent, rel, cand, need_e, need_r, need_c)
        return d_ent, d_rel, d_cand, None, None, None


class EncodeRowsFn(torch.autograd.Function):
    """dropout(table[ids]) (openkge/model.py:455-470 without the variants): forward okge_encode_rows with the Philox masks,
    backward the same masks (they depend on (seed, stream, step, row position, column), not on the row's content: encoding
    rows of ones returns keep / (1 - p)) and a scatter-add into the table's gradient"""

    @staticmethod
    def forward(ctx, table, ids, first_id, n, engine, drop):
        ctx.engine, ctx.drop, ctx.first_id, ctx.n, ctx.shape = engine, drop, first_id, n, tuple(table.shape)
        ctx.save_for_backward(ids if ids is not None else torch.zeros(0, dtype=torch.int32, device=table.device))
        ctx.has_ids = ids is not None
        return engine.encode_rows(table.detach(), ids, first_id, n, drop)

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        g = g.contiguous()
        d_table = torch.zeros(ctx.shape, dtype=g.dtype, device=g.device)     # dense, like the reference's embedding gradients
        # (row 0 -- nn.Embedding(padding_idx=PAD = 0), model.py:390-391 -- gets no gradient: the kernel skips it)
        ctx.engine.scatter_rows(g, ids if ctx.has_ids else None, ctx.first_id, d_table, ctx.drop if ctx.drop is not None else H.NO_DROP)
        return d_table, None, None, None, None, None


class MaskRowsFn(torch.autograd.Function):
    """dropout(rows) on rows that are already in hand (`lookup=False`, openkge/model.py:459-470)"""

    @staticmethod
    def forward(ctx, rows, engine, drop):
        ctx.engine, ctx.drop = engine, drop
        r = rows.detach().contiguous()
        return engine.encode_rows(r, None, 0, r.shape[0], drop)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if ctx.drop is not None and ctx.drop.p > 0:          # the forward's kernel on the gradient rows: the same masks
            g = ctx.engine.encode_rows(g, None, 0, g.shape[0], ctx.drop)
        return g, None, None
