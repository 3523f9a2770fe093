"""Training step with the entity table row-sharded over the GPUs of one node (one process per GPU,
torch.distributed backend "nccl" = RCCL over xGMI).

The reference has no counterpart (its only multi-device path is nn.DataParallel, openkge/trainer.py:143-145).
Every rank processes the SAME batch of prefixes against ITS OWN slice of the candidate entities
(tensor-parallel over the candidate axis); scores are independent per candidate and the BCE
loss is separable, so one step needs exactly two small exchanges:

    all-gather       [cap, d]   the masked prefix entity rows each rank OWNS (cap = the largest per-rank count, padded), when
                                the batch carries a host-built exchange plan (`make_exchange_plan`: the ids are known on
                                the host before the batch is uploaded) -- half the bytes of the fallback, an
    all-reduce(sum)  [B, d]     of the row block that is zero except on each row's owner; every rank then folds the rows
                                with its replicated relation rows into the query block itself
    all-reduce(sum)  [B, d]     partial query gradients dQ
(the scalar loss stays a per-rank partial until `reduce_loss()` is called: the reference looks at it every 100 steps)

The KL loss (log_softmax over ALL candidates, trainer.py:99-101) adds one: all-gather of the [B] per-shard row
log-sum-exp.  Evaluation (`ShardedEvaluator`) exchanges the true-answer scores (all-reduce max) and the integer
{#greater, #equal} counts (all-reduce sum); exact ranks need counts, not a per-shard top-k.

Entity rows, their dense gradients and Adagrad accumulators never leave their rank; the (small) relation table is
replicated and its gradient is formed identically everywhere from the exchanged entity rows.

The arithmetic is delegated to an `engine` (HotPath: the HIP kernels).  Tests inject a CPU engine so the exchange
protocol can run under gloo without a GPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import hotpath as H
from .train_step import FusedTrainStep


def shard_range(n_ent, world, rank):
    """Contiguous, equal-sized row ranges (the last one may be shorter)."""
    per = (n_ent + world - 1) // world
    return min(rank * per, n_ent), min((rank + 1) * per, n_ent)


class ExchangePlan:
    """Who owns which prefix entity row of ONE global batch, built on the host from the ids (they exist there before
    the batch is uploaded: the producer / the synthetic generator made them).
      cap        rows every rank contributes to the all-gather (the largest per-rank count)
      owned[r]   int64[cap] batch rows rank r sends: its own rows, padded with rows it does NOT own (zero on that rank)
      slot       int64[B]   position of batch row b in the gathered [world * cap] block"""

    def __init__(self, cap, owned, slot):
        self.cap, self.owned, self.slot = cap, owned, slot


def make_exchange_plan(po_obj, sp_subj, n_ent, world, device):
    """po_obj / sp_subj: HOST int arrays of the global batch (po rows first).  Returns None when the all-gather would
    move more rows than the all-reduce it replaces (very skewed ownership: world * cap > 2 B)."""
    import numpy as np
    ent = np.concatenate([np.asarray(po_obj, np.int64).reshape(-1), np.asarray(sp_subj, np.int64).reshape(-1)])
    B = len(ent)
    per = (n_ent + world - 1) // world
    owner = np.minimum(ent // per, world - 1)
    counts = np.bincount(owner, minlength=world)
    cap = int(counts.max()) if B else 0
    if B == 0 or world * cap > 2 * B:
        return None
    order = np.argsort(owner, kind="stable")
    start = np.concatenate([[0], np.cumsum(counts)])
    slot = np.empty(B, np.int64)
    owned = []
    for r in range(world):
        mine = order[start[r]:start[r + 1]]
        slot[mine] = r * cap + np.arange(len(mine))
        pad = cap - len(mine)
        if pad:
            others = np.flatnonzero(owner != r)[:1]                  # a row this rank does not own: zero in its block
            mine = np.concatenate([mine, np.repeat(others, pad)])
        owned.append(torch.from_numpy(mine.astype(np.int64)).to(device))
    return ExchangePlan(cap, owned, torch.from_numpy(slot).to(device))


class RowSegments:
    """Host-built plans for the prefix backward (okge_prefix_backward_segmented): batch rows grouped by relation id (`rel`)
    and by prefix entity id (`ent`), each (order int32[B], seg_ptr int32[n_seg + 1]) on the device, or None = keep atomics."""

    def __init__(self, rel=None, ent=None):
        self.rel, self.ent = rel, ent


def _segments(ids, device):
    import numpy as np
    order = np.argsort(ids, kind="stable")
    srt = ids[order]
    heads = np.flatnonzero(np.concatenate([[True], srt[1:] != srt[:-1]]))
    seg_ptr = np.concatenate([heads, [ids.size]])
    return (torch.from_numpy(order.astype(np.int32)).to(device), torch.from_numpy(seg_ptr.astype(np.int32)).to(device)), len(heads)


def make_row_segments(po_rel, po_obj, sp_subj, sp_rel, device, min_rows=1024, min_rows_per_relation=4.0):
    """HOST int arrays of the global batch (po rows first) -> RowSegments, or None for batches where the float atomics are
    cheaper than a second launch: fewer than `min_rows` rows (0.2 M atomics = 4 us at B = 512; the segment launch costs
    about that).  The relation plan is left out when relations hardly repeat (fewer than `min_rows_per_relation` rows per
    distinct relation: |R| = 100 k at the OLPBENCH shape), the entity plan is kept (one segment per distinct entity: plain
    read-modify-write instead of B x d atomics)."""
    import numpy as np
    rel = np.concatenate([np.asarray(po_rel, np.int64).reshape(-1), np.asarray(sp_rel, np.int64).reshape(-1)])
    ent = np.concatenate([np.asarray(po_obj, np.int64).reshape(-1), np.asarray(sp_subj, np.int64).reshape(-1)])
    if rel.size == 0 or rel.size < min_rows or rel.size != ent.size:
        return None
    r, n_r = _segments(rel, device)
    e, _ = _segments(ent, device)
    return RowSegments(rel=r if rel.size >= min_rows_per_relation * n_r else None, ent=e)


class ShardedTrainStep:
    def __init__(self, E_local, R, scorer, n_ent, min_entities_size=2, lr=0.3, weight_decay=1e-10, eps=1e-8,
                 loss="bce", label_smoothing=0.0, input_dropout=0.0, relation_input_dropout=0.0, seed=0, engine=None,
                 group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ent_lo, self.ent_hi = shard_range(n_ent, self.world, self.rank)
        if E_local.shape[0] != self.ent_hi - self.ent_lo:
            raise ValueError("E_local must hold exactly this rank's rows")
        if loss not in ("bce", "kl"):
            raise NotImplementedError(f"loss {loss!r}")
        self.E, self.R, self.scorer = E_local, R, scorer
        self.n_ent, self.min_ent = n_ent, min_entities_size
        self.lr, self.weight_decay, self.eps = lr, weight_decay, eps
        self.loss, self.label_smoothing = loss, label_smoothing
        self.input_dropout, self.relation_input_dropout, self.seed = input_dropout, relation_input_dropout, seed
        self.engine = engine or H.HotPath(E_local.device)
        self.dE, self.dR = torch.zeros_like(E_local), torch.zeros_like(R)
        self.sumE, self.sumR = torch.zeros_like(E_local), torch.zeros_like(R)
        self.steps = 0
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=E_local.device)
        # local candidates of the 1-vs-all list (global ids min_ent .. n_ent-1)
        c_lo = max(self.ent_lo, min_entities_size)
        self.cand_first_local = c_lo - self.ent_lo
        self.n_cand_local = max(0, self.ent_hi - c_lo)
        self.n_cand_global = n_ent - min_entities_size
        self.shard = H.Shard(self.ent_lo, self.ent_hi, c_lo - min_entities_size)
        # OKGE_SHARDED_FORCE_EXCHANGE=1: a one-rank group still walks the exchange path (collectives of one rank):
        # rehearsal of the RCCL calls on a one-GPU box
        import os
        self.force_exchange = os.environ.get("OKGE_SHARDED_FORCE_EXCHANGE") == "1"

    def state_tensors(self):
        """every tensor a step mutates (train_step.GraphedTrainStep snapshots them around its warm-up)"""
        return [self.E, self.R, self.dE, self.dR, self.sumE, self.sumR]

    def _set_dropout(self, batch):
        # step_dev (attached by GraphedTrainStep): the kernels read the step from a device counter the graph increments, so a
        # replay of the captured step -- collectives included -- draws fresh masks
        pe, pr, s, t, sd = self.input_dropout, self.relation_input_dropout, self.seed, self.steps, getattr(self, "step_dev", None)
        batch.drop_cand = H.DropoutSpec(pe, s, H.STREAM_CAND, t, step_dev=sd)
        batch.drop_po_ent = H.DropoutSpec(pe, s, H.STREAM_PO_ENT, t, step_dev=sd)
        batch.drop_sp_ent = H.DropoutSpec(pe, s, H.STREAM_SP_ENT, t, step_dev=sd)
        batch.drop_po_rel = H.DropoutSpec(pr, s, H.STREAM_PO_REL, t, step_dev=sd)
        batch.drop_sp_rel = H.DropoutSpec(pr, s, H.STREAM_SP_REL, t, step_dev=sd)

    def _entity_rows(self, batch, plan):
        """masked prefix entity rows of ALL prefixes on every rank: all-gather of the owned rows (plan) or all-reduce"""
        eng = self.engine
        er = eng.encode_entity_rows(self.E, self.R, self.scorer, batch, self.shard)
        if self.world == 1 and not self.force_exchange:
            return er
        if plan is None:
            dist.all_reduce(er, group=self.group)
            return er
        mine = er.index_select(0, plan.owned[self.rank])             # [cap, ld]; padding rows are zero here
        every = torch.empty((self.world * plan.cap, er.shape[1]), dtype=er.dtype, device=er.device)
        dist.all_gather_into_tensor(every, mine, group=self.group)
        er.zero_()
        er[:batch.B] = every.index_select(0, plan.slot)
        return er

    def step(self, batch: H.PrefixBatch, plan: ExchangePlan = None, rel_segments=None):
        """`batch` is the GLOBAL batch (identical on every rank); 1-vs-all candidates; positives carry global columns.
        `plan` (make_exchange_plan, optional): exchange 1 as an all-gather of owned rows instead of an all-reduce.
        `rel_segments` (make_row_segments, optional): relation / entity gradients by sorted segments instead of float atomics."""
        if batch.cand_ids is not None:
            raise NotImplementedError("batch-shared sampled candidates are too few to shard: use replicas")
        self.steps += 1
        self._set_dropout(batch)
        eng = self.engine
        if self.world == 1 and isinstance(eng, H.HotPath) and not self.force_exchange:
            # one rank owns everything: the fused single-device call (no slab reduction, no exchange buffers)
            local = H.PrefixBatch(po_rel=batch.po_rel, po_obj=batch.po_obj, sp_subj=batch.sp_subj, sp_rel=batch.sp_rel,
                                  pos_row=batch.pos_row, pos_col=batch.pos_col, cand_first=self.cand_first_local,
                                  n_cand=self.n_cand_local, drop_cand=batch.drop_cand, drop_po_ent=batch.drop_po_ent,
                                  drop_sp_ent=batch.drop_sp_ent, drop_po_rel=batch.drop_po_rel, drop_sp_rel=batch.drop_sp_rel)
            eng.forward_backward(self.E, self.R, self.scorer, local, self.dE, self.dR, loss=self.loss,
                                 label_smoothing=self.label_smoothing, normalizer=float(batch.B) * float(self.n_cand_global),
                                 loss_out=self.loss_out, grads_zero=True)
            eng.adagrad2(self.E, self.dE, self.sumE, self.R, self.dR, self.sumR, self.lr, self.weight_decay, self.eps,
                         zero_grad=True)
            return self.loss_out
        # 1. masked entity rows of the prefixes whose entity lives here -> all rows everywhere; fold locally
        er = self._entity_rows(batch, plan)
        qe = (eng.fold_queries(self.E, self.R, self.scorer, batch, er), er)
        # 2. local candidates: loss partial, local entity gradients, partial query gradients
        local = H.PrefixBatch(po_rel=batch.po_rel, po_obj=batch.po_obj, sp_subj=batch.sp_subj, sp_rel=batch.sp_rel,
                              pos_row=batch.pos_row, pos_col=batch.pos_col, cand_first=self.cand_first_local,
                              n_cand=self.n_cand_local, drop_cand=batch.drop_cand)
        dq = torch.empty_like(qe[0])
        row_lse = None
        if self.loss == "kl":
            lse = eng.row_logsumexp(self.E, self.R, self.scorer, qe[0], batch.B, local, self.shard)
            every = torch.empty(self.world * batch.B, dtype=lse.dtype, device=lse.device)
            dist.all_gather_into_tensor(every, lse, group=self.group)
            row_lse = eng.merge_logsumexp(every.view(self.world, batch.B))
        eng.train_tiles(self.E, self.R, self.scorer, qe[0], local, self.shard, self.dE, dq, self.n_cand_global,
                        loss=self.loss, label_smoothing=self.label_smoothing,
                        normalizer=float(batch.B) * float(self.n_cand_global), loss_out=self.loss_out, grads_zero=True,
                        row_lse=row_lse)
        if self.world > 1 or self.force_exchange:
            dist.all_reduce(dq, group=self.group)
        # 3. chain rule: entity rows by their owner, relation rows everywhere (identical)
        if rel_segments is not None:
            eng.prefix_backward(self.E, self.R, self.scorer, batch, self.shard, dq, qe[1], self.dE, self.dR, rel_segments=rel_segments)
        else:
            eng.prefix_backward(self.E, self.R, self.scorer, batch, self.shard, dq, qe[1], self.dE, self.dR)
        # 4. dense Adagrad on the local entity rows and on the replicated relation table
        eng.adagrad2(self.E, self.dE, self.sumE, self.R, self.dR, self.sumR, self.lr, self.weight_decay, self.eps,
                     zero_grad=True)
        return self.loss_out

    def reduce_loss(self):
        """The step's summed loss over ALL candidates (the value `step` returns covers this rank's candidates only).
        One small all-reduce, paid only when somebody looks at the loss: the reference prints it every
        `print_freq` = 100 steps (trainer.py:296-330), so it is kept off the per-step path."""
        total = self.loss_out.clone()
        dist.all_reduce(total, group=self.group)
        return total


class ShardedEvaluator:
    """Filtered ranks with the candidates sharded like the entity table (compute_metrics' rank rule,
    dataset.py:423-446; exchange plan of SURVEY.md section 8e).

    `ranks()` runs the FUSED counting sweep on this rank's candidates (okge_evaluate_fused_shard; slot sizes up to 512):
        exchange 1   the prefixes' entity rows -> folded queries (all-gather of the owned rows with an ExchangePlan, else
                     all-reduce), as in training
        points       every answer group's true score over the ids this rank holds       -> all-reduce(MAX)  [n_groups] floats
        sweep        the local candidate tiles against the global true scores, counting > / == in registers
        counts       + the filter correction for the filter columns this rank holds     -> all-reduce(SUM)  [n_groups, 2] int64
        rank = #greater + #equal // 2: exact, identical on every rank, bit-equal to the single-device evaluation.
    No (B, N / world) score block exists (5.1 GB per rank and batch at |E| = 2.5 M, B = 4096); `ranks_materialised()` keeps
    the score-block path (any slot size)."""

    def __init__(self, E_local, R, scorer, n_ent, min_entities_size=2, engine=None, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.ent_lo, self.ent_hi = shard_range(n_ent, self.world, self.rank)
        if E_local.shape[0] != self.ent_hi - self.ent_lo:
            raise ValueError("E_local must hold exactly this rank's rows")
        self.E, self.R, self.scorer = E_local, R, scorer
        self.engine = engine or H.HotPath(E_local.device)
        c_lo = max(self.ent_lo, min_entities_size)
        self.cand_first_local = c_lo - self.ent_lo
        self.n_cand_local = max(0, self.ent_hi - c_lo)
        self.n_cand_global = n_ent - min_entities_size
        self.col0 = c_lo - min_entities_size
        self.shard = H.Shard(self.ent_lo, self.ent_hi, self.col0)

    def queries(self, batch: H.PrefixBatch, plan: ExchangePlan = None):
        """folded query block of ALL prefixes on every rank (exchange 1 of the training step, eval mode)"""
        eng = self.engine
        er = eng.encode_entity_rows(self.E, self.R, self.scorer, batch, self.shard)
        if self.world > 1:
            if plan is None:
                dist.all_reduce(er, group=self.group)
            else:
                mine = er.index_select(0, plan.owned[self.rank])             # [cap, ld]; padding rows are zero here
                every = torch.empty((self.world * plan.cap, er.shape[1]), dtype=er.dtype, device=er.device)
                dist.all_gather_into_tensor(every, mine, group=self.group)
                er.zero_()
                er[:batch.B] = every.index_select(0, plan.slot)
        return eng.fold_queries(self.E, self.R, self.scorer, batch, er)

    def _local(self, batch):
        return H.PrefixBatch(po_rel=batch.po_rel, po_obj=batch.po_obj, sp_subj=batch.sp_subj, sp_rel=batch.sp_rel,
                             cand_first=self.cand_first_local, n_cand=self.n_cand_local)

    def local_scores(self, batch: H.PrefixBatch, plan: ExchangePlan = None):
        return self.engine.score_queries(self.E, self.R, self.scorer, self.queries(batch, plan), batch.B, self._local(batch),
                                         self.shard)

    def ranks(self, batch: H.PrefixBatch, filt_ptr, filt_col, row_ptr, grp_ptr, ids, plan: ExchangePlan = None):
        """int64 rank per answer group, identical on every rank.  Index arrays are global (positions in the full
        candidate list) and identical on every rank."""
        eng = self.engine
        n_groups = int(grp_ptr.numel()) - 1
        dev = self.E.device
        counts = torch.zeros((max(n_groups, 1), 2), dtype=torch.int64, device=dev)
        if n_groups == 0:
            return counts[:0, 0]
        Q = self.queries(batch, plan)
        true = torch.full((n_groups,), float("-inf"), dtype=torch.float32, device=dev)
        local = self._local(batch)
        args = (self.E, self.R, self.scorer, Q, batch.B, local, self.shard, self.n_cand_global, filt_ptr, filt_col, row_ptr,
                grp_ptr, ids, true, counts)
        if self.n_cand_local > 0:
            eng.evaluate_fused_shard(1, *args)
        dist.all_reduce(true, op=dist.ReduceOp.MAX, group=self.group)
        if self.n_cand_local > 0:
            eng.evaluate_fused_shard(2, *args)
            eng.evaluate_fused_shard(4, *args)
        dist.all_reduce(counts, group=self.group)
        return counts[:n_groups, 0] + counts[:n_groups, 1] // 2

    def ranks_materialised(self, batch: H.PrefixBatch, filt_ptr, filt_col, row_ptr, grp_ptr, ids, plan: ExchangePlan = None):
        """the same ranks through this rank's (B, N / world) score block (any slot size)"""
        eng = self.engine
        x = self.local_scores(batch, plan)
        true = eng.group_true_scores(x, self.col0, row_ptr, grp_ptr, ids)
        dist.all_reduce(true, op=dist.ReduceOp.MAX, group=self.group)
        counts = eng.rank_counts(x, self.col0, filt_ptr, filt_col, row_ptr, true)
        dist.all_reduce(counts, group=self.group)
        return counts[:, 0] + counts[:, 1] // 2


class ReplicaTrainStep(FusedTrainStep):
    """Replicas (last row): batch-shared sampled candidate lists (1-vs-N with N of a few
    thousand, dataset.py:853-860) are too short to shard, so every rank keeps both tables whole, runs the fused
    step on ITS OWN batch (own prefixes, own candidate list, own dropout stream) and the dense gradients are
    averaged with ONE all-reduce over a flat buffer holding dE and dR back to back; the Adagrad sweep then applies
    the same update everywhere, so the replicas never drift.  The reference's nn.DataParallel branch
    (trainer.py:143-145) is the closest counterpart."""

    def __init__(self, E, R, scorer, group=None, **kw):
        super().__init__(E, R, scorer, **kw)
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        n_e = (E.numel() + 3) // 4 * 4                       # keep dR 16-byte aligned for the Adagrad sweep
        self.flat_grad = torch.zeros(n_e + R.numel(), dtype=E.dtype, device=E.device)
        self.dE = self.flat_grad[:E.numel()].view_as(E)
        self.dR = self.flat_grad[n_e:].view_as(R)
        self.seed = self.seed + 1000003 * self.rank          # independent dropout masks per replica

    def step(self, batch: H.PrefixBatch, normalizer=None):
        self.steps += 1
        if normalizer is None:
            normalizer = float(batch.B) * float(batch.n_candidates)
        loss = self.forward_backward(batch, normalizer * self.world)       # mean over replicas
        dist.all_reduce(self.flat_grad, group=self.group)
        loss_work = dist.all_reduce(loss, group=self.group, async_op=True)
        self._grads_zero = False
        self.optimizer_step()
        self._grads_zero = True
        loss_work.wait()
        return loss


class ReplicaStep:
    """Data-parallel replicas around ANY step object that keeps dense gradients in persistent tensors -- used for the
    token-pooled models (BASELINE configs[4]: batch-shared candidate lists of a few thousand ids are too short to shard;
    the token tables are small enough to replicate).  Protocol, per step:
        inner.forward_backward(batch, normalizer * world)       own batch, own dropout stream; gradients of the mean
        exchange (below)                                        gradients summed, batch-norm running statistics averaged
        inner.optimizer_step()                                  the same update everywhere: replicas never drift

    The exchange moves only what the step touched.  A token-table gradient is dense storage but row-sparse content: a
    batch's <= N + B entities name a few 10^4 of the 2 x 10^5 token rows (cfg5: 256 MB of dense gradient, ~10-40 MB of
    touched rows), so all-reducing the whole table would cost several step times on any link.  Instead, while the
    forward / backward runs on the current stream, a side stream
        1. marks the rows this replica's batch touches in a byte mask over all sparse tables (ids -> token rows: known
           from the batch alone, `inner.sparse_grad_rows(batch)`),
        2. all-reduces the mask with MAX (the union over the replicas, identical everywhere; 250 KB at cfg5),
        3. lists the union's rows (the one host read of the step; it waits for (1)-(2) only, which finished long before
           the backward does).
    Then ONE all-reduce(sum) runs over a flat buffer [dense small gradients | running statistics | the union's rows of
    every sparse table, packed], and the summed rows are scattered back into the dense gradients the optimiser sweeps.
    Tables an inner step does not declare sparse (and everything under `sparse=False`) travel whole, as views into the
    same flat buffer.

    `inner` must expose grad_tensors() / stat_tensors() (lists of tensors), rebind(list, list) to accept views into the
    flat buffer, forward_backward(batch, normalizer) and optimizer_step(); optionally sparse_grad_rows(batch) ->
    [(index into grad_tensors(), int tensor of touched rows, duplicates allowed)]."""

    def __init__(self, inner, group=None, sparse=True):
        self.inner, self.group = inner, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.grads, self.stats = list(inner.grad_tensors()), list(inner.stat_tensors())
        self.sparse = bool(sparse) and hasattr(inner, "sparse_grad_rows")
        self.sparse_idx = sorted(inner.sparse_grad_indices()) if self.sparse else []
        self.dev, self.dt = self.grads[0].device, self.grads[0].dtype
        self._dense_idx = [i for i in range(len(self.grads)) if i not in self.sparse_idx]
        dense = [self.grads[i] for i in self._dense_idx] + self.stats
        self._sizes = [(t.numel() + 3) // 4 * 4 for t in dense]                   # 16-byte aligned sections
        self._n_dense_grad = sum(self._sizes[:len(self._dense_idx)])
        self._small = sum(self._sizes)
        self.flat = None
        self._bind(self._small)
        # mask over the rows of all sparse tables, back to back
        self._row0 = []
        n = 0
        for i in self.sparse_idx:
            self._row0.append(n)
            n += self.grads[i].shape[0]
        self.mask = torch.zeros(n, dtype=torch.uint8, device=self.dev) if self.sparse_idx else None
        self.side = torch.cuda.Stream(device=self.dev) if self.dev.type == "cuda" else None
        self.last_exchanged_elements = 0                                         # diagnostics: floats in the last all-reduce
        inner.seed = getattr(inner, "seed", 0) + 1000003 * self.rank             # independent dropout masks per replica

    def _bind(self, capacity):
        """(re)allocate the flat exchange buffer and point the inner step's dense gradients / statistics into its head"""
        old = self.flat
        self.flat = torch.zeros(capacity, dtype=self.dt, device=self.dev)
        grads, stats = list(self.inner.grad_tensors()), list(self.inner.stat_tensors())
        views, off = [], 0
        for t, n in zip([grads[i] for i in self._dense_idx] + stats, self._sizes):
            v = self.flat[off:off + t.numel()].view_as(t)
            v.copy_(t)
            views.append(v)
            off += n
        for k, i in enumerate(self._dense_idx):
            grads[i] = views[k]
        self.inner.rebind(grads, views[len(self._dense_idx):])
        self.grads, self.stats = grads, views[len(self._dense_idx):]
        del old

    def _union_rows(self, batch):
        """rows of every sparse table that ANY replica's batch touches: [(grad index, int64 rows)], identical on all ranks"""
        self.mask.zero_()
        for k, (i, rows) in enumerate(self.inner.sparse_grad_rows(batch)):
            assert i == self.sparse_idx[k]
            self.mask.index_fill_(0, rows.reshape(-1).long() + self._row0[k], 1)
        dist.all_reduce(self.mask, op=dist.ReduceOp.MAX, group=self.group)
        out = []
        for k, i in enumerate(self.sparse_idx):
            m = self.mask[self._row0[k]:self._row0[k] + self.grads[i].shape[0]]
            out.append((i, m.nonzero().squeeze(1)))                              # (host read: the sizes of the exchange)
        return out

    def forward_backward(self, batch, normalizer=None):
        """the inner step's forward + backward on this replica's batch, scaled so that the exchanged SUM is the mean"""
        if normalizer is None:
            normalizer = float(batch.B) * float(batch.n_candidates)
        return self.inner.forward_backward(batch, normalizer * self.world)

    def exchange(self, batch, loss=None, ready=None):
        """`ready`: an event recorded on the current stream BEFORE the forward / backward was issued (step() does that): the
        side stream's mask work needs the batch's ids, not the backward, and waits for that event only; without it the side
        stream joins behind everything issued so far (correct, no overlap)."""
        if self.world == 1:
            return
        if not self.sparse_idx:
            dist.all_reduce(self.flat, group=self.group)
            self.last_exchanged_elements = self.flat.numel()
        else:
            if self.side is not None:
                main = torch.cuda.current_stream(self.dev)
                if ready is not None:
                    self.side.wait_event(ready)
                else:
                    self.side.wait_stream(main)
                with torch.cuda.stream(self.side):                               # beside the backward still running on `main`
                    union = self._union_rows(batch)
                main.wait_stream(self.side)
                for _, rows in union:                   # allocated on the side stream, read by kernels of the current one:
                    rows.record_stream(main)            # the allocator must not hand the block out again before those ran
            else:
                union = self._union_rows(batch)
            need = self._small + sum(r.numel() * self.grads[i].shape[1] for i, r in union)
            if need > self.flat.numel():
                self._bind(max(need, 2 * (self.flat.numel() - self._small) + self._small))
            off = self._small
            spans = []
            for i, rows in union:
                g = self.grads[i]
                span = self.flat[off:off + rows.numel() * g.shape[1]].view(rows.numel(), g.shape[1])
                torch.index_select(g, 0, rows, out=span)
                spans.append((g, rows, span, i))
                off += span.numel()
            dist.all_reduce(self.flat[:off], group=self.group)
            self.last_exchanged_elements = off
            mark = getattr(self.inner, "mark_sparse_rows", None)
            for g, rows, span, i in spans:
                g.index_copy_(0, rows, span)
                if mark is not None:                    # rows other replicas touched: the optimizer's touched-row map must know
                    mark(i, rows)
        if self.stats:
            self.flat[self._n_dense_grad:self._small].mul_(1.0 / self.world)
        if loss is not None:
            dist.all_reduce(loss, group=self.group)

    def step(self, batch, normalizer=None):
        ready = torch.cuda.current_stream(self.dev).record_event() if (self.side is not None and self.world > 1) else None
        loss = self.forward_backward(batch, normalizer)
        self.exchange(batch, loss, ready)
        self.inner.optimizer_step()
        return loss

    def flush(self):
        """settle what the inner step defers (TokenPooledTrainStep.decay_window): call before reading its tables"""
        f = getattr(self.inner, "flush", None)
        if f is not None:
            f()
