"""Training step with the entity table row-sharded over the GPUs of one node (one process per GPU,
torch.distributed backend "nccl" = RCCL over xGMI).

The reference has no counterpart (its only multi-device path is nn.DataParallel, openkge/trainer.py:143-145).
Every rank processes the SAME batch of prefixes against ITS OWN slice of the candidate entities
(tensor-parallel over the candidate axis); scores are independent per candidate and the BCE
loss is separable, so one step needs exactly two small exchanges:

    all-reduce(sum)  [B, d]     masked prefix entity rows (each row is non-zero on its owner); every rank then folds
                                them with its replicated relation rows into the query block itself
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

    def _set_dropout(self, batch):
        pe, pr, s, t = self.input_dropout, self.relation_input_dropout, self.seed, self.steps
        batch.drop_cand = H.DropoutSpec(pe, s, H.STREAM_CAND, t)
        batch.drop_po_ent = H.DropoutSpec(pe, s, H.STREAM_PO_ENT, t)
        batch.drop_sp_ent = H.DropoutSpec(pe, s, H.STREAM_SP_ENT, t)
        batch.drop_po_rel = H.DropoutSpec(pr, s, H.STREAM_PO_REL, t)
        batch.drop_sp_rel = H.DropoutSpec(pr, s, H.STREAM_SP_REL, t)

    def step(self, batch: H.PrefixBatch):
        """`batch` is the GLOBAL batch (identical on every rank); 1-vs-all candidates; positives carry global columns."""
        if batch.cand_ids is not None:
            raise NotImplementedError("batch-shared sampled candidates are too few to shard: use replicas")
        self.steps += 1
        self._set_dropout(batch)
        eng = self.engine
        # 1. masked entity rows of the prefixes whose entity lives here; sum over ranks = all rows; fold locally
        er = eng.encode_entity_rows(self.E, self.R, self.scorer, batch, self.shard)
        dist.all_reduce(er, group=self.group)
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
        dist.all_reduce(dq, group=self.group)
        # 3. chain rule: entity rows by their owner, relation rows everywhere (identical)
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
    dataset.py:423-446; exchange plan of SURVEY.md section 8e)."""

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
        self.col0 = c_lo - min_entities_size
        self.shard = H.Shard(self.ent_lo, self.ent_hi, self.col0)

    def local_scores(self, batch: H.PrefixBatch):
        eng = self.engine
        er = eng.encode_entity_rows(self.E, self.R, self.scorer, batch, self.shard)
        dist.all_reduce(er, group=self.group)
        qe = (eng.fold_queries(self.E, self.R, self.scorer, batch, er), er)
        local = H.PrefixBatch(po_rel=batch.po_rel, po_obj=batch.po_obj, sp_subj=batch.sp_subj, sp_rel=batch.sp_rel,
                              cand_first=self.cand_first_local, n_cand=self.n_cand_local)
        return eng.score_queries(self.E, self.R, self.scorer, qe[0], batch.B, local, self.shard)

    def ranks(self, batch: H.PrefixBatch, filt_ptr, filt_col, row_ptr, grp_ptr, ids):
        """int64 rank per answer group, identical on every rank.  Index arrays are global (positions in the full
        candidate list) and identical on every rank."""
        eng = self.engine
        x = self.local_scores(batch)
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
