"""Trainer-side surface of the hot path (openkge/trainer.py:32-113, :181-272; openkge/dataset.py:423-453).

``AddLossModule`` keeps the reference's constructor and ``forward`` signature and return triple
``(loss, hook_loss, all_outputs)``.  In training mode one fused HIP call computes the loss AND the gradients;
the returned loss is wired into autograd by a custom Function so the reference Trainer's
``(loss.sum() / normalizer).backward()`` (trainer.py:217-234) deposits them in ``weight.grad`` unchanged.
``compute_metrics`` mirrors OneToNMentionRelationDataset.compute_metrics on the HIP rank kernel.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.nn import BCEWithLogitsLoss, KLDivLoss

import ctypes
import os

from . import _native as N
from . import hotpath as H
from .metrics import MetricResult

CALLER_THREAD_BACKWARD = os.environ.get("OKGE_BACKWARD_ON_CALLER_THREAD", "1") == "1"
FAST_CALL = os.environ.get("OKGE_DROPIN_FAST", "1") == "1"      # persistent argument structs for the common training call


class _FusedLossFn(torch.autograd.Function):
    """Autograd node standing for AddLossModule's whole forward: the gradients were already produced by the fused kernel,
    multiplied by `applied` = the 1 / (B N) the reference Trainer divides the summed loss by (dataset.py:935, trainer.py:221).
    backward compares that with the upstream scalar autograd delivers -- on the device, no host read -- and rescales only
    if they differ (one launch that reads a single float when they agree; before: two passes over 11.6 MB)."""

    @staticmethod
    def forward(ctx, e_weight, r_weight, loss, g_e, g_r, engine, applied):
        ctx.g_e, ctx.g_r, ctx.engine, ctx.applied = g_e, g_r, engine, applied
        return loss.to(torch.float32).reshape(())               # (the cast already yields a fresh tensor)

    @staticmethod
    def backward(ctx, grad_out):
        alpha = grad_out.reshape(1).to(torch.float32).contiguous()
        g_e, g_r, ctx.g_e, ctx.g_r = ctx.g_e, ctx.g_r, None, None   # no reference left behind: AccumulateGrad then TAKES the
        ctx.engine.rescale_gradients_(g_e, g_r, alpha, ctx.applied) # buffers as .grad instead of copying them (an 11.6 MB
        return g_e, g_r, None, None, None, None, None               # copy per step at FB15k-237)


def _flat(t):
    return None if t is None else t.reshape(-1)


class _VirtualTablesLossFn(torch.autograd.Function):
    """Loss of already ENCODED rows (embedder variants that fall through to torch): the rows of one batch form two small
    virtual tables -- EV = [candidates | po objects | sp subjects], RV = [po relations | sp relations] -- the fused HIP
    step runs on them without dropout (it was applied by the encode) and returns the dense row gradients, which autograd
    carries back through the projection / batch-norm / normalisation into the parameters."""

    @staticmethod
    def forward(ctx, EV, RV, engine, scorer, vb, kind, smoothing, scores):
        EVc, RVc = EV.detach().contiguous(), RV.detach().contiguous()
        # every row of the two virtual tables is one candidate or one batch row: all of them are STORED by the call
        dEV, dRV = torch.empty_like(EVc), torch.empty_like(RVc)
        loss = engine.forward_backward(EVc, RVc, scorer, vb, dEV, dRV, loss=kind, label_smoothing=smoothing, normalizer=1.0,
                                       scores=scores, grads_zero=True, distinct_prefix_rows=True)
        ctx.engine, ctx.grads = engine, (dEV, dRV)
        return loss.to(torch.float32).reshape(())               # (the cast already yields a fresh tensor)

    @staticmethod
    def backward(ctx, grad_out):
        alpha = grad_out.reshape(1).to(torch.float32).contiguous()
        grads, ctx.grads = ctx.grads, None
        for g in grads:
            ctx.engine.scale_(g, alpha)
        return grads[0], grads[1], None, None, None, None, None, None


class _TokenPooledLossFn(torch.autograd.Function):
    """AddLossModule's forward for the token-pooled models: the fused step on the pooled rows + the pooling / batch-norm
    backward already left the dense gradients of the SUMMED loss in the slots; backward scales them by the upstream
    scalar and hands them to autograd in the order of `params`."""

    @staticmethod
    def forward(ctx, loss, engine, grads, *params):
        ctx.engine, ctx.grads = engine, grads
        return loss.to(torch.float32).reshape(()).clone()

    @staticmethod
    def backward(ctx, grad_out):
        alpha = grad_out.reshape(1).to(torch.float32).contiguous()
        for g in ctx.grads:
            ctx.engine.scale_(g, alpha)
        return (None, None, None) + tuple(ctx.grads)


class AddLossModule(nn.Module):
    """openkge/trainer.py:32-113.

    `training_outputs=False` (not in the reference's signature) skips materialising `all_outputs` in TRAINING mode and
    returns None in its place: the reference's Trainer reads the predictions only in evaluation (trainer.py:258-272), and
    the (B, N) block is the one thing the fused training kernels never need to write (29.8 MB per step at FB15k-237)."""

    def __init__(self, model, loss, bce_label_smoothing=0.0, training_outputs=True):
        super().__init__()
        self.model = model
        self.loss = loss
        self.bce_label_smoothing = bce_label_smoothing
        self.training_outputs = training_outputs
        if isinstance(loss, (BCEWithLogitsLoss, KLDivLoss)) and loss.reduction != "sum":
            # trainer.py:106 sums whatever the loss returns and scripts/train.py builds both losses with reduction='sum';
            # the fused kernels produce exactly that sum -- any other reduction would be silently different
            raise NotImplementedError(f"loss.reduction={loss.reduction!r}: the fused path implements reduction='sum' "
                                      f"(scripts/train.py:92-99)")

    def _loss_kind(self):
        if isinstance(self.loss, KLDivLoss):
            return "kl"
        if isinstance(self.loss, BCEWithLogitsLoss):
            return "bce"
        raise NotImplementedError(f"{self.loss} not supported. Please choose either BCEWithLogitsLoss or KLDivLoss")

    def forward(self, inputs, labels, use_batch_shared_entities, batch_shared_entities, epoch=-1,
                input_style_triple_or_prefix="triple"):
        allowed = ["triple", "right_and_left_prefix"]
        if input_style_triple_or_prefix not in allowed:
            raise Exception("input_style_triple_or_prefix not in {}".format(allowed))
        if input_style_triple_or_prefix != "right_and_left_prefix":
            return None                                            # trainer.py:64 has no else branch
        kind = self._loss_kind()
        m = self.model
        eng = m.engine()
        dev = m.entity_embedding.weight.device
        token_model = hasattr(m, "entity_token_ids")
        n_ent, first = m.train_data.entities_size, m.train_data.min_entities_size
        po, sp = inputs
        if FAST_CALL and not token_model and m.training and torch.is_grad_enabled() and isinstance(labels, tuple) \
                and not getattr(m, "encode_in_torch", False) and not H.VALIDATE:
            fast = self._fast_training_call(m, eng, dev, po, sp, labels, use_batch_shared_entities, batch_shared_entities, kind, epoch)
            if fast is not None:
                return fast
        batch = H.PrefixBatch()
        if po is not None:
            batch.po_rel, batch.po_obj = _flat(po[0]), _flat(po[1])
        if sp is not None:
            batch.sp_subj, batch.sp_rel = _flat(sp[0]), _flat(sp[1])
        # candidates: all entities in eval without batch sharing (trainer.py:77-78), else the shared id list
        if batch_shared_entities is None or (not use_batch_shared_entities and not m.training):
            batch.cand_first = first
            batch.n_cand = n_ent - batch.cand_first
        else:
            ids = batch_shared_entities.reshape(-1)
            if ids.numel() == n_ent - first and not use_batch_shared_entities:
                batch.cand_first, batch.n_cand = first, int(ids.numel())      # arange(vocab)[offset:] (dataset.py:872)
            else:
                batch.cand_ids, batch.n_cand = ids, int(ids.numel())
        if isinstance(labels, tuple):                               # (pos_row, pos_col) coordinates, sorted by column
            batch.pos_row, batch.pos_col = labels
            if H.VALIDATE and batch.pos_col.numel() > 1 and bool((batch.pos_col[1:] < batch.pos_col[:-1]).any()):
                raise ValueError("coordinate labels must be sorted by column (the kernels partition them by candidate tile)")
        else:                                                       # dense (B, N) {0,1} (dataset.py:885-932)
            batch.pos_row, batch.pos_col = H.positives_from_dense(labels.to(dev))
        B, n = batch.B, batch.n_cand
        want_grad = torch.is_grad_enabled() and m.training
        if want_grad and CALLER_THREAD_BACKWARD and torch.autograd.is_multithreading_enabled():
            # The caller's `loss.backward()` (trainer.py:226) hands a CUDA graph to the autograd engine's device thread and waits
            # for it: ~35 us of thread hand-off per step on a graph of ONE node whose backward only returns the gradients this
            # forward already computed (0.199 -> 0.162 ms per drop-in step, S-FB).  Run it on the calling thread instead
            # (thread-local autograd state; it stays set for this thread).  OKGE_BACKWARD_ON_CALLER_THREAD=0 leaves autograd alone.
            torch.autograd.set_multithreading_enabled(False)
        all_outputs = None
        if self.training_outputs or not m.training:
            all_outputs = torch.empty((B, (n + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :n]
        smoothing = self.bce_label_smoothing if kind == "bce" else 0.0
        if getattr(m, "encode_in_torch", False):
            return self._variant_result(m, batch, kind, smoothing, want_grad, all_outputs, epoch,
                                        all_entities=not use_batch_shared_entities and not m.training,
                                        per_direction=batch_shared_entities is None)
        # (read AFTER the variant dispatch: the variants fill the hook while they encode, trainer.py:97-98)
        hook_loss = m.after_batch_loss_hook(epoch) if hasattr(m, "after_batch_loss_hook") else None
        if token_model:
            return self._token_result(m, batch, kind, smoothing, want_grad, all_outputs, hook_loss)
        if m.training:
            m.dropout_step += 1
        batch.drop_cand = m.dropout_spec(H.STREAM_CAND)
        batch.drop_po_ent, batch.drop_sp_ent = m.dropout_spec(H.STREAM_PO_ENT), m.dropout_spec(H.STREAM_SP_ENT)
        batch.drop_po_rel, batch.drop_sp_rel = m.dropout_spec(H.STREAM_PO_REL, True), m.dropout_spec(H.STREAM_SP_REL, True)
        if want_grad:
            E, R = m.E, m.R
            # a contiguous candidate range: the call stores the candidate rows and clears the rest itself (all of dR, the
            # reserved rows in front of the candidates) inside its first launch -- fresh buffers, no fill launches.
            # An id list: candidate rows accumulate, so the buffers start from zero.
            clear = batch.cand_ids is None
            g_e, g_r = (torch.empty_like(E), torch.empty_like(R)) if clear else (torch.zeros_like(E), torch.zeros_like(R))
            # the factor the Trainer will apply (trainer.py:221: loss / normalizer_loss, normalizer_loss = B x N,
            # dataset.py:935) goes into the fused step; _FusedLossFn.backward checks it against what autograd delivers
            # (torch divides a fp32 tensor by a Python number as a multiplication by fp32(1) / fp32(number), forward and
            #  backward: `applied` is built the same way, and the normalizer handed to the library is the one whose fp32
            #  reciprocal is exactly that)
            applied = np.float32(1.0) / np.float32(float(B) * float(n))
            loss = eng.forward_backward(E, R, m.scorer_name, batch, g_e, g_r, loss=kind, label_smoothing=smoothing,
                                        normalizer=1.0 / float(applied), scores=all_outputs, grads_zero=True, clear_grads=clear)
            result = _FusedLossFn.apply(m.entity_embedding.weight, m.relation_embedding.weight, loss, g_e, g_r, eng,
                                        float(applied))
        else:
            loss = eng.forward_backward(m.E, m.R, m.scorer_name, batch, None, None, loss=kind, label_smoothing=smoothing,
                                        normalizer=1.0, scores=all_outputs, loss_only=True)
            result = loss.to(torch.float32).reshape(())
        return result, hook_loss, all_outputs

    def _fast_training_call(self, m, eng, dev, po, sp, labels, use_batch_shared_entities, batch_shared_entities, kind, epoch):
        """The call the reference Trainer makes thousands of times per epoch -- lookup model in training mode, coordinate labels,
        int32 id tensors on the device -- with PERSISTENT argument structs: the general path below builds a PrefixBatch, ten
        DropoutSpec objects and four ctypes structs per call, ~35 us of Python on a step whose kernels take 130 us (the path is
        host-bound).  Same library call, same flags, same buffers: test_dropin_path.py compares the two
        bit for bit.  Returns None when an input is not in that form (the general path then converts it)."""
        prow, pcol = labels
        ids = []
        for pair in (po, sp):
            if pair is not None:
                ids += [pair[0], pair[1]]
        for x in (prow, pcol, *ids):
            if x.dtype != torch.int32 or not x.is_contiguous() or x.device != dev:
                return None
        n_ent, first = m.train_data.entities_size, m.train_data.min_entities_size
        cand = None
        if batch_shared_entities is None:
            n = n_ent - first
        else:
            cand = batch_shared_entities
            if cand.dtype != torch.int32 or not cand.is_contiguous() or cand.device != dev:
                return None
            n = cand.numel()
            if n == n_ent - first and not use_batch_shared_entities:
                cand = None                                         # arange(vocab)[offset:] (dataset.py:872)
        fd = getattr(self, "_fd", None)
        if fd is None:
            t, pb, c, pos = N.Tables(), N.PrefixBatch(), N.Candidates(), N.Positives()
            pb.drop_po_ent.stream, pb.drop_sp_ent.stream, c.drop.stream = H.STREAM_PO_ENT, H.STREAM_SP_ENT, H.STREAM_CAND
            pb.drop_po_rel.stream, pb.drop_sp_rel.stream = H.STREAM_PO_REL, H.STREAM_SP_REL
            fd = self._fd = (t, pb, c, pos, (pb.drop_po_ent, pb.drop_sp_ent, c.drop), (pb.drop_po_rel, pb.drop_sp_rel))
        t, pb, c, pos, ent_drops, rel_drops = fd
        E, R = m.E, m.R
        if E.dtype != torch.float32 or not E.is_contiguous() or not R.is_contiguous() or R.dtype != torch.float32:
            return None
        t.E, t.R, t.n_ent, t.n_rel, t.d, t.scorer = E.data_ptr(), R.data_ptr(), E.shape[0], R.shape[0], E.shape[1], N.SCORERS[m.scorer_name]
        hook_loss = m.after_batch_loss_hook(epoch) if hasattr(m, "after_batch_loss_hook") else None
        m.dropout_step += 1
        seed, step = int(m.dropout_seed) & 0xFFFFFFFFFFFFFFFF, int(m.dropout_step) & 0xFFFFFFFF
        p_ent = float(m.keep_prob_dropout(m.input_dropout, m.dropout))
        p_rel = float(m.keep_prob_dropout(m.relation_input_dropout, m.relation_dropout))
        for d_ in ent_drops:
            d_.p, d_.seed, d_.step = p_ent, seed, step
        for d_ in rel_drops:
            d_.p, d_.seed, d_.step = p_rel, seed, step
        n_po = po[0].numel() if po is not None else 0
        n_sp = sp[0].numel() if sp is not None else 0
        pb.po_rel, pb.po_obj = (po[0].data_ptr(), po[1].data_ptr()) if n_po else (None, None)
        pb.sp_subj, pb.sp_rel = (sp[0].data_ptr(), sp[1].data_ptr()) if n_sp else (None, None)
        pb.n_po, pb.n_sp = n_po, n_sp
        c.ids, c.first_id, c.n = (None if cand is None else cand.data_ptr()), int(first), int(n)
        pos.row, pos.col, pos.nnz = prow.data_ptr(), pcol.data_ptr(), prow.numel()
        B = n_po + n_sp
        all_outputs = None
        if self.training_outputs:
            all_outputs = torch.empty((B, (n + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :n]
        clear = cand is None
        g_e, g_r = (torch.empty_like(E), torch.empty_like(R)) if clear else (torch.zeros_like(E), torch.zeros_like(R))
        applied = np.float32(1.0) / np.float32(float(B) * float(n))           # (see forward below)
        loss = torch.empty(1, dtype=torch.float64, device=dev)
        ws = eng.workspace(B, n, t.d)
        smoothing = self.bce_label_smoothing if kind == "bce" else 0.0
        N.check(eng.lib.okge_train_forward_backward(
            ctypes.byref(t), ctypes.byref(pb), ctypes.byref(c), ctypes.byref(pos), N.LOSSES[kind], float(smoothing), 1.0 / float(applied),
            N.OKGE_TRAIN_GRADS_ZERO | (N.OKGE_TRAIN_CLEAR_GRADS if clear else 0), loss.data_ptr(), g_e.data_ptr(), g_r.data_ptr(),
            None if all_outputs is None else all_outputs.data_ptr(), 0 if all_outputs is None else all_outputs.stride(0),
            ws.data_ptr(), eng._ws_bytes, eng._stream()), "okge_train_forward_backward")
        if CALLER_THREAD_BACKWARD and torch.autograd.is_multithreading_enabled():
            torch.autograd.set_multithreading_enabled(False)                   # (see forward below)
        result = _FusedLossFn.apply(m.entity_embedding.weight, m.relation_embedding.weight, loss, g_e, g_r, eng, float(applied))
        return result, hook_loss, all_outputs

    def _variant_result(self, m, batch, kind, smoothing, want_grad, all_outputs, epoch, all_entities, per_direction):
        """lookup embedder with batch_norm / projection / normalize / l2_reg on (model.py:463-479): torch encodes in the
        reference's call order and the HIP scorer / loss / backward works on the encoded rows.
        With a shared id list (trainer.py:75-84): candidates once -- get_all_obj() in eval without batch sharing, else
        precompute_batch_shared_inputs -- then (rel, obj) of the po rows, (subj, rel) of the sp rows (model.py:52-74).
        With batch_shared_entities=None (trainer.py:86-87) each prefix scorer encodes its OWN candidate block:
        po_prefix_score get_all_subj() first, sp_prefix_score get_all_obj() last (model.py:60-61, :71-72) -- two blocks
        that differ under project_entity (subj_ vs obj_projection), and two batch-norm / l2-hook passes either way; the
        two directions then run as two fused calls whose losses add (reduction='sum' is a plain sum over rows)."""
        dev = m.entity_embedding.weight.device
        n_po, n_sp, n_c = batch.n_po, batch.n_sp, batch.n_candidates
        ar = lambda a, b: torch.arange(a, b, dtype=torch.int32, device=dev)        # noqa: E731
        calls = []                                                                  # (EV, RV, virtual batch, output rows)
        with torch.set_grad_enabled(want_grad):
            if per_direction:
                # the positives of one direction WITHOUT a host synchronisation (boolean-mask indexing reads the count back):
                # the other direction's entries become padding -- (row -1, column INT32_MAX), which the kernels skip and which
                # sorts behind every candidate tile (okge.h, okge_positives) -- and a stable device sort restores column order
                pad_col = torch.iinfo(torch.int32).max

                def direction_positives(keep, row_shift):
                    col = torch.where(keep, batch.pos_col, torch.full_like(batch.pos_col, pad_col))
                    row = torch.where(keep, batch.pos_row - row_shift, torch.full_like(batch.pos_row, -1))
                    order = torch.argsort(col, stable=True)
                    return row[order].contiguous(), col[order].contiguous()
                in_po = batch.pos_row < n_po
                if n_po:
                    cand = m.get_all_subj().reshape(n_c, -1)
                    rel = m.encode_rel(batch.po_rel).reshape(n_po, -1)
                    obj = m.encode_obj(batch.po_obj).reshape(n_po, -1)
                    prow, pcol = direction_positives(in_po, 0)
                    vb = H.PrefixBatch(po_rel=ar(0, n_po), po_obj=ar(n_c, n_c + n_po), pos_row=prow, pos_col=pcol, cand_first=0, n_cand=n_c)
                    calls.append((torch.cat([cand, obj]), rel, vb, slice(0, n_po)))
                if n_sp:
                    subj = m.encode_subj(batch.sp_subj).reshape(n_sp, -1)
                    rel = m.encode_rel(batch.sp_rel).reshape(n_sp, -1)
                    cand = m.get_all_obj().reshape(n_c, -1)
                    prow, pcol = direction_positives(~in_po, n_po)
                    vb = H.PrefixBatch(sp_subj=ar(n_c, n_c + n_sp), sp_rel=ar(0, n_sp), pos_row=prow, pos_col=pcol, cand_first=0, n_cand=n_c)
                    calls.append((torch.cat([cand, subj]), rel, vb, slice(n_po, n_po + n_sp)))
            else:
                if all_entities:
                    cand = m.get_all_obj()
                elif batch.cand_ids is not None:
                    cand = m.precompute_batch_shared_inputs(batch.cand_ids)
                else:
                    cand = m.precompute_batch_shared_inputs(ar(batch.cand_first, batch.cand_first + n_c))
                parts_e, parts_r = [cand.reshape(n_c, -1)], []
                if n_po:
                    parts_r.append(m.encode_rel(batch.po_rel).reshape(n_po, -1))
                    parts_e.append(m.encode_obj(batch.po_obj).reshape(n_po, -1))
                if n_sp:
                    parts_e.append(m.encode_subj(batch.sp_subj).reshape(n_sp, -1))
                    parts_r.append(m.encode_rel(batch.sp_rel).reshape(n_sp, -1))
                vb = H.PrefixBatch(po_rel=ar(0, n_po) if n_po else None, po_obj=ar(n_c, n_c + n_po) if n_po else None,
                                   sp_subj=ar(n_c + n_po, n_c + n_po + n_sp) if n_sp else None,
                                   sp_rel=ar(n_po, n_po + n_sp) if n_sp else None,
                                   pos_row=batch.pos_row, pos_col=batch.pos_col, cand_first=0, n_cand=n_c)
                calls.append((torch.cat(parts_e), torch.cat(parts_r), vb, slice(0, n_po + n_sp)))
        hook_loss = m.after_batch_loss_hook(epoch)
        eng = m.engine()
        result = None
        for EV, RV, vb, rows in calls:
            out = all_outputs[rows] if all_outputs is not None else None
            if want_grad:
                part = _VirtualTablesLossFn.apply(EV, RV, eng, m.scorer_name, vb, kind, smoothing, out)
            else:
                part = eng.forward_backward(EV.detach().contiguous(), RV.detach().contiguous(), m.scorer_name, vb, None, None,
                                            loss=kind, label_smoothing=smoothing, normalizer=1.0, scores=out,
                                            loss_only=True).to(torch.float32).reshape(())
            result = part if result is None else result + part
        return result, hook_loss, all_outputs

    def _token_result(self, m, batch, kind, smoothing, want_grad, all_outputs, hook_loss):
        """UnigramPooling* models (model.py:716-796): the pooled rows of the batch form two small virtual tables, the
        fused step runs on them, and the pooling / batch-norm backward deposits dense token-table gradients."""
        if want_grad:
            st = m.autograd_step(kind, smoothing)
            loss = st.forward_backward(batch, normalizer=1.0, scores=all_outputs)
            params, grads = m.autograd_params_and_grads(st)
            result = _TokenPooledLossFn.apply(loss, m.engine(), grads, *params)
        else:
            loss = m.loss_only(batch, kind, smoothing, all_outputs)
            result = loss.to(torch.float32).reshape(())
        return result, hook_loss, all_outputs


def compute_metrics(filter_mask, label_ids, predictions, engine: H.HotPath = None) -> MetricResult:
    """OneToNMentionRelationDataset.compute_metrics (dataset.py:423-453) on the HIP rank kernel.

    filter_mask (B, N) bool; label_ids list[B] of list[G_b] of int tensors (candidate-relative ids);
    predictions (B, N) fp32 on the GPU.  Returns the same seven meters, weighted by number of groups per row."""
    dev = predictions.device
    engine = engine or H.HotPath(dev)
    row_ptr, grp_ptr, ids = [0], [0], []
    for groups in label_ids:
        for g in groups:
            ids.extend(int(x) for x in g.reshape(-1).tolist())
            grp_ptr.append(len(ids))
        row_ptr.append(len(grp_ptr) - 1)
    fm = filter_mask.to(dev)
    nz = fm.nonzero()                                            # sorted by row: CSR columns
    filt_ptr = torch.zeros(fm.shape[0] + 1, dtype=torch.int64, device=dev)
    filt_ptr[1:] = torch.cumsum(fm.sum(1), 0)
    filt_col = nz[:, 1].to(torch.int32).contiguous()
    t = lambda a, dt: torch.tensor(a, dtype=dt, device=dev)        # noqa: E731
    preds = predictions if predictions.stride(1) == 1 else predictions.contiguous()
    ranks = engine.filtered_ranks(preds, filt_ptr, filt_col, t(row_ptr, torch.int64), t(grp_ptr, torch.int64),
                                  t(ids, torch.int32)).cpu()
    result = MetricResult()
    for b in range(len(label_ids)):
        r = ranks[row_ptr[b]:row_ptr[b + 1]]
        n = int(r.numel())
        if n == 0:
            continue
        result["mrr"].update((1.0 / (r + 1).float()).sum().item() / n, n)
        result["mr"].update(r.sum().item() / n, n)
        for k in (50, 10, 3, 1):
            result[f"h{k}"].update((r < k).float().sum().item() / n, n)
    return result
