"""Token-pooled embedder: UnigramPoolingRelationEmbedder + TokenBasedRelationEmbedder
(openkge/model.py:561-796) over the HIP kernels of csrc/okge_pool.hip, and the training step that drives the fused
prefix-scoring path on rows computed from tokens.

Design: the pooled (and batch-normed) rows of one batch -- candidates, po objects, sp subjects; po / sp relations -- are
written into two small "virtual" tables; the unchanged fused step (score -> loss -> backward, dropout included) runs on
those tables; its dense row gradients then go back through batch-norm and the pooling into the token tables' dense
gradients, and Adagrad sweeps the token tables and the batch-norm parameters.  Every `_encode` call of the reference
(candidates, po rows, sp rows -- trainer.py:75-91) normalises with ITS OWN batch statistics and updates the running
statistics; that order is kept.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _native as N
from . import hotpath as H
from .model import ComplexRelationScorer, DistmultRelationScorer, Models, RelationEmbedder

POOLS = {"sum": 0, "mean": 1, "max": 2}
BN_EPS, BN_MOMENTUM = 1e-5, 0.1                     # torch.nn.BatchNorm1d(momentum=0.1, eps=1e-5), model.py:611-612


def token_id_matrix(id_to_tokens_map, max_len, device="cpu"):
    """TokenBasedRelationEmbedder.__init__ (model.py:579-597): row i = the LAST max_len tokens of item i,
    right-padded with 0."""
    out = torch.zeros((len(id_to_tokens_map), max_len), dtype=torch.int32)
    for i, seq in enumerate(id_to_tokens_map):
        tail = list(seq)[-max_len:]
        out[i, :len(tail)] = torch.tensor(tail, dtype=torch.int32)
    return out.to(device)


class TokenSlot:
    """One embedder slot (entity or relation): token table, token-id matrix, optional batch-norm, gradients."""

    def __init__(self, W, token_ids, pool="sum", batchnorm=False, bn_weight=None, bn_bias=None):
        self.W, self.token_ids, self.pool = W, token_ids.to(torch.int32).contiguous(), pool
        d, dev = W.shape[1], W.device
        self.d = d
        # batch-norm parameters live in one flat buffer [weight | bias] so that one Adagrad launch covers them
        self.bn = None
        if batchnorm:
            self.bn = torch.empty(2 * d, dtype=torch.float32, device=dev)
            self.bn[:d] = torch.rand(d) if bn_weight is None else bn_weight        # init.uniform_(weight), model.py:613
            self.bn[d:] = 0.0 if bn_bias is None else bn_bias
            self.running_mean = torch.zeros(d, dtype=torch.float32, device=dev)
            self.running_var = torch.ones(d, dtype=torch.float32, device=dev)
            self.d_bn = torch.zeros(2 * d, dtype=torch.float32, device=dev)
            self.sum_bn = torch.zeros(2 * d, dtype=torch.float32, device=dev)
        self.dW = torch.zeros_like(W)
        self.sumW = torch.zeros_like(W)
        # touched-row map of dW for the optimizer sweep: the pooling backward stamps every row it writes, okge_adagrad_multi
        # skips the gradient rows that do not carry the current stamp (okge.h); the stamp moves on after every update
        self.touched = torch.zeros(W.shape[0], dtype=torch.uint8, device=dev) if d % 4 == 0 else None
        self.stamp = 1
        # optimizer steps every row has seen (TokenPooledTrainStep, decay_window > 1: deferred weight-decay-only updates)
        self.row_steps = torch.zeros(W.shape[0], dtype=torch.int32, device=dev) if d % 4 == 0 else None

    def next_stamp(self):
        self.stamp = self.stamp % 255 + 1

    @property
    def bn_weight(self):
        return None if self.bn is None else self.bn[:self.d]

    @property
    def bn_bias(self):
        return None if self.bn is None else self.bn[self.d:]

    def c(self):
        e = N.TokenEmbedder()
        e.W, e.token_ids = self.W.data_ptr(), self.token_ids.data_ptr()
        e.vocab, e.d, e.n_ids, e.max_len = self.W.shape[0], self.d, self.token_ids.shape[0], self.token_ids.shape[1]
        e.pool = POOLS[self.pool]
        if self.bn is not None:
            e.bn_weight, e.bn_bias = self.bn_weight.data_ptr(), self.bn_bias.data_ptr()
            e.bn_running_mean, e.bn_running_var = self.running_mean.data_ptr(), self.running_var.data_ptr()
            e.bn_eps, e.bn_momentum = BN_EPS, BN_MOMENTUM
        return e


class PoolEngine:
    """ctypes driver of okge_pool_encode / okge_pool_backward."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.lib = N.lib()
        self._ws, self._ws_bytes = None, 0
        self._state, self._state_bytes = None, 0

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, n, d):
        need = int(self.lib.okge_pool_workspace_bytes(n, d))
        if need > self._ws_bytes:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        return self._ws

    def encode(self, slot: TokenSlot, ids, first_id, n, training, raw, out, saved):
        """raw / out: (n, d) row blocks (views into the virtual tables); saved: (4*d,) or None without batch-norm"""
        if n == 0:
            return
        ws = self._workspace(n, slot.d)
        e = slot.c()
        N.check(self.lib.okge_pool_encode(ctypes.byref(e), None if ids is None else ids.data_ptr(), int(first_id), int(n),
                                          int(training), raw.data_ptr(), out.data_ptr(), raw.stride(0),
                                          None if saved is None else saved.data_ptr(), ws.data_ptr(), self._ws_bytes,
                                          self._stream()), "okge_pool_encode")

    def backward(self, slot: TokenSlot, ids, first_id, n, raw, d_out, saved):
        if n == 0:
            return
        ws = self._workspace(n, slot.d)
        e = slot.c()
        bn = slot.bn is not None
        N.check(self.lib.okge_pool_backward(ctypes.byref(e), None if ids is None else ids.data_ptr(), int(first_id), int(n),
                                            raw.data_ptr(), d_out.data_ptr(), raw.stride(0),
                                            saved.data_ptr() if bn else None, slot.dW.data_ptr(),
                                            slot.d_bn[:slot.d].data_ptr() if bn else None,
                                            slot.d_bn[slot.d:].data_ptr() if bn else None, ws.data_ptr(), self._ws_bytes,
                                            self._stream()), "okge_pool_backward")


    # -- a batch of _encode calls in one go (okge_pool_encode_calls / okge_pool_backward_calls) ----------------------
    def _calls(self, calls, backward, stamp_forward=False):
        """calls: [(slot, ids, first_id, n, raw, out_or_d_out, saved)] with n > 0 -> (ctypes array, keep-alive list)"""
        arr = (N.PoolCall * len(calls))()
        keep, need = [], 0
        for x, (slot, ids, first_id, n, raw, other, saved) in zip(arr, calls):
            e = slot.c()
            keep.append(e)
            x.e = ctypes.pointer(e)
            x.ids, x.first_id, x.n = None if ids is None else ids.data_ptr(), int(first_id), int(n)
            x.raw, x.ld = raw.data_ptr(), raw.stride(0)
            x.saved = None if saved is None else saved.data_ptr()
            if backward:
                x.d_out, x.dW = other.data_ptr(), slot.dW.data_ptr()
                if slot.bn is not None:
                    x.d_bn_weight, x.d_bn_bias = slot.d_bn[:slot.d].data_ptr(), slot.d_bn[slot.d:].data_ptr()
                touched = getattr(slot, "touched", None)
                if touched is not None:
                    x.row_touched, x.touched_stamp = touched.data_ptr(), int(slot.stamp)
            else:
                x.out = other.data_ptr()
                if stamp_forward and getattr(slot, "touched", None) is not None:      # the forward stamps every token row it reads
                    x.row_touched, x.touched_stamp = slot.touched.data_ptr(), int(slot.stamp)
            need += int(self.lib.okge_pool_workspace_bytes(int(n), slot.d))
        if backward and self.scatter_plan(calls):
            need = int(self.lib.okge_pool_backward_workspace_bytes(arr, len(calls)))
            state = int(self.lib.okge_pool_scatter_state_bytes(arr, len(calls)))
            if state > self._state_bytes:
                # zeroed ONCE: the kernels leave the state zero (okge.h); a graph capture must not allocate
                self._state = torch.zeros(state, dtype=torch.uint8, device=self.device)
                self._state_bytes = state
        if need > self._ws_bytes:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        return arr, keep

    def scatter_plan(self, calls):
        """store-and-sum scatter (bit-reproducible, okge.h) where it applies; OKGE_POOL_SCATTER=atomics keeps the old path"""
        if os.environ.get("OKGE_POOL_SCATTER", "plan") == "atomics":
            return False
        return all(c[0].pool != "max" and c[0].d % 4 == 0 for c in calls)

    def encode_calls(self, calls, training, stamp=False):
        calls = [c for c in calls if c[3] > 0]
        if not calls:
            return
        arr, keep = self._calls(calls, False, stamp_forward=stamp and training)
        N.check(self.lib.okge_pool_encode_calls(arr, len(calls), int(training), None if self._ws is None else self._ws.data_ptr(),
                                                self._ws_bytes, self._stream()), "okge_pool_encode_calls")
        del keep

    def catch_up_calls(self, calls, lazy, counters, lr, weight_decay, eps):
        """okge_pool_catch_up_calls: the token rows the calls name take their pending decay-only Adagrad steps (before the
        forward reads them).  calls: as encode_calls; lazy: hotpath.HotPath.lazy_tensors(...)"""
        calls = [c for c in calls if c[3] > 0]
        if not calls:
            return
        arr, keep = self._calls(calls, False)
        N.check(self.lib.okge_pool_catch_up_calls(arr, len(calls), lazy, len(lazy), counters.data_ptr(), float(lr), float(weight_decay),
                                                  float(eps), self._stream()), "okge_pool_catch_up_calls")
        del keep

    def backward_calls(self, calls):
        calls = [c for c in calls if c[3] > 0]
        if not calls:
            return
        arr, keep = self._calls(calls, True)
        plan = self.scatter_plan(calls)
        N.check(self.lib.okge_pool_backward_calls(arr, len(calls), None if self._ws is None else self._ws.data_ptr(), self._ws_bytes,
                                                  self._state.data_ptr() if plan else None, self._state_bytes if plan else 0,
                                                  self._stream()), "okge_pool_backward_calls")
        del keep


def _i32(t, dev):
    return None if t is None else t.reshape(-1).to(device=dev, dtype=torch.int32).contiguous()


class TokenPooledTrainStep:
    """forward + loss + backward + Adagrad for UnigramPooling{Complex,Distmult}RelationModel
    (Trainer.compute_one_batch, trainer.py:181-257, over model.py:762-796)."""

    def __init__(self, entity: TokenSlot, relation: TokenSlot, scorer, loss="bce", lr=0.1, weight_decay=1e-10, eps=1e-8,
                 label_smoothing=0.0, dropout=0.0, seed=0, engine=None, overlap_sweep=None, decay_window=None):
        self.entity, self.relation, self.scorer, self.loss = entity, relation, scorer, loss
        # decay_window (OKGE_LAZY_DECAY; 1 = every row every step): the reference's Adagrad moves EVERY token row in
        # every step by its weight-decay term (utils/optim.py:139-160) -- 1 GB of read-modify-write at configs[4], 175 us of a
        # 0.78 ms step, for rows nothing reads.  With a window W > 1 a row no batch names takes its pending decay-only steps
        # later, all at once in registers (okge_adagrad_lazy, okge.h): when a batch names it (catch-up before the forward),
        # when its turn in the rotating 1/W sweep comes, or at flush().  Same operations, same order: after flush() the tables
        # are bit-identical to W = 1 (test_lazy_decay_is_bit_equal_to_the_eager_sweep).  CONTRACT: between steps, rows no
        # batch named may lag by up to W - 1 decay-only steps; everything in this package that reads the tables (evaluation,
        # state_tensors / checkpoints, the module's encode methods and state_dict) calls flush() first -- do the same before
        # reading .W / .sumW directly, and before DISCARDING a step object whose tables live on (its pending steps go with it).
        # Default: window 8 for token tables of 96 MB and more, 1 below -- the deferral trades the sweep's HBM time (0.7 us per MB of
        # table) for a catch-up launch and replay arithmetic that do not shrink with the table: at 45 MB it LOSES 40 us per step
        # (tools/soak_lazy.py), at configs[4]'s 256 MB it wins 130.
        if decay_window is None:
            env = os.environ.get("OKGE_LAZY_DECAY")
            big = (entity.W.numel() + relation.W.numel()) * 4 >= (96 << 20)
            decay_window = int(env) if env else (8 if big else 1)
        lazy_ok = all(getattr(sl, "row_steps", None) is not None and getattr(sl, "touched", None) is not None for sl in (entity, relation))
        self.decay_window = max(1, int(decay_window)) if lazy_ok else 1
        self._counters = torch.zeros(2, dtype=torch.int32, device=entity.W.device)      # [optimizer steps taken, scratch]
        self._pending = None              # (lr, weight_decay, eps) of the deferred steps; None: every row is current
        # overlap_sweep (OKGE_OVERLAP_SWEEP=1; an experiment, OFF by default): the Adagrad update of the token rows NO token of the
        # batch names (85 % of them at configs[4]; the reference's weight decay reaches every row: 0.9 GB of read-modify-write per
        # step) on a side stream BESIDE the step's matrix kernels -- the pooling forward stamps the rows it reads, so the others are
        # known right behind it and nothing of the step reads or writes them.  Same arithmetic row for row (rows = 1 then rows = 2
        # of okge_adagrad_multi; bit-equal tables: test_overlapped_sweep_is_bit_equal_to_the_plain_step).  Measured at configs[4]:
        # the late sweep shrinks 178 -> ~25 us, but the fused tile kernel beside the side sweep stretches 292 -> 398 us and the
        # encode launch 15 -> 43 us: 0.826 -> 0.852 ms cold, 0.725 -> 0.763 warm -- a loss (fp32 MFMA shares the SIMD's issue
        # with the sweep's sqrt / div VALU work, profiles/round4_ablation.md section 5), like the S-FB attempt of round 2.
        if overlap_sweep is None:
            overlap_sweep = os.environ.get("OKGE_OVERLAP_SWEEP", "0") == "1"
        self.overlap_sweep = bool(overlap_sweep)
        self._side, self._side_done = None, None
        self.lr, self.weight_decay, self.eps, self.label_smoothing = lr, weight_decay, eps, label_smoothing
        self.dropout, self.seed, self.steps = dropout, seed, 0
        self.device = entity.W.device
        self.engine = engine or H.HotPath(self.device)
        self.pool = PoolEngine(self.device)
        # (rounds 1-2 ran the five encode / backward calls of a step one by one, the relation slot's on a side stream:
        #  35 small launches; they now go to the library as ONE batch each way: three launches forward, three backward)
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.step_dev = None              # device step counter, attached by GraphedTrainStep
        self._rows = 0

    def state_tensors(self):
        self.flush()
        out = []
        for sl in (self.entity, self.relation):
            out += [sl.W, sl.dW, sl.sumW]
            if sl.bn is not None:
                out += [sl.bn, sl.d_bn, sl.sum_bn, sl.running_mean, sl.running_var]
        if self.decay_window > 1:         # (a caller that restores a snapshot restores the step counters with it)
            out += [self.entity.row_steps, self.relation.row_steps, self._counters]
        return out

    # -- deferred weight-decay-only updates (decay_window > 1) ---------------------------------------------------------
    def _hparams(self):
        return (float(self.lr), float(self.weight_decay), float(self.eps))

    def _lazy_tables(self):
        return [(sl.W, sl.dW, sl.sumW, sl.row_steps, sl.touched, sl.stamp) for sl in (self.entity, self.relation)]

    def flush(self):
        """every token row takes the decay-only steps it still owes: afterwards the tables are those of the eager sweep"""
        if self._pending is None:
            return
        lr, wd, eps = self._pending
        self.engine.adagrad_lazy(self._lazy_tables(), self._counters, self.decay_window, True, lr, wd, eps)
        self._pending = None

    def mark_pending(self):
        """a captured graph holding this step was replayed: rows may owe steps again (GraphedTrainStep)"""
        if self.decay_window > 1:
            self._pending = self._hparams()

    def _settle_hparams(self):
        """pending steps were taken with the lr / weight decay / eps of their time: flush before these change"""
        if self._pending is not None and self._pending != self._hparams():
            self.flush()

    # -- sharded.ReplicaStep protocol: gradients / running statistics as views into one flat exchange buffer ---------
    def grad_tensors(self):
        out = []
        for sl in (self.entity, self.relation):
            out.append(sl.dW)
            if sl.bn is not None:
                out.append(sl.d_bn)
        return out

    def stat_tensors(self):
        out = []
        for sl in (self.entity, self.relation):
            if sl.bn is not None:
                out += [sl.running_mean, sl.running_var]
        return out

    def sparse_grad_indices(self):
        """positions in grad_tensors() of the row-sparse gradients (the token tables')"""
        return [0, 2 if self.entity.bn is not None else 1]

    def sparse_grad_rows(self, batch: H.PrefixBatch):
        """token-table rows this batch's backward can touch (with repeats; row 0 = padding included, its gradient is 0):
        the token ids of the candidate + prefix entities, and of the prefix relations -- known from the ids alone"""
        dev = self.device
        cand = _i32(batch.cand_ids, dev) if batch.cand_ids is not None else \
            torch.arange(batch.cand_first, batch.cand_first + batch.n_candidates, dtype=torch.int32, device=dev)
        ent = torch.cat([x for x in (cand, _i32(batch.po_obj, dev), _i32(batch.sp_subj, dev)) if x is not None]).long()
        rel = torch.cat([x for x in (_i32(batch.po_rel, dev), _i32(batch.sp_rel, dev)) if x is not None]).long()
        ie, ir = self.sparse_grad_indices()
        return [(ie, self.entity.token_ids.index_select(0, ent)), (ir, self.relation.token_ids.index_select(0, rel))]

    def rebind(self, grads, stats):
        gi, si = iter(grads), iter(stats)
        for sl in (self.entity, self.relation):
            sl.dW = next(gi)
            if sl.bn is not None:
                sl.d_bn = next(gi)
                sl.running_mean, sl.running_var = next(si), next(si)

    def _buffers(self, n_ent_rows, n_rel_rows):
        d = self.entity.d
        if n_ent_rows > self._rows or n_rel_rows > getattr(self, "_rrows", 0):
            dev = self.device
            self._rows, self._rrows = n_ent_rows, n_rel_rows
            self.EV, self.EX, self.dEV = (torch.zeros((n_ent_rows, d), device=dev) for _ in range(3))
            self.RV, self.RX, self.dRV = (torch.zeros((n_rel_rows, d), device=dev) for _ in range(3))
            self.saved = torch.zeros((5, 4 * d), device=dev)
        return self.EV, self.EX, self.dEV, self.RV, self.RX, self.dRV

    def step(self, batch: H.PrefixBatch, normalizer=None):
        """`batch` carries ENTITY / RELATION ids exactly as for the lookup models."""
        self._in_step = True
        try:
            loss = self.forward_backward(batch, normalizer)
        finally:
            self._in_step = False
        self.optimizer_step()
        return loss

    def forward_backward(self, batch: H.PrefixBatch, normalizer=None, scores=None):
        """Leaves the dense gradients in entity/relation .dW and .d_bn ([d weight | d bias])."""
        self.steps += 1
        dev = self.device
        n_po, n_sp, N_c = batch.n_po, batch.n_sp, batch.n_candidates
        B = n_po + n_sp
        EV, EX, dEV, RV, RX, dRV = self._buffers(N_c + B, B)
        ent, rel, pe = self.entity, self.relation, self.pool
        sv = self.saved
        bn_e, bn_r = ent.bn is not None, rel.bn is not None
        # the reference's encode order: candidates, (po rel, po obj), (sp subj, sp rel)   -- trainer.py:75-91
        calls = [(ent, _i32(batch.cand_ids, dev), batch.cand_first, N_c, EX[:N_c], EV[:N_c], dEV[:N_c], sv[0] if bn_e else None),
                 (rel, _i32(batch.po_rel, dev), 0, n_po, RX[:n_po], RV[:n_po], dRV[:n_po], sv[1] if bn_r else None),
                 (ent, _i32(batch.po_obj, dev), 0, n_po, EX[N_c:N_c + n_po], EV[N_c:N_c + n_po], dEV[N_c:N_c + n_po], sv[2] if bn_e else None),
                 (ent, _i32(batch.sp_subj, dev), 0, n_sp, EX[N_c + n_po:N_c + B], EV[N_c + n_po:N_c + B], dEV[N_c + n_po:N_c + B], sv[3] if bn_e else None),
                 (rel, _i32(batch.sp_rel, dev), 0, n_sp, RX[n_po:B], RV[n_po:B], dRV[n_po:B], sv[4] if bn_r else None)]
        # forward of all five calls: raw pooled rows -> EX / RX, batch-normed rows -> EV / RV (per-call statistics, running
        # statistics updated in this order)
        # (only inside step(): a caller that runs forward_backward alone -- the autograd bridge, ReplicaStep, whose other replicas'
        #  rows receive gradients in the exchange -- gets no early update)
        enc_calls = [(c_[0], c_[1], c_[2], c_[3], c_[4], c_[5] if c_[0].bn is not None else c_[4], c_[7]) for c_ in calls]
        if self.decay_window > 1:
            # the token rows this batch names take the decay-only steps they owe before the forward reads them (always
            # launched: a captured graph must hold it; with nothing pending it reads the batch's step counters and ends)
            self._settle_hparams()
            lr, wd, eps = self._hparams()
            pe.catch_up_calls(enc_calls, self.engine.lazy_tensors(self._lazy_tables()), self._counters, lr, wd, eps)
        overlap = (self.overlap_sweep and self.decay_window == 1 and getattr(self, "_in_step", False) and ent.touched is not None
                   and rel.touched is not None and not torch.cuda.is_current_stream_capturing())
        if self._side_done is not None:            # the previous step's side sweep wrote rows this forward may read
            torch.cuda.current_stream(dev).wait_event(self._side_done)
            self._side_done = None
        pe.encode_calls(enc_calls, True, stamp=overlap)
        self._early_swept = False
        if overlap:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self._side.wait_event(main.record_event())       # the maps are complete: every row without the stamp is free
            with torch.cuda.stream(self._side):
                self.engine.adagrad_multi([(sl.W, sl.dW, sl.sumW, sl.touched, sl.stamp, 1) for sl in (ent, rel)],
                                          self.lr, self.weight_decay, self.eps)
                self._side_done = self._side.record_event()
            self._early_swept = True
        EVt, RVt = (EV if bn_e else EX), (RV if bn_r else RX)
        # the fused step on the virtual tables: candidates are rows 0..N-1, prefix entities follow
        # row indices of the virtual tables: they depend on the batch's shape only -- built once per shape (four arange
        # launches per step otherwise: 18 us of a 0.9 ms step at configs[4])
        key = (N_c, n_po, n_sp)
        if getattr(self, "_ar_key", None) != key:
            rng = torch.arange(0, N_c + B, dtype=torch.int32, device=dev)
            self._ar_key, self._ar = key, (rng[:n_po], rng[N_c:N_c + n_po], rng[N_c + n_po:N_c + B], rng[n_po:B])
        ar_po_rel, ar_po_obj, ar_sp_subj, ar_sp_rel = self._ar
        p, s, t = self.dropout, self.seed, self.steps
        DS = lambda stream: H.DropoutSpec(p, s, stream, t, step_dev=self.step_dev)      # noqa: E731
        vb = H.PrefixBatch(po_rel=ar_po_rel if n_po else None, po_obj=ar_po_obj if n_po else None,
                           sp_subj=ar_sp_subj if n_sp else None, sp_rel=ar_sp_rel if n_sp else None,
                           pos_row=batch.pos_row, pos_col=batch.pos_col, cand_first=0, n_cand=N_c,
                           drop_cand=DS(H.STREAM_CAND), drop_po_ent=DS(H.STREAM_PO_ENT), drop_sp_ent=DS(H.STREAM_SP_ENT),
                           drop_po_rel=DS(H.STREAM_PO_REL), drop_sp_rel=DS(H.STREAM_SP_REL))
        self.engine.forward_backward(EVt[:N_c + B], RVt[:B], self.scorer, vb, dEV[:N_c + B], dRV[:B], loss=self.loss,
                                     label_smoothing=self.label_smoothing, normalizer=normalizer, loss_out=self.loss_out,
                                     scores=scores, grads_zero=True, distinct_prefix_rows=True)
        # (every row of dEV / dRV is STORED by that call -- the candidate rows by the tile kernel (grads_zero), the prefix rows,
        #  one per batch row in these virtual tables, by the prefix backward (distinct_prefix_rows): nothing to clear per step)
        pe.backward_calls([(c_[0], c_[1], c_[2], c_[3], c_[4], c_[6], c_[7]) for c_ in calls])
        return self.loss_out

    def mark_sparse_rows(self, index, rows):
        """sharded.ReplicaStep wrote other replicas' gradient rows into a sparse gradient (grad_tensors()[index]): stamp them"""
        sl = self.entity if index == self.sparse_grad_indices()[0] else self.relation
        if sl.touched is not None:
            sl.touched.index_fill_(0, rows.reshape(-1).long(), sl.stamp)

    def optimizer_step(self):
        eng, e, r = self.engine, self.entity, self.relation
        if self.decay_window > 1:
            # stamped rows: what they owe + this step with their gradient; one row in decay_window of the others: what they
            # owe + this step; batch-norm parameters: dense; the device step counter moves on (okge_adagrad_lazy)
            self._settle_hparams()
            lr, wd, eps = self._hparams()
            tensors = self._lazy_tables() + [(sl.bn, sl.d_bn, sl.sum_bn) for sl in (e, r) if sl.bn is not None]
            eng.adagrad_lazy(tensors, self._counters, self.decay_window, False, lr, wd, eps)
            self._pending = (lr, wd, eps)
            self._after_update()
            return
        # one launch: token tables (gradient rows the backward did not stamp are neither read nor cleared) + batch-norm parameters
        # (after an early sweep of the unstamped rows -- forward_backward, overlap_sweep -- only the stamped rows are left)
        rows = 2 if getattr(self, "_early_swept", False) else 0
        tensors = [(sl.W, sl.dW, sl.sumW, sl.touched, sl.stamp, rows) for sl in (e, r)]
        tensors += [(sl.bn, sl.d_bn, sl.sum_bn) for sl in (e, r) if sl.bn is not None]
        self._early_swept = False
        eng.adagrad_multi(tensors, self.lr, self.weight_decay, self.eps)
        self._after_update()

    def _after_update(self):
        for sl in (self.entity, self.relation):
            sl.next_stamp()
        for sl, bn in getattr(self, "module_batchnorms", ()):          # keep an attached nn.Module's parameters current
            bn.weight.data.copy_(sl.bn_weight)
            bn.bias.data.copy_(sl.bn_bias)
            # ... and its running statistics, once ReplicaStep.rebind has moved ours into the exchange buffer (the module's
            # eval-mode encode, state_dict and checkpoints read the module's buffers)
            if sl.running_mean.data_ptr() != bn.running_mean.data_ptr():
                bn.running_mean.copy_(sl.running_mean)
                bn.running_var.copy_(sl.running_var)


# ------------------------------------------------------------------------------------------------------------------
# API-compatible model classes (inference protocol; training goes through TokenPooledTrainStep)
# ------------------------------------------------------------------------------------------------------------------
def _flush_before_state_dict(module, prefix, keep_vars):
    module.flush_steps()


class UnigramPoolingRelationEmbedder(RelationEmbedder):
    """openkge/model.py:561-796.  Implemented: pool sum|mean|max, normalize None|'batchnorm', dropout; not implemented
    (raise): normalize='norm', activation, project_relation, sparse gradients.  Training: TokenPooledTrainStep (fused, own
    Adagrad) or trainer.AddLossModule (autograd bridge: any torch optimizer over the module's parameters)."""

    def __init__(self, entity_slot_size, relation_slot_size, train_data, pool='sum', normalize=None, dropout=0.0,
                 entity_dropout=None, relation_dropout=None, sparse=False, init_std=0.01, activation=None,
                 project_relation=False, seed=0):
        super().__init__()
        if normalize not in (None, 'batchnorm') or activation is not None or project_relation or sparse:
            raise NotImplementedError("token-pooled path implements pool sum/mean/max with optional batchnorm only")
        if relation_slot_size is None or relation_slot_size <= 0:
            relation_slot_size = entity_slot_size
        if relation_slot_size != entity_slot_size:
            raise NotImplementedError("relation slot size must equal the entity slot size without a relation projection")
        self.train_data, self.slot_size, self.pool, self.normalize = train_data, entity_slot_size, pool, normalize
        self.entity_dropout = entity_dropout if entity_dropout else dropout            # model.py:757-758
        self.relation_dropout = relation_dropout if relation_dropout else dropout
        self.register_buffer('entity_token_ids', token_id_matrix(train_data.entity_id_to_tokens_map, train_data.max_length[0]))
        self.register_buffer('relation_token_ids', token_id_matrix(train_data.relation_id_to_tokens_map, train_data.max_length[1]))
        self.entity_embedding = torch.nn.Embedding(train_data.entity_tokens_size, entity_slot_size, padding_idx=0)
        self.relation_embedding = torch.nn.Embedding(train_data.relation_tokens_size, entity_slot_size, padding_idx=0)
        torch.nn.init.normal_(self.entity_embedding.weight.data, std=init_std)         # model.py:660-661 (row 0 included)
        torch.nn.init.normal_(self.relation_embedding.weight.data, std=init_std)
        self.entity_batchnorm = self.relation_batchnorm = None
        if normalize == 'batchnorm':
            self.entity_batchnorm = torch.nn.BatchNorm1d(entity_slot_size, momentum=BN_MOMENTUM, eps=BN_EPS)
            self.relation_batchnorm = torch.nn.BatchNorm1d(entity_slot_size, momentum=BN_MOMENTUM, eps=BN_EPS)
            torch.nn.init.uniform_(self.entity_batchnorm.weight)
            torch.nn.init.uniform_(self.relation_batchnorm.weight)
        self.entity_projection = self.relation_projection = None                      # the attribute model.py:789 reads
        self.entity_embedding_from_tokens = self.relations_embedding_from_tokens = None
        self.dropout_seed, self.dropout_step = seed, 0
        self._pool_engine = self._engine = None
        # training drivers that update this module's parameters in place (train_step()): token rows no batch named may owe
        # decay-only Adagrad steps (TokenPooledTrainStep.decay_window) -- every reader below settles them first
        self._steps = []
        self.register_state_dict_pre_hook(_flush_before_state_dict)

    def __getstate__(self):                              # (weak references do not pickle; a copy starts without drivers)
        state = self.__dict__.copy()
        state["_steps"] = []
        return state

    def flush_steps(self):
        for ref in self._steps:
            st = ref()
            if st is not None:
                st.flush()

    # -- plumbing ----------------------------------------------------------------------------------------------
    def engine(self):
        dev = self.entity_embedding.weight.device
        if self._engine is None or self._engine.device != dev:
            self._engine, self._pool_engine = H.HotPath(dev), PoolEngine(dev)
        return self._engine

    def _slot(self, relation):
        emb, tok, bn = (self.relation_embedding, self.relation_token_ids, self.relation_batchnorm) if relation else \
            (self.entity_embedding, self.entity_token_ids, self.entity_batchnorm)
        s = TokenSlot.__new__(TokenSlot)
        s.W, s.token_ids, s.pool, s.d, s.bn = emb.weight.data, tok, self.pool, self.slot_size, None
        if bn is not None:
            s.bn = torch.cat([bn.weight.data, bn.bias.data])
            s.running_mean, s.running_var = bn.running_mean, bn.running_var
        return s

    def _encode(self, ids, relation, stream):
        """pool -> batch-norm (batch statistics in training mode, running statistics otherwise) -> dropout -> (n,1,d)"""
        eng = self.engine()
        self.flush_steps()
        ids = ids.reshape(-1).to(torch.int32).contiguous()
        n, d = ids.numel(), self.slot_size
        emb, tok, bn = (self.relation_embedding, self.relation_token_ids, self.relation_batchnorm) if relation else \
            (self.entity_embedding, self.entity_token_ids, self.entity_batchnorm)
        p = (self.relation_dropout if relation else self.entity_dropout) if self.training else 0.0
        if torch.is_grad_enabled() and (emb.weight.requires_grad or (bn is not None and bn.weight.requires_grad)):
            # called with gradients enabled outside AddLossModule (a caller's own loss): the reference's op sequence in
            # torch (model.py:762-786; differentiable, the BatchNorm1d module keeps its own running statistics), dropout by
            # the Philox mask kernel (autograd_score.MaskRowsFn)
            from . import autograd_score as AG
            tokens = tok[ids.long()].long()
            embedded = emb(tokens)
            if self.pool == 'max':
                encoded, _ = embedded.max(dim=1)
            elif self.pool == 'mean':
                encoded = embedded.sum(dim=1) / ((tokens > 0).float().sum(1, keepdim=True) + 1e-12)
            else:
                encoded = embedded.sum(dim=1)
            if bn is not None:
                encoded = bn(encoded.contiguous())
            if p > 0:
                encoded = AG.MaskRowsFn.apply(encoded, eng, H.DropoutSpec(p, self.dropout_seed, stream, self.dropout_step))
            return encoded.unsqueeze(1)
        slot = self._slot(relation)
        raw = torch.empty((n, d), device=ids.device)
        out = torch.empty_like(raw) if slot.bn is not None else raw
        saved = torch.empty(4 * d, device=ids.device) if slot.bn is not None else None
        self._pool_engine.encode(slot, ids, 0, n, self.training, raw, out, saved)
        if p > 0:
            out = eng.encode_rows(out, None, 0, n, H.DropoutSpec(p, self.dropout_seed, stream, self.dropout_step))
        return out.unsqueeze(1)

    def encode_subj(self, subj):
        return self._encode(subj, False, H.STREAM_SP_ENT)

    def encode_obj(self, obj):
        return self._encode(obj, False, H.STREAM_PO_ENT)

    def encode_rel(self, rel):
        return self._encode(rel, True, H.STREAM_SP_REL)

    # -- TokenBasedRelationEmbedder: tables precomputed from tokens (model.py:624-712) -----------------------------
    def train(self, mode=True):
        self.entity_embedding_from_tokens = self.relations_embedding_from_tokens = None
        return super().train(mode)

    def precompute_embeddings_from_tokens(self):
        if self.entity_embedding_from_tokens is None:
            was_training = self.training
            super().train(False)                       # the reference calls self.eval() here and stays in eval mode
            dev = self.entity_embedding.weight.device
            ar = lambda n: torch.arange(n, dtype=torch.int32, device=dev)      # noqa: E731
            with torch.no_grad():                      # model.py:682: the precomputed tables carry no graph
                self.entity_embedding_from_tokens = self._encode(ar(self.train_data.entities_size), False, H.STREAM_CAND).squeeze(1)
                self.relations_embedding_from_tokens = self._encode(ar(self.train_data.relations_size), True, H.STREAM_SP_REL).squeeze(1)
            del was_training

    def get_all_subj(self):
        self.precompute_embeddings_from_tokens()
        return self.entity_embedding_from_tokens[self.train_data.min_entities_size:]

    get_all_obj = get_all_subj

    def get_all_rel(self):
        self.precompute_embeddings_from_tokens()
        return self.relations_embedding_from_tokens[self.train_data.min_entities_size:]        # sic, model.py:630

    def get_subj(self, subj):
        self.precompute_embeddings_from_tokens()
        return self.entity_embedding_from_tokens[subj].unsqueeze(0)

    get_obj = get_subj

    def get_rel(self, rel):
        self.precompute_embeddings_from_tokens()
        return self.relations_embedding_from_tokens[rel]

    def get_slot_size(self):
        return self.slot_size

    # -- prefix scoring: RelationScorer.sp_prefix_score / po_prefix_score (model.py:52-74) ------------------------------
    def _prefix_score(self, batch: H.PrefixBatch, many=None):
        if many is None:
            many = self.get_all_obj()
        if batch.sp_subj is not None:
            return self._score(self.encode_subj(batch.sp_subj), self.encode_rel(batch.sp_rel), many, prefix=True, sp=True, po=False)
        return self._score(many, self.encode_rel(batch.po_rel), self.encode_obj(batch.po_obj), prefix=True, sp=False, po=True)

    # -- AddLossModule / autograd bridge (the reference Trainer's path: trainer.py:142, 206-234) ---------------------
    def _module_slots(self):
        slots = []
        for emb, tok, bn in ((self.entity_embedding, self.entity_token_ids, self.entity_batchnorm),
                             (self.relation_embedding, self.relation_token_ids, self.relation_batchnorm)):
            s = TokenSlot(emb.weight.data, tok, self.pool, bn is not None, None if bn is None else bn.weight.data,
                          None if bn is None else bn.bias.data)
            if bn is not None:
                s.running_mean, s.running_var = bn.running_mean, bn.running_var
            slots.append(s)
        return slots

    def autograd_step(self, loss, label_smoothing):
        """the cached TokenPooledTrainStep behind AddLossModule: shares the module's parameters; its optimizer is NOT used
        (the caller's torch optimizer steps the module parameters)"""
        self.flush_steps()
        st = getattr(self, "_ag_step", None)
        if st is None or st.loss != loss or st.label_smoothing != label_smoothing or st.entity.W.data_ptr() != self.entity_embedding.weight.data_ptr():
            e, r = self._module_slots()
            # (decay_window = 1: this step's own optimizer never runs -- the caller's torch optimizer moves every row every step)
            st = self._ag_step = TokenPooledTrainStep(e, r, self.scorer_name, loss=loss, label_smoothing=label_smoothing,
                                                      dropout=self.entity_dropout, seed=self.dropout_seed, decay_window=1)
        for sl, bn in ((st.entity, self.entity_batchnorm), (st.relation, self.relation_batchnorm)):
            sl.dW = torch.zeros_like(sl.W)                        # fresh gradient buffers: the last ones went to autograd
            if bn is not None:                                     # the module's parameters may have been stepped outside
                sl.bn[:sl.d].copy_(bn.weight.data)
                sl.bn[sl.d:].copy_(bn.bias.data)
                sl.d_bn = torch.zeros_like(sl.d_bn)
        st.steps = self.dropout_step
        self.dropout_step += 1
        return st

    def autograd_params_and_grads(self, st):
        params, grads = [self.entity_embedding.weight, self.relation_embedding.weight], [st.entity.dW, st.relation.dW]
        for sl, bn in ((st.entity, self.entity_batchnorm), (st.relation, self.relation_batchnorm)):
            if bn is not None:
                params += [bn.weight, bn.bias]
                grads += [sl.d_bn[:sl.d], sl.d_bn[sl.d:]]
        return params, grads

    def loss_only(self, batch: H.PrefixBatch, loss, label_smoothing, scores=None):
        """eval-mode loss (+ scores) of AddLossModule: rows encoded with the running statistics, no dropout"""
        eng, dev = self.engine(), self.entity_embedding.weight.device
        n_po, n_sp, n_c = batch.n_po, batch.n_sp, batch.n_candidates
        if batch.cand_ids is None:
            cand = self.get_all_obj()[batch.cand_first - self.train_data.min_entities_size:][:n_c] if not self.training else \
                self._encode(torch.arange(batch.cand_first, batch.cand_first + n_c, dtype=torch.int32, device=dev), False, H.STREAM_CAND).squeeze(1)
        else:
            cand = self._encode(batch.cand_ids, False, H.STREAM_CAND).squeeze(1)
        parts_e, parts_r = [cand], []
        if n_po:
            parts_r.append(self.encode_rel(batch.po_rel).squeeze(1))
            parts_e.append(self.encode_obj(batch.po_obj).squeeze(1))
        if n_sp:
            parts_e.append(self.encode_subj(batch.sp_subj).squeeze(1))
            parts_r.append(self.encode_rel(batch.sp_rel).squeeze(1))
        EV, RV = torch.cat(parts_e).contiguous(), torch.cat(parts_r).contiguous()
        ar = lambda a, b: torch.arange(a, b, dtype=torch.int32, device=dev)        # noqa: E731
        vb = H.PrefixBatch(po_rel=ar(0, n_po) if n_po else None, po_obj=ar(n_c, n_c + n_po) if n_po else None,
                           sp_subj=ar(n_c + n_po, n_c + n_po + n_sp) if n_sp else None, sp_rel=ar(n_po, n_po + n_sp) if n_sp else None,
                           pos_row=batch.pos_row, pos_col=batch.pos_col, cand_first=0, n_cand=n_c)
        return eng.forward_backward(EV, RV, self.scorer_name, vb, None, None, loss=loss, label_smoothing=label_smoothing,
                                    normalizer=1.0, scores=scores, loss_only=True)

    def train_step(self, loss="bce", lr=0.1, weight_decay=1e-10, eps=1e-8, label_smoothing=0.0):
        """The training driver for this model: shares the module's parameters (updated in place)."""
        self.flush_steps()                 # an earlier driver (another epoch's learning rate, ...) may still owe decay-only steps
        slots = []
        for emb, tok, bn in ((self.entity_embedding, self.entity_token_ids, self.entity_batchnorm),
                             (self.relation_embedding, self.relation_token_ids, self.relation_batchnorm)):
            s = TokenSlot(emb.weight.data, tok, self.pool, bn is not None, None if bn is None else bn.weight.data,
                          None if bn is None else bn.bias.data)
            if bn is not None:
                s.running_mean, s.running_var = bn.running_mean, bn.running_var
            slots.append(s)
        st = TokenPooledTrainStep(slots[0], slots[1], self.scorer_name, loss=loss, lr=lr, weight_decay=weight_decay, eps=eps,
                                  label_smoothing=label_smoothing, dropout=self.entity_dropout, seed=self.dropout_seed)
        if self.entity_batchnorm is not None:
            st.module_batchnorms = ((slots[0], self.entity_batchnorm), (slots[1], self.relation_batchnorm))
        import weakref
        self._steps = [r for r in self._steps if r() is not None] + [weakref.ref(st)]
        return st


class UnigramPoolingComplexRelationModel(ComplexRelationScorer, UnigramPoolingRelationEmbedder):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)


class UnigramPoolingDistmultRelationModel(DistmultRelationScorer, UnigramPoolingRelationEmbedder):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)


# registered like the reference's (model.py:1052-1066): getattr(Models, args["model"])
Models.UnigramPoolingComplexRelationModel = UnigramPoolingComplexRelationModel
Models.UnigramPoolingDistmultRelationModel = UnigramPoolingDistmultRelationModel
