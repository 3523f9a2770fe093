"""One training step of the hot path, as `Trainer.compute_one_batch(training=True)` drives it in the
reference (openkge/trainer.py:181-257): forward + loss + backward + dense Adagrad on both tables.

All arithmetic is in libokge_hip.so; this class only owns the buffers (tables, dense gradients, Adagrad
accumulators) and sequences the C-ABI calls on the current stream.  No host synchronisation per step.
"""
from __future__ import annotations

import ctypes

import torch

from . import _native as N
from . import hotpath as H


class FusedTrainStep:
    def __init__(self, E: torch.Tensor, R: torch.Tensor, scorer: str, loss: str = "bce", lr: float = 0.3,
                 weight_decay: float = 1e-10, eps: float = 1e-8, label_smoothing: float = 0.0,
                 input_dropout: float = 0.0, relation_input_dropout: float = 0.0, seed: int = 0, engine=None,
                 grad_clip: float = 0.0, accumulate: int = 1):
        """Defaults follow config/fb15k237/fb15k237-complex-kge.yaml and the optimizer OptimRegime actually
        builds (utils/optim.py:29,139-160): Adagrad(lr, weight_decay=1e-10, eps=1e-8 leaked from Adam)."""
        self.E, self.R = E, R
        self.scorer, self.loss = scorer, loss
        self.lr, self.weight_decay, self.eps = lr, weight_decay, eps
        self.label_smoothing = label_smoothing
        self.input_dropout, self.relation_input_dropout = input_dropout, relation_input_dropout
        self.seed = seed
        self.engine = engine or H.HotPath(E.device)
        self.dE, self.dR = torch.zeros_like(E), torch.zeros_like(R)       # dense .grad (model_config.sparse False)
        self.sumE, self.sumR = torch.zeros_like(E), torch.zeros_like(R)   # Adagrad state 'sum' (init 0)
        self.steps = 0
        self.step_dev = None              # device step counter, attached by GraphedTrainStep
        self._grads_zero = True           # fresh buffers; kept true by the zero_grad fused into Adagrad
        self._dE_stale = False            # dE holds last step's values (they get overwritten, not accumulated)
        self.loss_out = torch.zeros(1, dtype=torch.float64, device=E.device)
        # trainer.py:229-244: gradients of `accumulate` consecutive batches are summed before one optimizer step
        # (batch_size_for_backward = accumulate x batch_size), clipped to `grad_clip` (global 2-norm) if > 0
        self.grad_clip, self.accumulate = float(grad_clip or 0.0), max(1, int(accumulate))
        self._accumulated = 0
        # OKGE_FUSED_UPDATE=1: okge_train_step (the update inside the step's launches; needs no clipping / accumulation between
        # backward and update).  Built and measured in round 4 (profiles/round4_ablation.md): tables
        # bit-identical, S-FB 0.1302-0.1310 vs 0.1308-0.1313 ms/step -- under the 3 % bar, so the two-call sequence stays the default
        import os
        d = E.shape[1]
        self.fuse_update = (os.environ.get("OKGE_FUSED_UPDATE", "0") == "1" and self.grad_clip == 0 and self.accumulate == 1
                            and d % (4 if scorer == "distmult" else 8) == 0)
        self._fuse_now, self._fused_done, self._opt = False, False, None

    def state_tensors(self):
        """every tensor a step mutates (GraphedTrainStep snapshots them around its warm-up)"""
        return [self.E, self.R, self.dE, self.dR, self.sumE, self.sumR]

    def _set_dropout(self, batch: H.PrefixBatch, training=True):
        pe = self.input_dropout if training else 0.0
        pr = self.relation_input_dropout if training else 0.0
        s, t, sd = self.seed, self.steps, self.step_dev
        batch.drop_cand = H.DropoutSpec(pe, s, H.STREAM_CAND, t, step_dev=sd)
        batch.drop_po_ent = H.DropoutSpec(pe, s, H.STREAM_PO_ENT, t, step_dev=sd)
        batch.drop_sp_ent = H.DropoutSpec(pe, s, H.STREAM_SP_ENT, t, step_dev=sd)
        batch.drop_po_rel = H.DropoutSpec(pr, s, H.STREAM_PO_REL, t, step_dev=sd)
        batch.drop_sp_rel = H.DropoutSpec(pr, s, H.STREAM_SP_REL, t, step_dev=sd)

    def _covers_all_rows(self, batch: H.PrefixBatch):
        """1-vs-all: the candidate range is every row from cand_first on, so the step overwrites all of dE it uses"""
        return batch.cand_ids is None and batch.cand_first + batch.n_cand == self.E.shape[0]

    # -- the per-step call with persistent descriptors ----------------------------------------------------------------
    # A 0.14 ms step leaves the host ~100 us for everything; building five dropout and four argument structs per step
    # in Python (and doing it next to a producer thread that shares the interpreter lock) is a measurable part of that.
    def _descriptors(self):
        if getattr(self, "_desc", None) is None:
            t = self.engine._tables(self.E, self.R, self.scorer)
            pb, c, pos = N.PrefixBatch(), N.Candidates(), N.Positives()
            for d_, p_, stream in ((pb.drop_po_ent, self.input_dropout, H.STREAM_PO_ENT), (pb.drop_sp_ent, self.input_dropout, H.STREAM_SP_ENT),
                                   (pb.drop_po_rel, self.relation_input_dropout, H.STREAM_PO_REL),
                                   (pb.drop_sp_rel, self.relation_input_dropout, H.STREAM_SP_REL), (c.drop, self.input_dropout, H.STREAM_CAND)):
                d_.p, d_.seed, d_.stream = float(p_), int(self.seed) & 0xFFFFFFFFFFFFFFFF, stream
            self._desc = (t, pb, c, pos, (pb.drop_po_ent, pb.drop_sp_ent, pb.drop_po_rel, pb.drop_sp_rel, c.drop))
        return self._desc

    def _fast_forward_backward(self, batch: H.PrefixBatch, normalizer):
        """same call as HotPath.forward_backward for int32 device tensors (what the batch producer emits)"""
        if H.VALIDATE:
            H.validate_ids(batch, self.E.shape[0], self.R.shape[0])
        t, pb, c, pos, drops = self._descriptors()
        sd = self.step_dev.data_ptr() if self.step_dev is not None else None
        for d_ in drops:
            d_.step, d_.step_dev = self.steps & 0xFFFFFFFF, sd
        n_po, n_sp = batch.n_po, batch.n_sp
        pb.po_rel, pb.po_obj = (batch.po_rel.data_ptr(), batch.po_obj.data_ptr()) if n_po else (None, None)
        pb.sp_subj, pb.sp_rel = (batch.sp_subj.data_ptr(), batch.sp_rel.data_ptr()) if n_sp else (None, None)
        pb.n_po, pb.n_sp = n_po, n_sp
        n = batch.n_candidates
        c.ids, c.first_id, c.n = (batch.cand_ids.data_ptr() if batch.cand_ids is not None else None), int(batch.cand_first), n
        pos.row, pos.col, pos.nnz = batch.pos_row.data_ptr(), batch.pos_col.data_ptr(), batch.nnz
        eng = self.engine
        ws = eng.workspace(n_po + n_sp, n, t.d)
        flags = (N.OKGE_TRAIN_GRADS_ZERO if self._grads_zero else 0) | (N.OKGE_TRAIN_UNIQUE_CANDIDATES if batch.cand_unique else 0)
        if self._fuse_now:
            # the whole step incl. the Adagrad update in ONE library call (okge_train_step): the entity sweep rides in the
            # prefix-backward launch, a small launch finishes the prefix rows and the relation table
            if self._opt is None:
                self._prefix_flags = torch.zeros(self.E.shape[0], dtype=torch.int32, device=self.E.device)
                o = self._opt = N.AdagradOpt()
                o.sum_E, o.sum_R, o.prefix_flags = self.sumE.data_ptr(), self.sumR.data_ptr(), self._prefix_flags.data_ptr()
            o = self._opt
            o.lr, o.weight_decay, o.eps = float(self.lr), float(self.weight_decay), float(self.eps)
            o.zero_entity_grad = 0 if self._last_full else 1
            N.check(eng.lib.okge_train_step(
                ctypes.byref(t), ctypes.byref(pb), ctypes.byref(c), ctypes.byref(pos), N.LOSSES[self.loss], float(self.label_smoothing),
                float(normalizer if normalizer is not None else (n_po + n_sp) * n), flags, ctypes.byref(o), self.loss_out.data_ptr(),
                self.dE.data_ptr(), self.dR.data_ptr(), ws.data_ptr(), eng._ws_bytes, eng._stream()), "okge_train_step")
            self._fused_done = True
            return self.loss_out
        N.check(eng.lib.okge_train_forward_backward(
            ctypes.byref(t), ctypes.byref(pb), ctypes.byref(c), ctypes.byref(pos), N.LOSSES[self.loss], float(self.label_smoothing),
            float(normalizer if normalizer is not None else (n_po + n_sp) * n), flags, self.loss_out.data_ptr(), self.dE.data_ptr(),
            self.dR.data_ptr(), None, 0, ws.data_ptr(), eng._ws_bytes, eng._stream()), "okge_train_forward_backward")
        return self.loss_out

    def _plain(self, batch: H.PrefixBatch):
        """int32 contiguous tensors ON THE ENGINE'S DEVICE everywhere and no replayed masks: nothing for the generic path
        to convert (a host tensor's data_ptr handed to a kernel would be a GPU fault)"""
        dev = self.E.device
        for x in (batch.po_rel, batch.po_obj, batch.sp_subj, batch.sp_rel, batch.pos_row, batch.pos_col, batch.cand_ids):
            if x is not None and (x.dtype != torch.int32 or not x.is_contiguous() or x.dim() != 1 or x.device != dev):
                return False
        return batch.pos_row is not None and batch.cand_table is None

    def forward_backward(self, batch: H.PrefixBatch, normalizer=None):
        """trainer.py:206-234.  Returns the summed loss (device double[1], valid after stream sync)."""
        if self._dE_stale and not (self._grads_zero and self._covers_all_rows(batch)):
            self.dE.zero_()               # a sampled candidate list leaves rows untouched: they must read as zero
            self._dE_stale = False
        self._last_full = self._covers_all_rows(batch)
        if isinstance(self.engine, H.HotPath) and self._plain(batch):
            return self._fast_forward_backward(batch, normalizer)
        self._set_dropout(batch)
        return self.engine.forward_backward(self.E, self.R, self.scorer, batch, self.dE, self.dR, loss=self.loss,
                                            label_smoothing=self.label_smoothing, normalizer=normalizer,
                                            loss_out=self.loss_out, grads_zero=self._grads_zero)

    def optimizer_step(self, lazy_zero=False):
        """trainer.py:240-244: optimizer.step() then zero_grad() -- one sweep per table.  lazy_zero (used by step()
        after a 1-vs-all batch): the entity gradient is not cleared here because the next 1-vs-all step overwrites
        every row it uses; it is cleared on demand if a sampled candidate list comes next."""
        self.engine.adagrad2(self.E, self.dE, self.sumE, self.R, self.dR, self.sumR, self.lr, self.weight_decay,
                             self.eps, zero_grad=2 if lazy_zero else 1)
        self._dE_stale = bool(lazy_zero)

    def step(self, batch: H.PrefixBatch, normalizer=None):
        """forward + loss + backward; every `accumulate`-th call also clips (grad_clip > 0) and takes the Adagrad step"""
        self.steps += 1
        self._fuse_now, self._fused_done = self.fuse_update, False
        loss = self.forward_backward(batch, normalizer)
        self._fuse_now = False
        self._grads_zero = False
        if self._fused_done:                 # the update happened inside the call (dR cleared; dE cleared unless 1-vs-all)
            self._dE_stale = bool(self._last_full)
            self._grads_zero = True
            return loss
        self._accumulated += 1
        if self._accumulated < self.accumulate:
            return loss                      # trainer.py:233-234, 246-248: no optimizer step yet, gradients keep adding up
        self._accumulated = 0
        if self.grad_clip > 0:
            self.engine.clip_grad_norm_(self.dE, self.dR, self.grad_clip)
        self.optimizer_step(lazy_zero=self._last_full and self.accumulate == 1)
        self._grads_zero = True
        return loss


class GraphedTrainStep:
    """Replays one training step as a HIP graph (torch.cuda.CUDAGraph records the library's launches on the capture
    stream): a step is ~5 kernels of 5-80 us for the lookup models and ~50 for the token-pooled ones, so the Python /
    ctypes launch path, not the GPU, bounds small steps.  Shapes are fixed at capture: n_po, n_sp, the candidate
    count and a CAPACITY for the positives -- shorter lists are padded with (row -1, col INT32_MAX), which sort
    behind every candidate tile and are skipped by the kernels.  The dropout step counter lives on the device and is
    incremented inside the graph, so every replay draws new masks (okge_dropout.step_dev).

    `inner` is a FusedTrainStep or a TokenPooledTrainStep (anything with .step(batch), .device-resident state and a
    `step_dev` attribute)."""

    PAD_COL = 2 ** 31 - 1

    def __init__(self, inner, example: H.PrefixBatch, pos_capacity, normalizer=None, counter=None):
        self.inner = inner
        dev = example.pos_row.device
        self.device = dev
        self.pos_capacity = int(pos_capacity)
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)      # noqa: E731
        self.static = H.PrefixBatch(
            po_rel=i32(example.n_po) if example.n_po else None, po_obj=i32(example.n_po) if example.n_po else None,
            sp_subj=i32(example.n_sp) if example.n_sp else None, sp_rel=i32(example.n_sp) if example.n_sp else None,
            pos_row=i32(self.pos_capacity), pos_col=i32(self.pos_capacity),
            cand_ids=None if example.cand_ids is None else i32(example.cand_ids.numel()),
            cand_first=example.cand_first, n_cand=example.n_cand, cand_unique=example.cand_unique)
        self.counter = torch.zeros(1, dtype=torch.int32, device=dev) if counter is None else counter
        inner.step_dev = self.counter
        if getattr(inner, "overlap_sweep", False):
            inner.overlap_sweep = False         # (token-pooled step: the side-stream sweep is a launch-time arrangement, not captured)
        self.normalizer = normalizer
        self._load(example)
        state = inner.state_tensors()
        saved, steps0 = [t.clone() for t in state], inner.steps
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                       # warm-up outside capture: workspaces get allocated
            for _ in range(2):
                inner.step(self.static, normalizer)
        torch.cuda.current_stream(dev).wait_stream(side)
        for t, s0 in zip(state, saved):                     # the warm-up steps must not count as training
            t.copy_(s0)
        inner.steps = steps0
        if counter is None:
            self.counter.fill_(steps0)
        del saved
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: a sharded step carries RCCL collectives, and the process group's watchdog thread polls the events of
        # earlier collectives while this thread captures -- under the default (global) mode that query aborts the process
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.counter += 1
            self.loss = inner.step(self.static, normalizer)

    def _load(self, b: H.PrefixBatch):
        s = self.static
        if b.n_po != s.n_po or b.n_sp != s.n_sp or b.n_candidates != s.n_candidates or b.nnz > self.pos_capacity:
            raise ValueError("batch shape differs from the captured one (n_po, n_sp, candidates fixed; positives <= capacity)")
        for name in ("po_rel", "po_obj", "sp_subj", "sp_rel", "cand_ids"):
            dst, src = getattr(s, name), getattr(b, name)
            if dst is not None:
                dst.copy_(src.reshape(-1), non_blocking=True)
        n = b.nnz
        s.pos_row[:n].copy_(b.pos_row, non_blocking=True)
        s.pos_col[:n].copy_(b.pos_col, non_blocking=True)
        if n < self.pos_capacity:
            s.pos_row[n:].fill_(-1)
            s.pos_col[n:].fill_(self.PAD_COL)

    def step(self, batch: H.PrefixBatch):
        """Copies the batch into the captured buffers and replays; returns the device loss (double[1])."""
        self._load(batch)
        return self.replay()

    def replay(self):
        """Replays on whatever the captured buffers hold (a producer may fill `self.static` directly)."""
        self.graph.replay()
        mark = getattr(self.inner, "mark_pending", None)      # (token-pooled step: deferred decay steps are owed again)
        if mark is not None:
            mark()
        return self.loss
