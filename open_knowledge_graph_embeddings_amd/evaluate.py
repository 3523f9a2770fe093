"""Evaluation loop: `Trainer.evaluate` -> `compute_one_batch(training=False)` -> `compute_metrics`
(openkge/trainer.py:363-369, :258-272; openkge/dataset.py:423-453) over the HIP kernels, with the two kernels of a batch
on two streams: scoring a batch is compute-bound, ranking it is a chain of memory round trips, so ranking batch i
overlaps scoring batch i+1 (two score buffers).  Meters accumulate on the device; one host read at the end."""
from __future__ import annotations

import ctypes

import torch

from . import _native as N
from . import hotpath as H
from .metrics import MetricResult


def _meters(a):
    n = int(a[0])
    out = MetricResult()
    if n:
        out["mrr"].update(a[1] / n, n)
        out["mr"].update(a[2] / n, n)
        for k, v in (("h1", a[3]), ("h3", a[4]), ("h10", a[5]), ("h50", a[6])):
            out[k].update(v / n, n)
    return out, n


class FusedEvaluator:
    """The evaluation loop on okge_evaluate_fused_batches: no (B, N) score block, no host read until the end.  The library
    issues a run of batches per call (`run_len`; the first runs are short so that the device starts early): batch i on
    stream i % n_streams (the current stream + n_streams - 1 of its own; default 3), each stream an independent chain of
    two launches per batch -- the tile sweep with in-register counting, then one small launch holding the ranks + meters
    of that batch and the point scores of the chain's next batch -- so one chain's small latency-bound launches, and the
    CUs a sweep's 228 tiles leave empty, are filled with the other chains' work without any per-batch cross-stream wait
    (those cost ~10 us each; round 2's first design paid one per batch).  Measured per batch at the FB15k-237 shape:
    1 chain 0.056 ms, 2 chains 0.046, 3 chains 0.043, 4 chains 0.043-0.048.
    Slot sizes up to 256, eval mode; PipelinedEvaluator (materialised scores) covers the rest."""

    def __init__(self, E, R, scorer, engine=None, run_len=48, two_streams=True, n_streams=None):
        self.E, self.R, self.scorer = E, R, scorer
        self.device = E.device
        self.engine = engine or H.HotPath(self.device)
        self.run_len = int(run_len)
        self.n_streams = int(n_streams) if n_streams is not None else (3 if two_streams else 1)
        self.sides = [torch.cuda.Stream(device=self.device) for _ in range(self.n_streams - 1)]
        self.side = self.sides[0] if self.sides else None
        self._ws = None
        self._ranks = None
        self._t = self.engine._tables(E, R, scorer)
        self._arr = (N.EvalBatch * self.run_len)()

    def _fill(self, slot, cb):
        eng, b = self.engine, cb.batch
        pb, c, keep = eng._batch(b)
        n_groups, n_filter = int(cb.grp_ptr.numel()) - 1, int(cb.filt_col.numel())
        x = self._arr[slot]
        x.batch, x.cand = pb, c
        x.filt_ptr, x.filt_col, x.n_filter = cb.filt_ptr.data_ptr(), (cb.filt_col.data_ptr() if n_filter else None), n_filter
        x.row_ptr, x.grp_ptr, x.ids, x.n_groups = cb.row_ptr.data_ptr(), cb.grp_ptr.data_ptr(), cb.ids.data_ptr(), n_groups
        need = int(eng.lib.okge_eval_workspace_bytes(b.B, c.n, self._t.d, n_groups, n_filter))
        return need, n_groups, keep

    def _issue(self, n, need, n_groups, acc, stream_h):
        slots = 2 * self.n_streams
        slot = (need + 255) // 256 * 256
        if self._ws is None or self._ws.numel() < slots * slot or self._ranks.numel() < slots * n_groups:
            self._ws = torch.empty(slots * max(slot, 0 if self._ws is None else self._ws.numel() // slots), dtype=torch.uint8,
                                   device=self.device)
            self._ranks = torch.empty(max(slots * n_groups, 4096, 0 if self._ranks is None else self._ranks.numel()),
                                      dtype=torch.int64, device=self.device)
        for i in range(n):                                   # the ranks themselves are scratch here: the regions rotate
            self._arr[i].rank_offset = (i % slots) * n_groups
        handles = (ctypes.c_void_p * self.n_streams)(stream_h, *[ctypes.c_void_p(x.cuda_stream) for x in self.sides])
        # (the library orders the current stream behind the other streams' share at the end of every call; letting the
        #  chains run on across calls instead measured no faster)
        N.check(self.engine.lib.okge_evaluate_fused_batches(ctypes.byref(self._t), self._arr, n, self._ranks.data_ptr(),
                                                            acc.data_ptr(), self._ws.data_ptr(), self._ws.numel(), handles,
                                                            self.n_streams), "okge_evaluate_fused_batches")

    def run(self, batches):
        """batches: iterable of dataset.CollatedBatch built with is_training_data=False -> (MetricResult, #groups)"""
        acc = torch.zeros(7, dtype=torch.float64, device=self.device)
        main = torch.cuda.current_stream(self.device)
        stream_h = ctypes.c_void_p(main.cuda_stream)
        n, need, n_groups, run = 0, 0, 0, min(max(2, self.n_streams), self.run_len)
        keep = []                       # a run's tensors (and id conversions) stay referenced until it has been issued:
        for cb in batches:              # after that the current stream's order protects them (see _issue)
            nd, ng, ka = self._fill(n, cb)
            keep.append((ka, cb))
            need, n_groups, n = max(need, nd), max(n_groups, ng), n + 1
            if n == run:
                self._issue(n, need, n_groups, acc, stream_h)
                keep.clear()
                n, need, n_groups, run = 0, 0, 0, min(2 * run, self.run_len)
        if n:
            self._issue(n, need, n_groups, acc, stream_h)
        out = _meters(acc.cpu().tolist())          # (synchronises: every kernel that read a batch has finished)
        keep.clear()
        return out


class PipelinedEvaluator:
    def __init__(self, E, R, scorer, engine=None):
        self.E, self.R, self.scorer = E, R, scorer
        self.device = E.device
        self.engine = engine or H.HotPath(self.device)
        self.rank_stream = torch.cuda.Stream(device=self.device)
        self._bufs = [None, None]
        self._ranks = [None, None]
        # the descriptors of a call are built once and only their per-batch pointers change: at ~60 us of device work
        # per batch the Python cost of filling three ctypes structs is what decides whether the two streams overlap
        self._t = self.engine._tables(E, R, scorer)
        self._pb, self._c = N.PrefixBatch(), N.Candidates()
        self._rs = ctypes.c_void_p(self.rank_stream.cuda_stream)

    def _buffer(self, slot, B, n):
        """score buffer of a slot: grown (never shrunk) to the largest B x ld seen, and only after the rank stream has
        drained -- the ranks kernel of an earlier batch may still be reading the old one -- so a slot keeps ONE
        pointer for the library's per-buffer events; the rank stream is recorded as a user of the allocation"""
        ld = (n + 3) // 4 * 4
        buf = self._bufs[slot]
        if buf is None or buf.shape[0] < B or buf.shape[1] < ld:
            self.rank_stream.synchronize()
            rows = max(B, 0 if buf is None else buf.shape[0])
            cols = max(ld, 0 if buf is None else buf.shape[1])
            buf = self._bufs[slot] = torch.empty((rows, cols), dtype=torch.float32, device=self.device)
            buf.record_stream(self.rank_stream)
        return buf[:B, :n]

    def run(self, batches):
        """batches: iterable of dataset.CollatedBatch built with is_training_data=False.
        -> (MetricResult with mrr / mr / h1 / h3 / h10 / h50 filled like compute_metrics, number of answer groups)"""
        eng = self.engine
        acc = torch.zeros(7, dtype=torch.float64, device=self.device)
        main = torch.cuda.current_stream(self.device)
        keep = None
        for i, cb in enumerate(batches):
            slot = i & 1
            x = self._buffer(slot, cb.batch.B, cb.n_cand)
            n_groups = int(cb.grp_ptr.numel()) - 1
            if self._ranks[slot] is None or self._ranks[slot].numel() < n_groups:
                self.rank_stream.synchronize()
                self._ranks[slot] = torch.empty(max(n_groups, 1024), dtype=torch.int64, device=self.device)
                self._ranks[slot].record_stream(self.rank_stream)
            for t in (cb.filt_ptr, cb.filt_col, cb.row_ptr, cb.grp_ptr, cb.ids):
                t.record_stream(self.rank_stream)                # read there after `cb` may have gone out of scope here
            # one library call: scores on the current stream, ranks + meters on rank_stream, ordered by library events;
            # the buffer of batch i-2 is reused only after its ranks were counted
            b, pb, c = cb.batch, self._pb, self._c
            pb.po_rel, pb.po_obj = (b.po_rel.data_ptr(), b.po_obj.data_ptr()) if b.po_rel is not None else (None, None)
            pb.sp_subj, pb.sp_rel = (b.sp_subj.data_ptr(), b.sp_rel.data_ptr()) if b.sp_subj is not None else (None, None)
            pb.n_po, pb.n_sp = b.n_po, b.n_sp
            c.ids = b.cand_ids.data_ptr() if b.cand_ids is not None else None
            c.first_id, c.n = b.cand_first, cb.n_cand
            ws = eng.workspace(b.B, cb.n_cand, self._t.d, "score")
            N.check(eng.lib.okge_evaluate_batch(ctypes.byref(self._t), ctypes.byref(pb), ctypes.byref(c),
                                                cb.filt_ptr.data_ptr(), cb.filt_col.data_ptr() if cb.filt_col.numel() else None,
                                                cb.row_ptr.data_ptr(), cb.grp_ptr.data_ptr(), cb.ids.data_ptr(), n_groups,
                                                x.data_ptr(), x.stride(0), self._ranks[slot].data_ptr(), acc.data_ptr(),
                                                ws.data_ptr(), eng._ws_bytes, eng._stream(), self._rs), "okge_evaluate_batch")
            keep = cb
        main.wait_stream(self.rank_stream)
        a = acc.cpu().tolist()
        del keep
        return _meters(a)
