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
    Slot sizes up to 512 (round 4: above 256 on the register-tile kernel's counting mode), eval mode; PipelinedEvaluator
    (materialised scores) covers dropout and candidate tables."""

    def __init__(self, E, R, scorer, engine=None, run_len=48, two_streams=True, n_streams=None, collect_ranks=False):
        self.E, self.R, self.scorer = E, R, scorer
        # collect_ranks: keep every answer group's rank (self.ranks after run(): int64, batches in order, groups in the
        # batch's row / group order) instead of treating the ranks as scratch -- for parity tests and error analysis
        self.collect_ranks, self.ranks, self._kept = bool(collect_ranks), None, []
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
        ranks = self._ranks
        if self.collect_ranks:                               # one region per batch, kept: a fresh block per issued run
            counts = [int(self._arr[i].n_groups) for i in range(n)]
            ranks = torch.empty(max(sum(counts), 1), dtype=torch.int64, device=self.device)
            off = 0
            for i in range(n):
                self._arr[i].rank_offset = off
                off += counts[i]
            self._kept.append(ranks[:off])
        else:
            for i in range(n):                               # the ranks themselves are scratch here: the regions rotate
                self._arr[i].rank_offset = (i % slots) * n_groups
        handles = (ctypes.c_void_p * self.n_streams)(stream_h, *[ctypes.c_void_p(x.cuda_stream) for x in self.sides])
        # (the library orders the current stream behind the other streams' share at the end of every call; letting the
        #  chains run on across calls instead measured no faster)
        N.check(self.engine.lib.okge_evaluate_fused_batches(ctypes.byref(self._t), self._arr, n, ranks.data_ptr(),
                                                            acc.data_ptr(), self._ws.data_ptr(), self._ws.numel(), handles,
                                                            self.n_streams), "okge_evaluate_fused_batches")

    def run(self, batches):
        """batches: iterable of dataset.CollatedBatch built with is_training_data=False -> (MetricResult, #groups)"""
        acc = torch.zeros(7, dtype=torch.float64, device=self.device)
        self._kept = []
        main = torch.cuda.current_stream(self.device)
        stream_h = ctypes.c_void_p(main.cuda_stream)
        n, need, n_groups, run = 0, 0, 0, min(2 * self.n_streams, self.run_len)   # two batches per chain, then x4:
        keep = []                       # a run's tensors (and id conversions) stay referenced until it has been issued:
        for cb in batches:              # after that the current stream's order protects them (see _issue)
            nd, ng, ka = self._fill(n, cb)
            keep.append((ka, cb))
            need, n_groups, n = max(need, nd), max(n_groups, ng), n + 1
            if n == run:
                self._issue(n, need, n_groups, acc, stream_h)
                keep.clear()
                n, need, n_groups, run = 0, 0, 0, min(4 * run, self.run_len)          # the host fills ~4 batches per batch time
        if n:
            self._issue(n, need, n_groups, acc, stream_h)
        out = _meters(acc.cpu().tolist())          # (synchronises: every kernel that read a batch has finished)
        keep.clear()
        if self.collect_ranks:
            self.ranks = torch.cat(self._kept) if self._kept else torch.zeros(0, dtype=torch.int64, device=self.device)
            self._kept = []
        return out


class PipelinedEvaluator:
    """The evaluation loop on the MATERIALISING path (okge_evaluate_batch: scores into a (B, N) block, then filtered ranks
    and meters) -- any slot size, dropout, candidate tables; FusedEvaluator is the path without the score block in
    eval mode.  Batch i runs on stream i % n_streams with that stream's own score block, rank buffer and query workspace:
    independent chains [scores] [ranks] [meters], no cross-stream wait per batch (each costs ~10 us of queue latency on
    this hardware; the first version ordered a scoring and a ranking stream with two events per batch), so the ranking of
    one batch runs beside the scoring of the next ones.  Meters accumulate on the device; one host read at the end."""

    def __init__(self, E, R, scorer, engine=None, n_streams=3, collect_ranks=False):
        self.E, self.R, self.scorer = E, R, scorer
        self.collect_ranks, self.ranks = bool(collect_ranks), None          # as FusedEvaluator
        self.device = E.device
        self.engine = engine or H.HotPath(self.device)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(n_streams)))]
        k = len(self.streams)
        self._bufs, self._ranks, self._ws = [None] * k, [None] * k, [None] * k
        # the descriptors of a call are built once and only their per-batch pointers change
        self._t = self.engine._tables(E, R, scorer)
        self._pb, self._c = N.PrefixBatch(), N.Candidates()

    def _buffer(self, slot, B, n):
        """score block of a chain: grown (never shrunk) to the largest B x ld seen.  Allocated and freed on the caller's
        stream; every chain joins that stream at the start and at the end of run(), and inside a run a chain's buffer
        changes only between two of ITS OWN batches, i.e. in its own stream's order, after a synchronize."""
        ld = (n + 3) // 4 * 4
        buf = self._bufs[slot]
        if buf is None or buf.shape[0] < B or buf.shape[1] < ld:
            self.streams[slot].synchronize()
            rows = max(B, 0 if buf is None else buf.shape[0])
            cols = max(ld, 0 if buf is None else buf.shape[1])
            buf = self._bufs[slot] = torch.empty((rows, cols), dtype=torch.float32, device=self.device)
            buf.record_stream(self.streams[slot])
        return buf[:B, :n]

    def run(self, batches):
        """batches: iterable of dataset.CollatedBatch built with is_training_data=False.
        -> (MetricResult with mrr / mr / h1 / h3 / h10 / h50 filled like compute_metrics, number of answer groups)"""
        eng = self.engine
        # one meter row per chain: the chains are unordered among each other, so a shared row would be a race between
        # their meter launches (and the order of the double sums would change from run to run); summed after the join
        acc = torch.zeros((len(self.streams), 7), dtype=torch.float64, device=self.device)
        main = torch.cuda.current_stream(self.device)
        for st in self.streams:
            st.wait_stream(main)                             # tables / batches were produced on the current stream
        keep, kept = [], []                                  # every batch stays referenced until the chains have joined
        for i, cb in enumerate(batches):
            slot = i % len(self.streams)
            st = self.streams[slot]
            x = self._buffer(slot, cb.batch.B, cb.n_cand)
            n_groups = int(cb.grp_ptr.numel()) - 1
            if self.collect_ranks:                           # a block of its own per batch, kept until the end
                rk = torch.empty(max(n_groups, 1), dtype=torch.int64, device=self.device)
                rk.record_stream(st)
                kept.append(rk[:n_groups])
            else:
                if self._ranks[slot] is None or self._ranks[slot].numel() < n_groups:
                    st.synchronize()
                    self._ranks[slot] = torch.empty(max(n_groups, 1024), dtype=torch.int64, device=self.device)
                    self._ranks[slot].record_stream(st)
                rk = self._ranks[slot]
            b, pb, c = cb.batch, self._pb, self._c
            pb.po_rel, pb.po_obj = (b.po_rel.data_ptr(), b.po_obj.data_ptr()) if b.po_rel is not None else (None, None)
            pb.sp_subj, pb.sp_rel = (b.sp_subj.data_ptr(), b.sp_rel.data_ptr()) if b.sp_subj is not None else (None, None)
            pb.n_po, pb.n_sp = b.n_po, b.n_sp
            c.ids = b.cand_ids.data_ptr() if b.cand_ids is not None else None
            c.first_id, c.n = b.cand_first, cb.n_cand
            need = int(eng.lib.okge_score_workspace_bytes(b.B, self._t.d))
            if self._ws[slot] is None or self._ws[slot].numel() < need:
                st.synchronize()
                self._ws[slot] = torch.empty(need, dtype=torch.uint8, device=self.device)
                self._ws[slot].record_stream(st)
            sh = ctypes.c_void_p(st.cuda_stream)
            # one library call: scores, ranks and meters of the batch, all in the chain's stream order
            N.check(eng.lib.okge_evaluate_batch(ctypes.byref(self._t), ctypes.byref(pb), ctypes.byref(c),
                                                cb.filt_ptr.data_ptr(), cb.filt_col.data_ptr() if cb.filt_col.numel() else None,
                                                cb.row_ptr.data_ptr(), cb.grp_ptr.data_ptr(), cb.ids.data_ptr(), n_groups,
                                                x.data_ptr(), x.stride(0), rk.data_ptr(), acc[slot].data_ptr(),
                                                self._ws[slot].data_ptr(), self._ws[slot].numel(), sh, sh), "okge_evaluate_batch")
            keep.append(cb)
        for st in self.streams:
            main.wait_stream(st)
        a = acc.sum(0).cpu().tolist()
        if self.collect_ranks:
            self.ranks = torch.cat(kept) if kept else torch.zeros(0, dtype=torch.int64, device=self.device)
        del keep
        return _meters(a)
