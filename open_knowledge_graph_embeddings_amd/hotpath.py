"""Host driver of the HIP hot path: torch tensors are containers (device memory + stream), all arithmetic
happens in libokge_hip.so through the C ABI of include/okge.h.

Reference call sites this module stands in for (paths relative to the reference checkout):
  AddLossModule.forward + backward    openkge/trainer.py:48-113, :217-234   -> HotPath.forward_backward
  *_prefix_score / _score             openkge/model.py:52-74, :198-229, :268-274 -> HotPath.score
  optimizer.step / zero_grad          utils/optim.py:139-160, trainer.py:229-244 -> HotPath.adagrad
  compute_metrics rank rule           openkge/dataset.py:423-446            -> HotPath.filtered_ranks
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Optional

import torch

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)

from . import _native as N


@dataclass
class DropoutSpec:
    """Dropout on gathered rows (model.py:461-462).  `keep` (uint8 [rows, d], device) overrides the
    counter-based Philox mask -- used to replay masks captured from the reference."""
    p: float = 0.0
    seed: int = 0
    stream: int = 0
    step: int = 0
    keep: Optional[torch.Tensor] = None
    step_dev: Optional[torch.Tensor] = None   # int32[1] on the device: the kernels read the step from it (graph replay)

    def c(self) -> N.Dropout:
        d = N.Dropout()
        d.p = float(self.p)
        d.seed = int(self.seed) & 0xFFFFFFFFFFFFFFFF
        d.stream = int(self.stream) & 0xFFFFFFFF
        d.step = int(self.step) & 0xFFFFFFFF
        d.keep = self.keep.data_ptr() if (self.keep is not None and self.p > 0) else None
        d.step_dev = self.step_dev.data_ptr() if self.step_dev is not None else None
        return d


NO_DROP = DropoutSpec()

# OKGE_VALIDATE=1: check every id tensor against the table sizes before a call (one host sync per call).  The kernels
# trust their ids -- they live in device memory -- so a bad id is an out-of-bounds access; use this while integrating.
VALIDATE = os.environ.get("OKGE_VALIDATE") == "1"


def validate_ids(batch, n_ent, n_rel):
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        return                                     # no host reads inside a graph capture
    def check(name, x, hi):
        if x is not None and x.numel():
            lo_, hi_ = int(x.min()), int(x.max())
            if lo_ < 0 or hi_ >= hi:
                raise N.OkgeError(f"{name}: ids span [{lo_}, {hi_}] but the table has {hi} rows")
    check("po_rel", batch.po_rel, n_rel)
    check("sp_rel", batch.sp_rel, n_rel)
    check("po_obj", batch.po_obj, n_ent)
    check("sp_subj", batch.sp_subj, n_ent)
    if batch.cand_table is None:
        check("cand_ids", batch.cand_ids, n_ent)
        if batch.cand_ids is None and (batch.cand_first < 0 or batch.cand_first + batch.n_cand > n_ent):
            raise N.OkgeError("candidate range outside the entity table")
    n_c = batch.n_candidates if batch.cand_table is None else batch.cand_table.shape[0]
    if batch.pos_row is not None and batch.pos_row.numel():
        live = batch.pos_row >= 0                  # (row -1, col INT32_MAX) pads a fixed-capacity list (GraphedTrainStep)
        check("pos_col", batch.pos_col[live], n_c)
        check("pos_row", batch.pos_row[live], batch.B)

# Philox stream ids: one per place the reference draws an independent Bernoulli mask
STREAM_CAND, STREAM_PO_ENT, STREAM_PO_REL, STREAM_SP_ENT, STREAM_SP_REL = 0, 1, 2, 3, 4


@dataclass
class PrefixBatch:
    """One batch in device memory: po rows (rel, obj) first, then sp rows (subj, rel); positives as
    (row, col) coordinates sorted by col (col = position in the candidate list)."""
    po_rel: Optional[torch.Tensor] = None   # int32 [n_po]
    po_obj: Optional[torch.Tensor] = None
    sp_subj: Optional[torch.Tensor] = None  # int32 [n_sp]
    sp_rel: Optional[torch.Tensor] = None
    pos_row: Optional[torch.Tensor] = None  # int32 [nnz]
    pos_col: Optional[torch.Tensor] = None
    cand_ids: Optional[torch.Tensor] = None  # int32 [N] or None for the range cand_first .. cand_first+N-1
    cand_first: int = 2
    n_cand: int = 0
    cand_table: Optional[torch.Tensor] = None  # scoring only: gather candidates from this (rows, d) table, not E
    cand_unique: bool = False               # cand_ids names every entity at most once (the collator's lists do)
    drop_po_ent: DropoutSpec = field(default_factory=DropoutSpec)
    drop_po_rel: DropoutSpec = field(default_factory=DropoutSpec)
    drop_sp_ent: DropoutSpec = field(default_factory=DropoutSpec)
    drop_sp_rel: DropoutSpec = field(default_factory=DropoutSpec)
    drop_cand: DropoutSpec = field(default_factory=DropoutSpec)

    @property
    def n_po(self):
        return 0 if self.po_rel is None else int(self.po_rel.numel())

    @property
    def n_sp(self):
        return 0 if self.sp_subj is None else int(self.sp_subj.numel())

    @property
    def B(self):
        return self.n_po + self.n_sp

    @property
    def nnz(self):
        return 0 if self.pos_row is None else int(self.pos_row.numel())

    @property
    def n_candidates(self):
        return int(self.n_cand if self.cand_ids is None else self.cand_ids.numel())


@dataclass
class Shard:
    """Rows [ent_lo, ent_hi) of the entity table live on this rank; cand_col0 = position of the first local
    candidate in the un-sharded candidate list (include/okge.h, okge_shard)."""
    ent_lo: int
    ent_hi: int
    cand_col0: int = 0

    def c(self) -> N.Shard:
        s = N.Shard()
        s.ent_lo, s.ent_hi, s.cand_col0 = int(self.ent_lo), int(self.ent_hi), int(self.cand_col0)
        return s


def _ptr(t):
    return None if t is None else t.data_ptr()


def _i32(t, dev):
    if t is None:
        return None
    t = t.reshape(-1)
    if t.dtype != torch.int32 or t.device != dev or not t.is_contiguous():
        t = t.to(device=dev, dtype=torch.int32).contiguous()
    return t


class HotPath:
    """Owns the scratch workspace for one device and issues the C-ABI calls on torch's current stream."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.OkgeError("the open-KGE hot path runs on an MI355X (torch device 'cuda'); there is no CPU path")
        self.lib = N.lib()
        self._ws = None
        self._ws_bytes = 0

    # -- plumbing ------------------------------------------------------------------------------------
    def _stream(self):
        # the raw handle of torch's current stream on this device (torch.cuda.current_stream builds a Stream object:
        # ~5 us per call, and every library call asks)
        idx = self.device.index
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device() if idx is None else idx))

    def workspace(self, B, n_cand, d, kind="train"):
        """scratch for one call; kind 'score' (query block only) and 'lse' are much smaller than 'train'"""
        if kind == "score":
            need = int(self.lib.okge_score_workspace_bytes(B, d))
        elif kind == "lse":
            need = int(self.lib.okge_lse_workspace_bytes(B, n_cand, d))
        else:
            need = int(self.lib.okge_train_workspace_bytes(B, n_cand, d))
        if need == 0:
            raise N.OkgeError(f"invalid problem size B={B} N={n_cand} d={d}")
        if need > self._ws_bytes:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        return self._ws

    def _check(self, batch, E, R):
        if VALIDATE:
            validate_ids(batch, E.shape[0], R.shape[0])

    def _tables(self, E, R, scorer):
        for t in (E, R):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device:
                raise N.OkgeError("embedding tables must be contiguous fp32 tensors on the engine's device")
        t = N.Tables()
        t.E, t.R = E.data_ptr(), R.data_ptr()
        t.n_ent, t.n_rel, t.d = E.shape[0], R.shape[0], E.shape[1]
        t.scorer = N.SCORERS[scorer] if isinstance(scorer, str) else int(scorer)
        return t

    def _batch(self, b: PrefixBatch):
        dev = self.device
        keep = [_i32(b.po_rel, dev), _i32(b.po_obj, dev), _i32(b.sp_subj, dev), _i32(b.sp_rel, dev),
                _i32(b.cand_ids, dev)]
        pb = N.PrefixBatch()
        pb.po_rel, pb.po_obj, pb.sp_subj, pb.sp_rel = (_ptr(x) for x in keep[:4])
        pb.n_po, pb.n_sp = b.n_po, b.n_sp
        pb.drop_po_ent, pb.drop_po_rel = b.drop_po_ent.c(), b.drop_po_rel.c()
        pb.drop_sp_ent, pb.drop_sp_rel = b.drop_sp_ent.c(), b.drop_sp_rel.c()
        c = N.Candidates()
        c.ids = _ptr(keep[4])
        c.first_id = int(b.cand_first)
        c.n = int(b.n_cand if keep[4] is None else keep[4].numel())
        c.drop = b.drop_cand.c()
        if b.cand_table is not None:
            ct = b.cand_table
            if ct.dtype != torch.float32 or not ct.is_contiguous() or ct.device != dev:
                raise N.OkgeError("candidate table must be a contiguous fp32 tensor on the engine's device")
            c.table, c.table_rows = ct.data_ptr(), ct.shape[0]
            keep.append(ct)
        return pb, c, keep

    # -- entry points -------------------------------------------------------------------------------
    def score(self, E, R, scorer, batch: PrefixBatch, out=None):
        """(B, N) scores of every prefix against every candidate (eval / *_prefix_score)."""
        self._check(batch, E, R)
        pb, c, keep = self._batch(batch)
        t = self._tables(E, R, scorer)
        B, n = batch.B, c.n
        ws = self.workspace(B, n, t.d, "score")
        if out is None:
            ld = (n + 3) // 4 * 4
            out = torch.empty((B, ld), dtype=torch.float32, device=self.device)[:, :n]
        N.check(self.lib.okge_score_prefixes(ctypes.byref(t), ctypes.byref(pb), ctypes.byref(c), out.data_ptr(),
                                             out.stride(0), ws.data_ptr(), self._ws_bytes, self._stream()),
                "okge_score_prefixes")
        del keep
        return out

    def forward_backward(self, E, R, scorer, batch: PrefixBatch, dE, dR, loss="bce", label_smoothing=0.0,
                         normalizer=None, loss_out=None, scores=None, grads_zero=False, loss_only=False,
                         distinct_prefix_rows=False, clear_grads=False):
        """Fused forward + loss + backward; accumulates into dE / dR (clear_grads: into whatever-they-held buffers that the
        call itself clears, okge.h OKGE_TRAIN_CLEAR_GRADS); returns the summed loss as a device double[1] tensor (no host sync)."""
        self._check(batch, E, R)
        pb, c, keep = self._batch(batch)
        t = self._tables(E, R, scorer)
        B, n = batch.B, c.n
        ws = self.workspace(B, n, t.d)
        pos = N.Positives()
        prow, pcol = _i32(batch.pos_row, self.device), _i32(batch.pos_col, self.device)
        pos.row, pos.col, pos.nnz = _ptr(prow), _ptr(pcol), batch.nnz
        if normalizer is None:
            normalizer = float(B) * float(n)
        if loss_out is None:
            loss_out = torch.empty(1, dtype=torch.float64, device=self.device)
        N.check(self.lib.okge_train_forward_backward(
            ctypes.byref(t), ctypes.byref(pb), ctypes.byref(c), ctypes.byref(pos),
            N.LOSSES[loss] if isinstance(loss, str) else int(loss), float(label_smoothing), float(normalizer),
            (N.OKGE_TRAIN_GRADS_ZERO if grads_zero else 0) | (N.OKGE_TRAIN_LOSS_ONLY if loss_only else 0) |
            (N.OKGE_TRAIN_UNIQUE_CANDIDATES if batch.cand_unique else 0) |
            (N.OKGE_TRAIN_DISTINCT_PREFIX_ROWS if distinct_prefix_rows else 0) |
            (N.OKGE_TRAIN_CLEAR_GRADS if clear_grads else 0),
            loss_out.data_ptr(), _ptr(dE), _ptr(dR),
            None if scores is None else scores.data_ptr(), 0 if scores is None else scores.stride(0),
            ws.data_ptr(), self._ws_bytes, self._stream()), "okge_train_forward_backward")
        del keep, prow, pcol
        return loss_out

    # -- entity-sharded phases (include/okge.h: okge_encode_queries / okge_train_tiles / okge_prefix_backward) ----
    def query_shape(self, B, d):
        return int(self.lib.okge_query_rows(B)), int(self.lib.okge_query_ld(d))

    def encode_queries(self, E_local, R, scorer, batch: PrefixBatch, shard: Shard, out=None):
        """-> buffer [2][rows][ld]: out[0] = folded queries, out[1] = masked prefix entity rows; rows of prefixes
        whose entity another rank owns are zero (the caller all-reduces the buffer)."""
        pb, c, keep = self._batch(batch)
        t = self._tables(E_local, R, scorer)
        rows, ld = self.query_shape(batch.B, t.d)
        if out is None:
            out = torch.empty((2, rows, ld), dtype=torch.float32, device=self.device)
        sh = shard.c()
        N.check(self.lib.okge_encode_queries(ctypes.byref(t), ctypes.byref(sh), ctypes.byref(pb), out[0].data_ptr(), ld,
                                             out[1].data_ptr(), self._stream()), "okge_encode_queries")
        del keep
        return out

    def encode_entity_rows(self, E_local, R, scorer, batch: PrefixBatch, shard: Shard, out=None):
        """-> [rows][ld] masked prefix entity rows of the prefixes whose entity lives here, zeros elsewhere (the
        caller all-reduces the buffer, then every rank calls fold_queries)."""
        pb, c, keep = self._batch(batch)
        t = self._tables(E_local, R, scorer)
        rows, ld = self.query_shape(batch.B, t.d)
        if out is None:
            out = torch.empty((rows, ld), dtype=torch.float32, device=self.device)
        sh = shard.c()
        N.check(self.lib.okge_encode_queries(ctypes.byref(t), ctypes.byref(sh), ctypes.byref(pb), None, ld,
                                             out.data_ptr(), self._stream()), "okge_encode_queries")
        del keep
        return out

    def fold_queries(self, E_local, R, scorer, batch: PrefixBatch, ent_rows, out=None):
        """query block from exchanged masked entity rows and the replicated relation table"""
        pb, c, keep = self._batch(batch)
        t = self._tables(E_local, R, scorer)
        if out is None:
            out = torch.empty_like(ent_rows)
        N.check(self.lib.okge_fold_queries(ctypes.byref(t), ctypes.byref(pb), ent_rows.data_ptr(), ent_rows.stride(0),
                                           out.data_ptr(), self._stream()), "okge_fold_queries")
        del keep
        return out

    def train_tiles(self, E_local, R, scorer, Q, batch: PrefixBatch, shard: Shard, dE, dQ, n_cand_global, loss="bce",
                    label_smoothing=0.0, normalizer=None, loss_out=None, grads_zero=False, row_lse=None):
        """Local candidates only: batch.cand_first / n_cand are LOCAL row indices, batch.pos_col GLOBAL columns.
        KL loss: `row_lse` = log-sum-exp of every row's scores over ALL shards' candidates."""
        pb, c, keep = self._batch(batch)
        t = self._tables(E_local, R, scorer)
        B, n = batch.B, c.n
        ws = self.workspace(B, n, t.d)
        pos = N.Positives()
        prow, pcol = _i32(batch.pos_row, self.device), _i32(batch.pos_col, self.device)
        pos.row, pos.col, pos.nnz = _ptr(prow), _ptr(pcol), batch.nnz
        if normalizer is None:
            normalizer = float(B) * float(n_cand_global)
        if loss_out is None:
            loss_out = torch.empty(1, dtype=torch.float64, device=self.device)
        sh = shard.c()
        N.check(self.lib.okge_train_tiles(
            ctypes.byref(t), ctypes.byref(sh), Q.data_ptr(), Q.stride(0), B, ctypes.byref(c), ctypes.byref(pos),
            N.LOSSES[loss] if isinstance(loss, str) else int(loss), float(label_smoothing), float(normalizer),
            int(n_cand_global), N.OKGE_TRAIN_GRADS_ZERO if grads_zero else 0, _ptr(row_lse), loss_out.data_ptr(),
            dE.data_ptr(), dQ.data_ptr(), ws.data_ptr(), self._ws_bytes, self._stream()), "okge_train_tiles")
        del keep, prow, pcol
        return loss_out

    def score_queries(self, E_local, R, scorer, Q, B, batch: PrefixBatch, shard: Shard, out=None):
        """(B, n_local) scores of precomputed query rows against the local candidates."""
        pb, c, keep = self._batch(batch)
        t = self._tables(E_local, R, scorer)
        if out is None:
            out = torch.empty((B, (c.n + 3) // 4 * 4), dtype=torch.float32, device=self.device)[:, :c.n]
        sh = shard.c()
        N.check(self.lib.okge_score_queries(ctypes.byref(t), ctypes.byref(sh), Q.data_ptr(), Q.stride(0), B,
                                            ctypes.byref(c), out.data_ptr(), out.stride(0), self._stream()),
                "okge_score_queries")
        del keep
        return out

    def row_logsumexp(self, E_local, R, scorer, Q, B, batch: PrefixBatch, shard: Shard):
        """(B,) log-sum-exp of each query row's scores over the local candidates (scores are not materialised)."""
        pb, c, keep = self._batch(batch)
        t = self._tables(E_local, R, scorer)
        ws = self.workspace(B, c.n, t.d, "lse")
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        sh = shard.c()
        N.check(self.lib.okge_row_logsumexp(ctypes.byref(t), ctypes.byref(sh), Q.data_ptr(), Q.stride(0), B,
                                            ctypes.byref(c), out.data_ptr(), ws.data_ptr(), self._ws_bytes,
                                            self._stream()), "okge_row_logsumexp")
        del keep
        return out

    def prefix_backward(self, E_local, R, scorer, batch: PrefixBatch, shard: Shard, dQ, ent_rows, dE, dR, rel_segments=None):
        """rel_segments: sharded.RowSegments (make_row_segments) -- relation and / or entity gradients by sorted segments
        instead of float atomics"""
        pb, c, keep = self._batch(batch)
        t = self._tables(E_local, R, scorer)
        sh = shard.c()
        if rel_segments is not None:
            sg = rel_segments
            if getattr(self, "_grad_rows", None) is None or self._grad_rows.shape[1:] != dQ.shape:
                self._grad_rows = torch.empty((2,) + tuple(dQ.shape), dtype=dQ.dtype, device=dQ.device)
            r_o, r_p = sg.rel if sg.rel is not None else (None, None)
            e_o, e_p = sg.ent if sg.ent is not None else (None, None)
            N.check(self.lib.okge_prefix_backward_segmented(ctypes.byref(t), ctypes.byref(sh), ctypes.byref(pb), dQ.data_ptr(),
                                                            dQ.stride(0), _ptr(ent_rows), _ptr(r_o), _ptr(r_p),
                                                            0 if r_p is None else int(r_p.numel()) - 1, _ptr(e_o), _ptr(e_p),
                                                            0 if e_p is None else int(e_p.numel()) - 1, self._grad_rows.data_ptr(),
                                                            dE.data_ptr(), dR.data_ptr(), self._stream()),
                    "okge_prefix_backward_segmented")
            del keep
            return
        N.check(self.lib.okge_prefix_backward(ctypes.byref(t), ctypes.byref(sh), ctypes.byref(pb), dQ.data_ptr(),
                                              dQ.stride(0), _ptr(ent_rows), dE.data_ptr(), dR.data_ptr(),
                                              self._stream()), "okge_prefix_backward")
        del keep

    def score_triples(self, scorer, subj, rel, obj):
        """(b, 1) scores of b encoded triples (rows (b, d), last dim contiguous): triple_score (model.py:178-179)."""
        rows = [x.reshape(-1, x.shape[-1]) for x in (subj, rel, obj)]
        b, d = rows[0].shape
        for x in rows:
            if x.dtype != torch.float32 or x.stride(1) != 1 or x.device != self.device or x.shape != (b, d):
                raise N.OkgeError("triple rows must be fp32 (b, d) on the engine's device with a contiguous last dim")
        out = torch.empty((b, 1), dtype=torch.float32, device=self.device)
        N.check(self.lib.okge_score_triples(N.SCORERS[scorer] if isinstance(scorer, str) else int(scorer),
                                            rows[0].data_ptr(), rows[0].stride(0), rows[1].data_ptr(), rows[1].stride(0),
                                            rows[2].data_ptr(), rows[2].stride(0), b, d, out.data_ptr(), self._stream()),
                "okge_score_triples")
        return out

    def encode_rows(self, table, ids=None, first_id=0, n=None, drop: DropoutSpec = NO_DROP, out=None):
        """dropout(table[ids]) -> (n, d): LookupBaseRelationEmbedder._encode (model.py:455-480)."""
        ids = _i32(ids, self.device)
        n = int(ids.numel()) if ids is not None else int(n)
        d = table.shape[1]
        if out is None:
            out = torch.empty((n, d), dtype=torch.float32, device=self.device)
        dc = drop.c()
        N.check(self.lib.okge_encode_rows(table.data_ptr(), table.shape[0], d, _ptr(ids), int(first_id), n,
                                          ctypes.byref(dc), out.data_ptr(), out.stride(0), self._stream()),
                "okge_encode_rows")
        return out

    def prefix_score_backward(self, scorer, sp, g, ent, rel, cand, need_ent=True, need_rel=True, need_cand=True):
        """(d_ent, d_rel, d_cand) of scores = fold(ent, rel) . cand^T from the dense (b, n) gradient g (okge_prefix_score_backward)"""
        b, n = g.shape
        d = cand.shape[1]
        need = int(self.lib.okge_prefix_score_backward_workspace_bytes(b, n, d))
        if getattr(self, "_sb_ws", None) is None or self._sb_ws.numel() < need:
            self._sb_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out = [torch.empty((b, d), dtype=torch.float32, device=self.device) if need_ent else None,
               torch.empty((b, d), dtype=torch.float32, device=self.device) if need_rel else None,
               torch.empty((n, d), dtype=torch.float32, device=self.device) if need_cand else None]
        N.check(self.lib.okge_prefix_score_backward(N.SCORERS[scorer] if isinstance(scorer, str) else int(scorer), 1 if sp else 0,
                                                    g.data_ptr(), g.stride(0), b, n, ent.data_ptr(), ent.stride(0), rel.data_ptr(),
                                                    rel.stride(0), cand.data_ptr(), cand.stride(0), d, _ptr(out[0]), _ptr(out[1]),
                                                    _ptr(out[2]), self._sb_ws.data_ptr(), self._sb_ws.numel(), self._stream()),
                "okge_prefix_score_backward")
        return out

    def scatter_rows(self, rows, ids, first_id, table_grad, drop: DropoutSpec = NO_DROP):
        """table_grad[id] += masked rows of the positions naming id, summed in sorted-position order (okge_scatter_rows)"""
        n, d = rows.shape
        order = None
        if ids is not None:
            ids = _i32(ids, self.device)
            order = torch.argsort(ids, stable=True).to(torch.int32)          # (index plumbing; the sums are the kernel's)
        dc = drop.c()
        N.check(self.lib.okge_scatter_rows(rows.data_ptr(), rows.stride(0), _ptr(ids), _ptr(order), int(first_id), n, d, ctypes.byref(dc),
                                           table_grad.data_ptr(), table_grad.shape[0], self._stream()), "okge_scatter_rows")
        return table_grad

    def scale_(self, x, alpha_dev):
        """x *= alpha (device fp32 scalar) in place."""
        N.check(self.lib.okge_scale_inplace(x.data_ptr(), x.numel(), alpha_dev.data_ptr(), self._stream()),
                "okge_scale_inplace")
        return x

    def rescale_gradients_(self, g0, g1, alpha_dev, applied):
        """g0, g1 *= alpha / applied (alpha: device fp32 scalar) in ONE launch; free when the two are the same number"""
        N.check(self.lib.okge_rescale_gradients(g0.data_ptr(), g0.numel(), g1.data_ptr(), g1.numel(), alpha_dev.data_ptr(),
                                                float(applied), self._stream()), "okge_rescale_gradients")

    def adagrad(self, p, g, state_sum, lr, weight_decay=1e-10, eps=1e-8, zero_grad=True):
        N.check(self.lib.okge_adagrad_step(p.data_ptr(), g.data_ptr(), state_sum.data_ptr(), p.numel(), float(lr),
                                           float(weight_decay), float(eps), 1 if zero_grad else 0, self._stream()),
                "okge_adagrad_step")

    def adagrad2(self, p0, g0, s0, p1, g1, s1, lr, weight_decay=1e-10, eps=1e-8, zero_grad=True):
        """Both tables in one launch.  zero_grad: False/0 none, True/1 both, 2 only g1 (see include/okge.h)."""
        N.check(self.lib.okge_adagrad_step2(p0.data_ptr(), g0.data_ptr(), s0.data_ptr(), p0.numel(), p1.data_ptr(),
                                            g1.data_ptr(), s1.data_ptr(), p1.numel(), float(lr), float(weight_decay),
                                            float(eps), int(zero_grad), self._stream()), "okge_adagrad_step2")

    def adagrad_multi(self, tensors, lr, weight_decay=1e-10, eps=1e-8):
        """One launch over up to four (p, g, state_sum[, touched_map, stamp[, rows]]) tuples, gradients cleared in the sweep
        (okge_adagrad_multi): a tensor with a touched-row byte map skips the gradient rows the map does not stamp."""
        arr = (N.AdagradTensor * len(tensors))()
        for a, t in zip(arr, tensors):
            p, g, s = t[:3]
            a.p, a.g, a.state_sum, a.n, a.zero_grad = p.data_ptr(), g.data_ptr(), s.data_ptr(), p.numel(), 1
            if len(t) > 3 and t[3] is not None:
                a.row_touched, a.row_len, a.touched_stamp = t[3].data_ptr(), p.shape[1], int(t[4])
                a.rows = int(t[5]) if len(t) > 5 else 0          # 0 all rows, 1 the unstamped rows only, 2 the stamped rows only
        N.check(self.lib.okge_adagrad_multi(arr, len(tensors), float(lr), float(weight_decay), float(eps), self._stream()),
                "okge_adagrad_multi")

    @staticmethod
    def lazy_tensors(tensors):
        """(p, g, state_sum[, row_steps, touched_map, stamp]) tuples -> the okge_lazy_tensor array of okge_adagrad_lazy"""
        arr = (N.LazyTensor * len(tensors))()
        for a, t in zip(arr, tensors):
            p, g, s = t[:3]
            a.p, a.g, a.state_sum = p.data_ptr(), g.data_ptr(), s.data_ptr()
            if len(t) > 3 and t[3] is not None:
                a.rows, a.row_len, a.row_steps = p.shape[0], p.shape[1], t[3].data_ptr()
                if len(t) > 4 and t[4] is not None:
                    a.row_touched, a.touched_stamp = t[4].data_ptr(), int(t[5])
            else:
                a.rows, a.row_len = 1, p.numel()
        return arr

    def adagrad_lazy(self, tensors, counters, window, flush, lr, weight_decay=1e-10, eps=1e-8):
        """okge_adagrad_lazy: the update with the weight-decay-only steps of rows no batch names deferred (flush=False: one
        optimizer step; flush=True: every row brought to the current step).  counters: int32[2] on the device."""
        arr = self.lazy_tensors(tensors)
        N.check(self.lib.okge_adagrad_lazy(arr, len(tensors), counters.data_ptr(), int(window), 1 if flush else 0, float(lr),
                                           float(weight_decay), float(eps), self._stream()), "okge_adagrad_lazy")

    def clip_grad_norm_(self, g0, g1, max_norm, norm_out=None):
        """torch.nn.utils.clip_grad_norm_ over two dense gradient tensors, in place (trainer.py:236-240)"""
        if getattr(self, "_clip_ws", None) is None:
            self._clip_ws = torch.empty(8448, dtype=torch.uint8, device=self.device)
        N.check(self.lib.okge_clip_grad_norm(g0.data_ptr(), g0.numel(), _ptr(g1), 0 if g1 is None else g1.numel(), float(max_norm),
                                             _ptr(norm_out), self._clip_ws.data_ptr(), self._clip_ws.numel(), self._stream()),
                "okge_clip_grad_norm")

    def merge_logsumexp(self, parts, out=None):
        """parts (world, B) fp32 per-shard row log-sum-exps -> (B,) log-sum-exp over the shards"""
        world, B = parts.shape
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.device)
        N.check(self.lib.okge_merge_logsumexp(parts.data_ptr(), world, B, out.data_ptr(), self._stream()), "okge_merge_logsumexp")
        return out

    def filtered_ranks(self, scores, filt_ptr, filt_col, row_ptr, grp_ptr, ids):
        """int64 ranks per answer group; all index arrays on the device (int64 ptr arrays, int32 ids)."""
        n_groups = int(grp_ptr.numel()) - 1
        ranks = torch.empty(n_groups, dtype=torch.int64, device=self.device)
        B, n = scores.shape
        N.check(self.lib.okge_filtered_ranks(scores.data_ptr(), scores.stride(0), B, n, filt_ptr.data_ptr(),
                                             _ptr(filt_col), row_ptr.data_ptr(), grp_ptr.data_ptr(), ids.data_ptr(),
                                             ranks.data_ptr(), self._stream()), "okge_filtered_ranks")
        return ranks

    def evaluate_batch(self, E, R, scorer, batch: PrefixBatch, filt_ptr, filt_col, row_ptr, grp_ptr, ids, scores, ranks,
                       acc, rank_stream):
        """score (current stream) -> filtered ranks -> meters (rank_stream) in one library call; `scores` (B, >=N) and
        `ranks` (>= n_groups int64) are caller buffers, `acc` 7 device doubles."""
        pb, c, keep = self._batch(batch)
        t = self._tables(E, R, scorer)
        ws = self.workspace(batch.B, c.n, t.d, "score")
        n_groups = int(grp_ptr.numel()) - 1
        N.check(self.lib.okge_evaluate_batch(ctypes.byref(t), ctypes.byref(pb), ctypes.byref(c), filt_ptr.data_ptr(),
                                             _ptr(filt_col), row_ptr.data_ptr(), grp_ptr.data_ptr(), ids.data_ptr(),
                                             n_groups, scores.data_ptr(), scores.stride(0), ranks.data_ptr(),
                                             acc.data_ptr(), ws.data_ptr(), self._ws_bytes, self._stream(),
                                             ctypes.c_void_p(rank_stream.cuda_stream)), "okge_evaluate_batch")
        del keep

    def evaluate_fused(self, E, R, scorer, batch: PrefixBatch, filt_ptr, filt_col, row_ptr, grp_ptr, ids, ranks=None, acc=None):
        """filtered ranks + meters of one evaluation batch without the (B, N) score block (okge_evaluate_fused): ranks
        (int64 per answer group, bit-equal to score() + filtered_ranks()) and acc (7 device doubles, accumulated)."""
        pb, c, keep = self._batch(batch)
        t = self._tables(E, R, scorer)
        n_groups, n_filter = int(grp_ptr.numel()) - 1, int(filt_col.numel())
        need = int(self.lib.okge_eval_workspace_bytes(batch.B, c.n, t.d, n_groups, n_filter))
        if need > self._ws_bytes:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_bytes = need
        if ranks is None:
            ranks = torch.empty(max(n_groups, 1), dtype=torch.int64, device=self.device)
        if acc is None:
            acc = torch.zeros(7, dtype=torch.float64, device=self.device)
        N.check(self.lib.okge_evaluate_fused(ctypes.byref(t), ctypes.byref(pb), ctypes.byref(c), filt_ptr.data_ptr(),
                                             _ptr(filt_col) if n_filter else None, n_filter, row_ptr.data_ptr(),
                                             grp_ptr.data_ptr(), ids.data_ptr(), n_groups, ranks.data_ptr(), acc.data_ptr(),
                                             self._ws.data_ptr(), self._ws_bytes, self._stream()), "okge_evaluate_fused")
        del keep
        return ranks[:n_groups], acc

    def evaluate_fused_shard(self, phase, E_local, R, scorer, Q, B, batch: PrefixBatch, shard: "Shard", n_cand_global, filt_ptr,
                             filt_col, row_ptr, grp_ptr, ids, true_scores, counts):
        """one launch of the candidate-sharded fused evaluation (okge_evaluate_fused_shard): phase 1 point scores ->
        true_scores (local maxima), 2 tile sweep, 4 counts; `batch` carries the LOCAL candidate range / list"""
        c = N.Candidates()
        cid = _i32(batch.cand_ids, self.device)
        c.ids, c.first_id, c.n = _ptr(cid), int(batch.cand_first), int(batch.n_candidates)
        t = self._tables(E_local, R, scorer)
        n_groups, n_filter = int(grp_ptr.numel()) - 1, int(filt_col.numel())
        need = int(self.lib.okge_eval_workspace_bytes(B, c.n, t.d, n_groups, n_filter))
        if phase == 1 and (getattr(self, "_ews", None) is None or self._ews.numel() < need):
            self._ews = torch.empty(need, dtype=torch.uint8, device=self.device)       # lives across the three phases
        sh = shard.c()
        N.check(self.lib.okge_evaluate_fused_shard(int(phase), ctypes.byref(t), ctypes.byref(sh), Q.data_ptr(), Q.stride(0), int(B),
                                                   ctypes.byref(c), int(n_cand_global), filt_ptr.data_ptr(),
                                                   _ptr(filt_col) if n_filter else None, n_filter, row_ptr.data_ptr(),
                                                   grp_ptr.data_ptr(), _ptr(ids), n_groups, true_scores.data_ptr(),
                                                   counts.data_ptr(), self._ews.data_ptr(), self._ews.numel(), self._stream()),
                "okge_evaluate_fused_shard")
        del cid

    def rank_metrics(self, ranks, acc):
        """acc (7 device doubles) += {n, sum 1/(r+1), sum r, #r<1, #r<3, #r<10, #r<50}"""
        N.check(self.lib.okge_rank_metrics(ranks.data_ptr(), int(ranks.numel()), acc.data_ptr(), self._stream()),
                "okge_rank_metrics")

    def group_true_scores(self, scores, col0, row_ptr, grp_ptr, ids):
        """float32 per answer group: max score over the group's ids inside columns [col0, col0 + n_local)."""
        out = torch.empty(int(grp_ptr.numel()) - 1, dtype=torch.float32, device=self.device)
        B, n = scores.shape
        N.check(self.lib.okge_group_true_scores(scores.data_ptr(), scores.stride(0), B, int(col0), n, row_ptr.data_ptr(),
                                                grp_ptr.data_ptr(), ids.data_ptr(), out.data_ptr(), self._stream()),
                "okge_group_true_scores")
        return out

    def rank_counts(self, scores, col0, filt_ptr, filt_col, row_ptr, true_scores):
        """int64 (n_groups, 2): {#greater, #equal} over the local candidate columns."""
        out = torch.empty((int(true_scores.numel()), 2), dtype=torch.int64, device=self.device)
        B, n = scores.shape
        N.check(self.lib.okge_rank_counts(scores.data_ptr(), scores.stride(0), B, int(col0), n, filt_ptr.data_ptr(),
                                          _ptr(filt_col), row_ptr.data_ptr(), true_scores.data_ptr(), out.data_ptr(),
                                          self._stream()), "okge_rank_counts")
        return out

    # -- measurement --------------------------------------------------------------------------------
    def timing(self, on):
        self.lib.okge_timing_enable(1 if on else 0)
        self.lib.okge_timing_reset()

    def timing_collect(self):
        return N.timing_collect()


def positives_from_dense(labels: torch.Tensor):
    """(B, N) {0,1} label matrix (what the reference's collate builds, dataset.py:885-932) -> coordinates
    sorted by column.  Container plumbing for the API-compatible entry points; the fused path consumes
    coordinates directly."""
    idx = labels.t().nonzero()               # sorted by (col, row)
    return idx[:, 1].to(torch.int32).contiguous(), idx[:, 0].to(torch.int32).contiguous()
