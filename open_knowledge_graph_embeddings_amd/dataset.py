"""Dataset-side types the model plugin consumes (openkge/dataset.py:25-39)."""
from __future__ import annotations

import ctypes
import queue
import threading
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _native as N
from .hotpath import PrefixBatch


@dataclass
class EntityRelationDatasetMeta:
    """Same fields as the reference's dataclass; the model reads entities_size, relations_size,
    min_entities_size and min_relations_size (ids 0 and 1 are PAD/UNK, openkge/index_mapper.py:14)."""
    entity_id_count_map: dict = field(default_factory=dict)
    relation_id_count_map: dict = field(default_factory=dict)
    entity_token_id_count_map: dict = field(default_factory=dict)
    relation_token_id_count_map: dict = field(default_factory=dict)
    entity_id_to_tokens_map: dict = field(default_factory=dict)
    relation_id_to_tokens_map: dict = field(default_factory=dict)
    entities_size: int = 0
    relations_size: int = 0
    min_entities_size: int = 2
    min_relations_size: int = 2
    entity_tokens_size: int = 0
    relation_tokens_size: int = 0
    max_length: int = 1


# ------------------------------------------------------------------------------------------------------------------
# Batch producer: OneToNMentionRelationDataset.get_loader + OneToNMentionRelationDataset_collate_func
# (openkge/dataset.py:455-478, :724-940) over the reference's three int32 tensors, emitting device-ready
# coordinate / CSR batches through the C ABI's host-side okge_collate_batch (csrc/okge_collate.cpp).
# ------------------------------------------------------------------------------------------------------------------


def pack_groups(groups):
    """utils/misc.py:56-70 (pack_list_of_lists): k answer groups -> [b_0+L, ..., b_k+L, 0, ids...], L = k+2."""
    lens = np.fromiter((len(g) for g in groups), dtype=np.int64, count=len(groups))
    header = np.concatenate([[0], np.cumsum(lens)]) + (len(groups) + 2)
    flat = np.fromiter((e for g in groups for e in g), dtype=np.int64, count=int(lens.sum()))
    return np.concatenate([header, [0], flat]).astype(np.int32)


@dataclass
class CollatedBatch:
    """What the reference's collate returns (dataset.py:937-940), in coordinate / CSR form on `device`."""
    batch: PrefixBatch                      # prefix ids, positives (col-sorted coordinates), candidate ids
    normalizer_loss: float                  # B * N
    normalizer_metric: float                # number of labels
    n_cand: int
    row_ptr: Optional[torch.Tensor] = None  # evaluation: answer groups (label_ids) as CSR of CSR ...
    grp_ptr: Optional[torch.Tensor] = None
    ids: Optional[torch.Tensor] = None
    filt_ptr: Optional[torch.Tensor] = None  # ... and the all-splits filter mask as CSR
    filt_col: Optional[torch.Tensor] = None

    def dense_labels(self):
        """(B, N) {0,1} tensor, as the reference builds it -- for API-compatible callers and tests only."""
        b = self.batch
        y = torch.zeros((b.B, self.n_cand), dtype=torch.float32, device=b.pos_row.device)
        y[b.pos_row.long(), b.pos_col.long()] = 1
        return y


class OneToNBatchProducer:
    """Iterates a split in batches of `batch_size` prefixes.  Constructor arguments follow
    OneToNMentionRelationDataset (dataset.py:352-382) / get_loader (:455-478) where they exist there."""

    def __init__(self, seen_prefixes_tensor, seen_entities_tensor, all_splits_entities_tensor, entity_vocab_size,
                 entity_special_vocab_size=2, batch_size=512, is_training_data=True, use_batch_shared_entities=False,
                 min_size_batch_labels=-1, device="cpu", shuffle=False, drop_last=True, seed=0, prefetch=2,
                 batches_per_call=16):
        i32 = lambda x: np.ascontiguousarray(np.asarray(x.cpu() if isinstance(x, torch.Tensor) else x, dtype=np.int32))  # noqa: E731
        self.prefixes = i32(seen_prefixes_tensor).reshape(-1, 7)
        self.seen = i32(seen_entities_tensor).reshape(-1)
        self.all_splits = i32(all_splits_entities_tensor if all_splits_entities_tensor is not None else []).reshape(-1)
        self.n_entities, self.offset = int(entity_vocab_size), int(entity_special_vocab_size)
        self.batch_size, self.training = int(batch_size), bool(is_training_data)
        self.shared, self.min_size = bool(use_batch_shared_entities), int(min_size_batch_labels or 0)
        self.device = torch.device(device)
        self.shuffle, self.drop_last, self.seed, self.prefetch = shuffle, drop_last, int(seed), max(1, int(prefetch))
        self.batches_per_call = max(1, int(batches_per_call))
        self.epoch = 0
        self._lib = N.lib()
        t = N.PrefixTable()
        t.prefixes, t.n_prefixes = self.prefixes.ctypes.data, self.prefixes.shape[0]
        t.seen_entities, t.n_seen = self.seen.ctypes.data, self.seen.shape[0]
        t.all_splits_entities, t.n_all = (self.all_splits.ctypes.data if self.all_splits.size else None), self.all_splits.shape[0]
        t.n_entities, t.entity_offset = self.n_entities, self.offset
        self._table = t
        self._len_this = (self.prefixes[:, 3] - self.prefixes[:, 2]).astype(np.int64)
        self._len_all = (self.prefixes[:, 5] - self.prefixes[:, 4]).astype(np.int64)

    def __len__(self):
        n = self.prefixes.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # -- one batch on the host -------------------------------------------------------------------------------------
    def collate_host(self, rows, seed=0):
        """rows: prefix-table indices in sampler order -> (arena tensor [pinned if a GPU is there], layout dict).
        All int32 outputs live in one arena so that the batch crosses PCIe in ONE copy."""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        B = int(rows.shape[0])
        if B == 0 or rows.min() < 0 or rows.max() >= self.prefixes.shape[0]:
            raise N.OkgeError("collate: empty batch or prefix row outside the table")
        cap_this = int(self._len_this[rows].sum())
        cap_all = 0 if self.training else int(self._len_all[rows].sum())
        cap_cand = max(self.min_size, cap_all if not self.training else cap_this) if self.shared else 0
        sizes = [("row_ptr", 2 * (B + 1)), ("grp_ptr", 2 * (cap_this + 1)), ("filt_ptr", 2 * (B + 1)),     # int64 first
                 ("po_rel", B), ("po_obj", B), ("sp_subj", B), ("sp_rel", B), ("pos_row", cap_this),
                 ("pos_col", cap_this), ("cand_ids", cap_cand), ("ids", cap_this), ("filt_col", cap_all)]
        if self.training:
            sizes = [(k, 0 if k in ("row_ptr", "grp_ptr", "filt_ptr", "ids", "filt_col") else n) for k, n in sizes]
        off, layout = 0, {}
        for k, n in sizes:
            layout[k] = (off, n)
            off += (n + 1) // 2 * 2                                      # keep every array 8-byte aligned
        arena = torch.empty(max(off, 2), dtype=torch.int32, pin_memory=self.device.type == "cuda")
        base = arena.data_ptr()
        c = N.Collated()
        c.cap_rows, c.cap_pos, c.cap_cand = B, cap_this, cap_cand
        c.cap_groups, c.cap_ids, c.cap_filter = cap_this, cap_this, cap_all
        for k, _ in sizes:
            setattr(c, k, base + 4 * layout[k][0])          # zero-capacity arrays get a valid, never-written address
        N.check(self._lib.okge_collate_batch(ctypes.byref(self._table), rows.ctypes.data, B, int(self.training),
                                             int(self.shared), self.min_size, ctypes.c_uint64(seed), ctypes.byref(c)),
                "okge_collate_batch")
        return arena, layout, c

    def to_device(self, arena, layout, c) -> CollatedBatch:
        dev = arena.to(self.device, non_blocking=True) if self.device.type == "cuda" else arena

        def i32(k, n):
            return dev[layout[k][0]:layout[k][0] + n]

        def i64(k, n):
            o = layout[k][0]
            return dev[o:o + 2 * n].view(torch.int64)

        n_po, n_sp, nnz = int(c.n_po), int(c.n_sp), int(c.nnz)
        pb = PrefixBatch(po_rel=i32("po_rel", n_po) if n_po else None, po_obj=i32("po_obj", n_po) if n_po else None,
                         sp_subj=i32("sp_subj", n_sp) if n_sp else None, sp_rel=i32("sp_rel", n_sp) if n_sp else None,
                         pos_row=i32("pos_row", nnz), pos_col=i32("pos_col", nnz),
                         cand_ids=i32("cand_ids", int(c.n_cand)) if self.shared else None, cand_unique=True,
                         cand_first=self.offset,
                         n_cand=int(c.n_cand))
        out = CollatedBatch(pb, float(c.normalizer_loss), float(c.normalizer_metric), int(c.n_cand))
        if not self.training:
            B = n_po + n_sp
            out.row_ptr, out.grp_ptr = i64("row_ptr", B + 1), i64("grp_ptr", int(c.n_groups) + 1)
            out.ids, out.filt_ptr = i32("ids", int(c.n_ids)), i64("filt_ptr", B + 1)
            out.filt_col = i32("filt_col", int(c.n_filter))
        return out

    def collate(self, rows, seed=0) -> CollatedBatch:
        return self.to_device(*self.collate_host(rows, seed))

    # -- epoch iteration: a host thread collates ahead of the device (ctypes releases the GIL) ---------------------
    def batch_rows(self):
        n = self.prefixes.shape[0]
        order = (np.random.default_rng(self.seed + self.epoch).permutation(n) if self.shuffle else np.arange(n)).astype(np.int64)
        stop = n - n % self.batch_size if self.drop_last else n
        return [order[i:i + self.batch_size] for i in range(0, stop, self.batch_size)]

    # -- several batches per library call: one arena, one H2D copy, little interpreter time per step ---------------
    def collate_group_host(self, rows_list, seed=0):
        """equal-sized batches -> (pinned arena, ctypes layout array); okge_collate_batches runs without the GIL"""
        K, B = len(rows_list), int(rows_list[0].shape[0])
        rows = np.ascontiguousarray(np.concatenate(rows_list), dtype=np.int64)
        if rows.min() < 0 or rows.max() >= self.prefixes.shape[0]:
            raise N.OkgeError("collate: prefix row outside the table")
        # per batch the library carves 8B + 5*len_this + len_all + cand capacity (+ alignment) int32 elements
        len_this, len_all = int(self._len_this[rows].sum()), 0 if self.training else int(self._len_all[rows].sum())
        cap = 6 * len_this + len_all + 10 * int(rows.shape[0]) + 64 * K
        if self.shared:
            cap += K * max(self.min_size, 0) + (len_this if self.training else len_all)
        arena = torch.empty(cap, dtype=torch.int32, pin_memory=self.device.type == "cuda")
        layout = (N.ArenaBatch * K)()
        used = ctypes.c_int64()
        N.check(self._lib.okge_collate_batches(ctypes.byref(self._table), rows.ctypes.data, K, B, int(self.training),
                                               int(self.shared), self.min_size, ctypes.c_uint64(seed), arena.data_ptr(), cap,
                                               layout, ctypes.byref(used)), "okge_collate_batches")
        return arena[:used.value], layout

    def group_to_device(self, arena, layout):
        """one copy for the whole group, then views"""
        dev = arena.to(self.device, non_blocking=True) if self.device.type == "cuda" else arena
        out = []
        for L in layout:
            i32 = lambda o, n: dev[o:o + n]                                   # noqa: E731
            i64 = lambda o, n: dev[o:o + 2 * n].view(torch.int64)             # noqa: E731
            n_po, n_sp = L.n_po, L.n_sp
            pb = PrefixBatch(po_rel=i32(L.off_po_rel, n_po) if n_po else None, po_obj=i32(L.off_po_obj, n_po) if n_po else None,
                             sp_subj=i32(L.off_sp_subj, n_sp) if n_sp else None, sp_rel=i32(L.off_sp_rel, n_sp) if n_sp else None,
                             pos_row=i32(L.off_pos_row, L.nnz), pos_col=i32(L.off_pos_col, L.nnz),
                             cand_ids=i32(L.off_cand, L.n_cand) if self.shared else None, cand_unique=True,
                             cand_first=self.offset, n_cand=int(L.n_cand))
            cb = CollatedBatch(pb, L.normalizer_loss, L.normalizer_metric, int(L.n_cand))
            if not self.training:
                B = n_po + n_sp
                cb.row_ptr, cb.grp_ptr = i64(L.off_row_ptr, B + 1), i64(L.off_grp_ptr, L.n_groups + 1)
                cb.ids, cb.filt_ptr, cb.filt_col = i32(L.off_ids, L.n_ids), i64(L.off_filt_ptr, B + 1), i32(L.off_filt_col, L.n_filter)
            out.append(cb)
        return out

    def __iter__(self):
        plan = self.batch_rows()
        self.epoch += 1
        q: "queue.Queue" = queue.Queue(maxsize=self.prefetch)
        K = self.batches_per_call
        # full-size batches go K at a time; a shorter last batch (drop_last=False) goes alone
        groups, cur = [], []
        for rows in plan:
            if cur and (len(cur) == K or rows.shape[0] != cur[0].shape[0]):
                groups.append(cur)
                cur = []
            cur.append(rows)
        if cur:
            groups.append(cur)

        def work():
            try:
                for i, g in enumerate(groups):
                    q.put(self.collate_group_host(g, seed=((self.seed << 20) ^ (self.epoch << 40) ^ i) & (2 ** 64 - 1)))
                q.put(None)
            except BaseException as e:          # surfaced in the consumer
                q.put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            yield from self.group_to_device(*item)
        th.join()


# ------------------------------------------------------------------------------------------------------------------
# On-disk formats: text files -> dataset tensors through the C ABI's host-side loader
# (csrc/okge_dataset.cpp), and the id-map files -> vocabulary sizes.
# ------------------------------------------------------------------------------------------------------------------
SPLITS = {"train": 0, "valid": 1, "test": 2}


def load_dataset_tensors(dataset_dir, train_input_file="train.txt", valid_input_file="valid.txt",
                         test_input_file="test.txt", max_size_prefix_label=-1):
    """OneToNMentionRelationDataset._collect_seen_triples + merge_all_splits_triples + create_data_tensors
    (dataset.py:480-710) in one call, no cache files written.
    -> {"train"|"valid"|"test": (seen_prefixes (P,7) int32, seen_entities int32)}, all_splits_entities int32,
       (max entity id, max relation id) seen in the files."""
    import os
    L = N.lib()
    h = ctypes.c_void_p()
    paths = [os.path.join(dataset_dir, f).encode() for f in (train_input_file, valid_input_file, test_input_file)]
    N.check(L.okge_dataset_open(paths[0], paths[1], paths[2], int(max_size_prefix_label), ctypes.byref(h)),
            "okge_dataset_open")
    try:
        out, all_splits, max_ids = {}, None, (0, 0)
        for name, s in SPLITS.items():
            n_p, n_s, n_a = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
            me, mr = ctypes.c_int32(), ctypes.c_int32()
            N.check(L.okge_dataset_sizes(h, s, ctypes.byref(n_p), ctypes.byref(n_s), ctypes.byref(n_a), ctypes.byref(me),
                                         ctypes.byref(mr)), "okge_dataset_sizes")
            pref = np.empty((n_p.value, 7), np.int32)
            seen = np.empty(n_s.value, np.int32)
            if all_splits is None:
                all_splits = np.empty(n_a.value, np.int32)
            N.check(L.okge_dataset_copy(h, s, pref.ctypes.data, seen.ctypes.data, all_splits.ctypes.data if s == 0 else None),
                    "okge_dataset_copy")
            out[name] = (pref, seen)
            max_ids = (me.value, mr.value)
        return out, all_splits, max_ids
    finally:
        L.okge_dataset_close(h)


def read_id_map_size(path):
    """`*_id_map.txt` (`# token\\tid\\tcount`, ids start at 2; index_mapper.py:95-108): vocabulary size = max id + 1,
    as EntityRelationDatasetBase.load_vocab computes it (dataset.py:172-184, :303-304)."""
    size = -1
    with open(path, encoding="utf-8") as f:
        for i, line in enumerate(f):
            if i == 0 and line.startswith("#"):
                continue
            parts = line.split("\t")
            if len(parts) >= 2:
                size = max(size, int(parts[1]))
    return size + 1


def dataset_meta(dataset_dir, entity_id_map_file="entity_id_map.txt", relation_id_map_file="relation_id_map.txt"):
    """EntityRelationDatasetMeta with the sizes the lookup models read (get_dataset_meta_dict, dataset.py:127-142)."""
    import os
    return EntityRelationDatasetMeta(entities_size=read_id_map_size(os.path.join(dataset_dir, entity_id_map_file)),
                                     relations_size=read_id_map_size(os.path.join(dataset_dir, relation_id_map_file)))
