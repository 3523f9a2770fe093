"""ctypes binding of libokge_hip.so (include/okge.h).  Thin: structs, argtypes, error -> exception.

The library is built in-tree by ``build_native()`` (hipcc --offload-arch=gfx950) and there is NO CPU
fallback: if it cannot be loaded, or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint8, c_uint32, \
    c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libokge_hip.so")
SOURCES = ["okge_api.hip", "okge_gemm.hip", "okge_train.hip", "okge_train32.hip", "okge_train64.hip", "okge_train64k.hip", "okge_misc.hip", "okge_pool.hip", "okge_collate.cpp", "okge_dataset.cpp"]
# every header a source may include: all of csrc/*.h (listed by the directory, so a new header cannot be forgotten) + the ABI
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "okge.h")]

OKGE_COMPLEX, OKGE_DISTMULT = 0, 1
OKGE_LOSS_BCE, OKGE_LOSS_KL = 0, 1
OKGE_TRAIN_GRADS_ZERO = 1
OKGE_TRAIN_LOSS_ONLY = 2
OKGE_TRAIN_UNIQUE_CANDIDATES = 4
OKGE_TRAIN_DISTINCT_PREFIX_ROWS = 8
OKGE_TRAIN_CLEAR_GRADS = 16
SCORERS = {"complex": OKGE_COMPLEX, "distmult": OKGE_DISTMULT}
LOSSES = {"bce": OKGE_LOSS_BCE, "kl": OKGE_LOSS_KL}

# every symbol include/okge.h declares
EXPORTS = ["okge_abi_version", "okge_last_error", "okge_score_prefixes", "okge_train_forward_backward",
           "okge_train_workspace_bytes", "okge_score_workspace_bytes", "okge_lse_workspace_bytes", "okge_query_ld", "okge_query_rows", "okge_encode_queries", "okge_train_tiles",
           "okge_prefix_backward", "okge_prefix_backward_segmented", "okge_fold_queries", "okge_score_queries", "okge_row_logsumexp", "okge_group_true_scores",
           "okge_rank_counts", "okge_rank_metrics", "okge_evaluate_batch", "okge_evaluate_fused", "okge_evaluate_fused_shard", "okge_evaluate_fused_batches", "okge_eval_workspace_bytes", "okge_score_triples", "okge_pool_workspace_bytes", "okge_pool_encode", "okge_pool_backward", "okge_pool_encode_calls", "okge_pool_backward_calls", "okge_pool_scatter_state_bytes", "okge_pool_backward_workspace_bytes", "okge_adagrad_multi", "okge_adagrad_lazy", "okge_pool_catch_up_calls", "okge_train_step", "okge_prefix_score_backward", "okge_prefix_score_backward_workspace_bytes", "okge_scatter_rows",
           "okge_collate_batch", "okge_collate_batches", "okge_dataset_open", "okge_dataset_sizes",
           "okge_dataset_copy", "okge_dataset_close", "okge_encode_rows", "okge_scale_inplace", "okge_rescale_gradients", "okge_adagrad_step", "okge_adagrad_step2", "okge_id_errors", "okge_clip_grad_norm", "okge_merge_logsumexp", "okge_filtered_ranks", "okge_timing_enable",
           "okge_timing_reset", "okge_timing_collect"]


class OkgeError(RuntimeError):
    pass


class Dropout(Structure):
    _fields_ = [("p", c_float), ("stream", c_uint32), ("step", c_uint32), ("_pad", c_uint32), ("seed", c_uint64),
                ("keep", c_void_p), ("step_dev", c_void_p)]


class PrefixBatch(Structure):
    _fields_ = [("po_rel", c_void_p), ("po_obj", c_void_p), ("sp_subj", c_void_p), ("sp_rel", c_void_p),
                ("n_po", c_int32), ("n_sp", c_int32),
                ("drop_po_ent", Dropout), ("drop_po_rel", Dropout), ("drop_sp_ent", Dropout), ("drop_sp_rel", Dropout)]


class Candidates(Structure):
    _fields_ = [("ids", c_void_p), ("first_id", c_int32), ("n", c_int32), ("drop", Dropout), ("table", c_void_p),
                ("table_rows", c_int32), ("_pad", c_int32)]


class EvalBatch(Structure):
    _fields_ = [("batch", PrefixBatch), ("cand", Candidates), ("filt_ptr", c_void_p), ("filt_col", c_void_p),
                ("n_filter", c_int64), ("row_ptr", c_void_p), ("grp_ptr", c_void_p), ("ids", c_void_p),
                ("n_groups", c_int64), ("rank_offset", c_int64)]


class Tables(Structure):
    _fields_ = [("E", c_void_p), ("R", c_void_p), ("n_ent", c_int32), ("n_rel", c_int32), ("d", c_int32),
                ("scorer", c_int32)]


class Shard(Structure):
    _fields_ = [("ent_lo", c_int32), ("ent_hi", c_int32), ("cand_col0", c_int32), ("_pad", c_int32)]


class Positives(Structure):
    _fields_ = [("col", c_void_p), ("row", c_void_p), ("nnz", c_int32)]


class TokenEmbedder(Structure):
    _fields_ = [("W", c_void_p), ("token_ids", c_void_p), ("vocab", c_int32), ("d", c_int32), ("n_ids", c_int32),
                ("max_len", c_int32), ("pool", c_int32), ("_pad", c_int32), ("bn_weight", c_void_p), ("bn_bias", c_void_p),
                ("bn_running_mean", c_void_p), ("bn_running_var", c_void_p), ("bn_eps", c_float), ("bn_momentum", c_float)]


class PoolCall(Structure):
    _fields_ = [("e", POINTER(TokenEmbedder)), ("ids", c_void_p), ("first_id", c_int32), ("n", c_int32), ("raw", c_void_p),
                ("out", c_void_p), ("ld", c_int64), ("saved", c_void_p), ("d_out", c_void_p), ("dW", c_void_p),
                ("d_bn_weight", c_void_p), ("d_bn_bias", c_void_p), ("row_touched", c_void_p), ("touched_stamp", c_int32),
                ("_pad", c_int32)]


class AdagradOpt(Structure):
    _fields_ = [("sum_E", c_void_p), ("sum_R", c_void_p), ("lr", c_float), ("weight_decay", c_float), ("eps", c_float),
                ("zero_entity_grad", c_int32), ("prefix_flags", c_void_p)]


class AdagradTensor(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("state_sum", c_void_p), ("n", c_int64), ("row_touched", c_void_p),
                ("row_len", c_int32), ("touched_stamp", c_int32), ("zero_grad", c_int32), ("rows", c_int32)]


class LazyTensor(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("state_sum", c_void_p), ("rows", c_int64), ("row_steps", c_void_p),
                ("row_touched", c_void_p), ("row_len", c_int32), ("touched_stamp", c_int32)]


class PrefixTable(Structure):
    _fields_ = [("prefixes", c_void_p), ("n_prefixes", c_int64), ("seen_entities", c_void_p), ("n_seen", c_int64),
                ("all_splits_entities", c_void_p), ("n_all", c_int64), ("n_entities", c_int32), ("entity_offset", c_int32)]


class Collated(Structure):
    _fields_ = [("cap_rows", c_int64), ("cap_pos", c_int64), ("cap_cand", c_int64), ("cap_groups", c_int64),
                ("cap_ids", c_int64), ("cap_filter", c_int64),
                ("po_rel", c_void_p), ("po_obj", c_void_p), ("sp_subj", c_void_p), ("sp_rel", c_void_p),
                ("pos_row", c_void_p), ("pos_col", c_void_p), ("cand_ids", c_void_p), ("row_ptr", c_void_p),
                ("grp_ptr", c_void_p), ("ids", c_void_p), ("filt_ptr", c_void_p), ("filt_col", c_void_p),
                ("n_po", c_int32), ("n_sp", c_int32), ("nnz", c_int64), ("n_cand", c_int64), ("n_groups", c_int64),
                ("n_ids", c_int64), ("n_filter", c_int64), ("normalizer_loss", c_double), ("normalizer_metric", c_double)]


class ArenaBatch(Structure):
    _fields_ = [(k, c_int64) for k in ("off_po_rel", "off_po_obj", "off_sp_subj", "off_sp_rel", "off_pos_row", "off_pos_col",
                                       "off_cand", "off_row_ptr", "off_grp_ptr", "off_ids", "off_filt_ptr", "off_filt_col",
                                       "nnz", "n_cand", "n_groups", "n_ids", "n_filter")] + \
               [("n_po", c_int32), ("n_sp", c_int32), ("normalizer_loss", c_double), ("normalizer_metric", c_double)]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_native(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU.  One object per source (compiled in parallel, rebuilt only when
    the source or a header is newer), then one link: ~15 s cold, a few seconds after a one-file edit."""
    if not force and not needs_build():
        return LIB_PATH
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj)
                and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src)))):
            continue
        cmd = ["hipcc"] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB_PATH


_LIB = None


def lib():
    """Load (never build implicitly on a box without hipcc) the C-ABI library; raise loudly if absent."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise OkgeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(hipcc --offload-arch=gfx950).  There is no CPU fallback for this path.")
    L = ctypes.CDLL(LIB_PATH)
    L.okge_abi_version.restype = c_int32
    L.okge_last_error.restype = c_char_p
    L.okge_train_workspace_bytes.restype = c_size_t
    L.okge_train_workspace_bytes.argtypes = [c_int32, c_int32, c_int32]
    L.okge_score_workspace_bytes.restype = c_size_t
    L.okge_score_workspace_bytes.argtypes = [c_int32, c_int32]
    L.okge_lse_workspace_bytes.restype = c_size_t
    L.okge_lse_workspace_bytes.argtypes = [c_int32, c_int32, c_int32]
    L.okge_score_prefixes.restype = c_int32
    L.okge_score_prefixes.argtypes = [POINTER(Tables), POINTER(PrefixBatch), POINTER(Candidates), c_void_p, c_int64,
                                      c_void_p, c_size_t, c_void_p]
    L.okge_train_forward_backward.restype = c_int32
    L.okge_train_forward_backward.argtypes = [POINTER(Tables), POINTER(PrefixBatch), POINTER(Candidates),
                                              POINTER(Positives), c_int32, c_float, c_double, c_int32, c_void_p,
                                              c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_size_t, c_void_p]
    L.okge_query_ld.restype = c_int64
    L.okge_query_ld.argtypes = [c_int32]
    L.okge_query_rows.restype = c_int32
    L.okge_query_rows.argtypes = [c_int32]
    L.okge_encode_queries.restype = c_int32
    L.okge_encode_queries.argtypes = [POINTER(Tables), POINTER(Shard), POINTER(PrefixBatch), c_void_p, c_int64, c_void_p,
                                      c_void_p]
    L.okge_fold_queries.restype = c_int32
    L.okge_fold_queries.argtypes = [POINTER(Tables), POINTER(PrefixBatch), c_void_p, c_int64, c_void_p, c_void_p]
    L.okge_train_tiles.restype = c_int32
    L.okge_train_tiles.argtypes = [POINTER(Tables), POINTER(Shard), c_void_p, c_int64, c_int32, POINTER(Candidates),
                                   POINTER(Positives), c_int32, c_float, c_double, c_int32, c_int32, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_score_queries.restype = c_int32
    L.okge_score_queries.argtypes = [POINTER(Tables), POINTER(Shard), c_void_p, c_int64, c_int32, POINTER(Candidates),
                                     c_void_p, c_int64, c_void_p]
    L.okge_row_logsumexp.restype = c_int32
    L.okge_row_logsumexp.argtypes = [POINTER(Tables), POINTER(Shard), c_void_p, c_int64, c_int32, POINTER(Candidates),
                                     c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_group_true_scores.restype = c_int32
    L.okge_group_true_scores.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p]
    L.okge_evaluate_batch.restype = c_int32
    L.okge_evaluate_batch.argtypes = [POINTER(Tables), POINTER(PrefixBatch), POINTER(Candidates), c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                      c_void_p, c_size_t, c_void_p, c_void_p]
    L.okge_rank_metrics.restype = c_int32
    L.okge_rank_metrics.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
    L.okge_rank_counts.restype = c_int32
    L.okge_rank_counts.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]
    L.okge_prefix_backward.restype = c_int32
    L.okge_prefix_backward.argtypes = [POINTER(Tables), POINTER(Shard), POINTER(PrefixBatch), c_void_p, c_int64, c_void_p,
                                       c_void_p, c_void_p, c_void_p]
    L.okge_prefix_backward_segmented.restype = c_int32
    L.okge_prefix_backward_segmented.argtypes = [POINTER(Tables), POINTER(Shard), POINTER(PrefixBatch), c_void_p, c_int64, c_void_p,
                                                 c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                                 c_void_p, c_void_p]
    L.okge_encode_rows.restype = c_int32
    L.okge_encode_rows.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, POINTER(Dropout), c_void_p,
                                   c_int64, c_void_p]
    L.okge_score_triples.restype = c_int32
    L.okge_score_triples.argtypes = [c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32,
                                     c_void_p, c_void_p]
    L.okge_collate_batch.restype = c_int32
    L.okge_collate_batch.argtypes = [POINTER(PrefixTable), c_void_p, c_int32, c_int32, c_int32, c_int32, c_uint64,
                                     POINTER(Collated)]
    L.okge_collate_batches.restype = c_int32
    L.okge_collate_batches.argtypes = [POINTER(PrefixTable), c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint64,
                                       c_void_p, c_int64, POINTER(ArenaBatch), POINTER(c_int64)]
    L.okge_dataset_open.restype = c_int32
    L.okge_dataset_open.argtypes = [c_char_p, c_char_p, c_char_p, c_int32, POINTER(c_void_p)]
    L.okge_dataset_sizes.restype = c_int32
    L.okge_dataset_sizes.argtypes = [c_void_p, c_int32, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                     POINTER(c_int32), POINTER(c_int32)]
    L.okge_dataset_copy.restype = c_int32
    L.okge_dataset_copy.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]
    L.okge_dataset_close.restype = None
    L.okge_dataset_close.argtypes = [c_void_p]
    L.okge_evaluate_fused.restype = c_int32
    L.okge_evaluate_fused.argtypes = [POINTER(Tables), POINTER(PrefixBatch), POINTER(Candidates), c_void_p, c_void_p, c_int64,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_evaluate_fused_shard.restype = c_int32
    L.okge_evaluate_fused_shard.argtypes = [c_int32, POINTER(Tables), POINTER(Shard), c_void_p, c_int64, c_int32, POINTER(Candidates),
                                            c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                            c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_evaluate_fused_batches.restype = c_int32
    L.okge_evaluate_fused_batches.argtypes = [POINTER(Tables), POINTER(EvalBatch), c_int32, c_void_p, c_void_p, c_void_p,
                                              c_size_t, POINTER(c_void_p), c_int32]
    L.okge_eval_workspace_bytes.restype = c_size_t
    L.okge_eval_workspace_bytes.argtypes = [c_int32, c_int32, c_int32, c_int64, c_int64]
    L.okge_pool_workspace_bytes.restype = c_size_t
    L.okge_pool_workspace_bytes.argtypes = [c_int32, c_int32]
    L.okge_pool_encode.restype = c_int32
    L.okge_pool_encode.argtypes = [POINTER(TokenEmbedder), c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64,
                                   c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_pool_backward.restype = c_int32
    L.okge_pool_backward.argtypes = [POINTER(TokenEmbedder), c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int64,
                                     c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_pool_encode_calls.restype = c_int32
    L.okge_pool_encode_calls.argtypes = [POINTER(PoolCall), c_int32, c_int32, c_void_p, c_size_t, c_void_p]
    L.okge_pool_backward_calls.restype = c_int32
    L.okge_pool_backward_calls.argtypes = [POINTER(PoolCall), c_int32, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]
    L.okge_pool_scatter_state_bytes.restype = c_size_t
    L.okge_pool_scatter_state_bytes.argtypes = [POINTER(PoolCall), c_int32]
    L.okge_pool_backward_workspace_bytes.restype = c_size_t
    L.okge_pool_backward_workspace_bytes.argtypes = [POINTER(PoolCall), c_int32]
    L.okge_prefix_score_backward_workspace_bytes.restype = c_size_t
    L.okge_prefix_score_backward_workspace_bytes.argtypes = [c_int32, c_int32, c_int32]
    L.okge_prefix_score_backward.restype = c_int32
    L.okge_prefix_score_backward.argtypes = [c_int32, c_int32, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_int64, c_void_p, c_int64,
                                             c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_scatter_rows.restype = c_int32
    L.okge_scatter_rows.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(Dropout), c_void_p,
                                    c_int32, c_void_p]
    L.okge_train_step.restype = c_int32
    L.okge_train_step.argtypes = [POINTER(Tables), POINTER(PrefixBatch), POINTER(Candidates), POINTER(Positives), c_int32, c_float,
                                  c_double, c_int32, POINTER(AdagradOpt), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_adagrad_multi.restype = c_int32
    L.okge_adagrad_multi.argtypes = [POINTER(AdagradTensor), c_int32, c_float, c_float, c_float, c_void_p]
    L.okge_adagrad_lazy.restype = c_int32
    L.okge_adagrad_lazy.argtypes = [POINTER(LazyTensor), c_int32, c_void_p, c_int32, c_int32, c_float, c_float, c_float, c_void_p]
    L.okge_pool_catch_up_calls.restype = c_int32
    L.okge_pool_catch_up_calls.argtypes = [POINTER(PoolCall), c_int32, POINTER(LazyTensor), c_int32, c_void_p, c_float, c_float, c_float, c_void_p]
    L.okge_scale_inplace.restype = c_int32
    L.okge_scale_inplace.argtypes = [c_void_p, c_int64, c_void_p, c_void_p]
    L.okge_rescale_gradients.restype = c_int32
    L.okge_rescale_gradients.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_float, c_void_p]
    L.okge_adagrad_step.restype = c_int32
    L.okge_adagrad_step.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_int32, c_void_p]
    L.okge_adagrad_step2.restype = c_int32
    L.okge_adagrad_step2.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64,
                                     c_float, c_float, c_float, c_int32, c_void_p]
    L.okge_id_errors.restype = c_int32
    L.okge_id_errors.argtypes = [POINTER(c_int64)]
    L.okge_clip_grad_norm.restype = c_int32
    L.okge_clip_grad_norm.argtypes = [c_void_p, c_int64, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_size_t, c_void_p]
    L.okge_merge_logsumexp.restype = c_int32
    L.okge_merge_logsumexp.argtypes = [c_void_p, c_int32, c_int32, c_void_p, c_void_p]
    L.okge_filtered_ranks.restype = c_int32
    L.okge_filtered_ranks.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p]
    L.okge_timing_enable.restype = c_int32
    L.okge_timing_enable.argtypes = [c_int32]
    L.okge_timing_reset.restype = c_int32
    L.okge_timing_collect.restype = c_int32
    L.okge_timing_collect.argtypes = [POINTER(c_char_p), POINTER(c_double), POINTER(c_int64), c_int32]
    if L.okge_abi_version() != 1:
        raise OkgeError("libokge_hip.so ABI version mismatch")
    _LIB = L
    return L


def id_errors():
    """number of out-of-range ids the kernels met since the last call (they substituted row 0); synchronises"""
    n = c_int64(0)
    check(lib().okge_id_errors(ctypes.byref(n)), "okge_id_errors")
    return int(n.value)


def check_ids():
    """raise if a kernel met an out-of-range id (the reference raises inside torch.nn.Embedding, model.py:457-460)"""
    n = id_errors()
    if n:
        raise OkgeError(f"{n} out-of-range id(s) reached the kernels (row 0 was used instead): entity / relation / candidate "
                        f"ids must lie inside their tables")


def check(rc, what):
    if rc != 0:
        msg = lib().okge_last_error()
        raise OkgeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def timing_collect():
    """{kernel name: (total_ms, launches)} since the last reset."""
    L = lib()
    cap = 32
    names = (c_char_p * cap)()
    ms = (c_double * cap)()
    cnt = (c_int64 * cap)()
    n = L.okge_timing_collect(names, ms, cnt, cap)
    return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}


_ = (c_uint8,)
