/*
 * okge.h -- C ABI of the MI355X-native open-KGE hot path (libokge_hip.so, gfx950).
 *
 * Drop-in boundary for the batched prefix-scoring training/evaluation loop of
 * samuelbroscheit/open_knowledge_graph_embeddings.  The reference has no FFI on this path: it is a
 * Python class protocol over ATen ops.  Each entry point below names the reference interface whose
 * arithmetic it replaces (paths relative to the reference checkout); the Python host classes in
 * open_knowledge_graph_embeddings_amd/ keep the reference's method names and call these through ctypes.
 * INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, explicit sizes/strides, no torch types;
 *   - every call enqueues on the HIP stream passed as `stream` (a hipStream_t cast to void*; NULL =
 *     default stream) and returns without synchronising;
 *   - return value: OKGE_OK (0) or a negative error code; okge_last_error() returns the message of
 *     the last failing call on the calling thread;
 *   - ids are int32 (reference: dataset.py:891,932), tables/scores/gradients fp32 row-major;
 *   - batch rows are ordered po-rows first, then sp-rows (trainer.py:69-71,91); either may be empty.
 */
#ifndef OKGE_H
#define OKGE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OKGE_ABI_VERSION 1

enum okge_status {
    OKGE_OK = 0,
    OKGE_ERR_INVALID = -1,      /* bad argument (shape, NULL pointer, unsupported size)            */
    OKGE_ERR_UNSUPPORTED = -2,  /* configuration outside the fused path (e.g. slot size too large) */
    OKGE_ERR_WORKSPACE = -3,    /* workspace too small; call okge_train_workspace_bytes()          */
    OKGE_ERR_HIP = -4           /* a HIP runtime call or kernel launch failed                      */
};

enum okge_scorer { OKGE_COMPLEX = 0, OKGE_DISTMULT = 1 };   /* model.py:176-240 / :243-278 */
enum okge_loss   { OKGE_LOSS_BCE = 0, OKGE_LOSS_KL = 1 };   /* trainer.py:93-106           */

/* Dropout applied to gathered embedding rows (model.py:461-462, F.dropout in training mode).
 * p == 0 disables it.  If `keep` is non-NULL it is an explicit keep-mask, uint8 [rows][d]
 * (1 = keep) -- used for parity runs against masks captured from the reference; otherwise the mask
 * is Philox4x32-10(key = seed, counter = (row position, column/8, stream, step)) read as eight 16-bit
 * numbers (element column&7: word (column&7)/2, low half if even, high half if odd),
 * keep <=> number >= floor(p * 65536).  Kept values are multiplied by 1/(1-p). */
typedef struct okge_dropout {
    float          p;
    uint32_t       stream;
    uint32_t       step;
    uint32_t       _pad;
    uint64_t       seed;
    const uint8_t *keep;
    /* Optional DEVICE counter: if non-NULL the kernels read `step` from it when they run (and ignore the
     * host value above), so a captured HIP graph draws a fresh mask on every replay. */
    const uint32_t *step_dev;
} okge_dropout;

/* One batch of prefixes.  Replaces the `inputs` list AddLossModule.forward receives
 * (trainer.py:48-71): [ (rel(b0,1), obj(b0,1)) | None , (subj(b1,1), rel(b1,1)) | None ]. */
typedef struct okge_prefix_batch {
    const int32_t *po_rel;  /* [n_po] relation ids of the (?, r, o) rows   */
    const int32_t *po_obj;  /* [n_po] object entity ids                    */
    const int32_t *sp_subj; /* [n_sp] subject entity ids of (s, r, ?) rows */
    const int32_t *sp_rel;  /* [n_sp] relation ids                         */
    int32_t        n_po;
    int32_t        n_sp;
    okge_dropout   drop_po_ent, drop_po_rel, drop_sp_ent, drop_sp_rel;
} okge_prefix_batch;

/* Candidate entity set shared by every row of the batch (trainer.py:75-87).
 * ids == NULL means the contiguous range first_id .. first_id + n - 1 (1-vs-all: first_id = 2,
 * model.py:512-523 `weight[min_entities_size:]`); otherwise ids[n] (batch-shared sample,
 * model.py:76-77); an id may repeat (its gradient rows are then accumulated with atomics, see
 * OKGE_TRAIN_UNIQUE_CANDIDATES). */
typedef struct okge_candidates {
    const int32_t *ids;
    int32_t        first_id;
    int32_t        n;
    okge_dropout   drop;
    /* Optional: gather candidate rows from this (table_rows, d) fp32 table instead of okge_tables.E -- the
     * reference's _score(subj, rel, obj, prefix=True) receives already-encoded candidate rows
     * (model.py:181-229).  Scoring only; the training entry point requires table == NULL. */
    const float   *table;
    int32_t        table_rows;
    int32_t        _pad;
} okge_candidates;

/* Embedding tables: entity_embedding.weight (n_ent, d), relation_embedding.weight (n_rel, d)
 * (model.py:390-391), contiguous fp32. */
typedef struct okge_tables {
    float  *E;
    float  *R;
    int32_t n_ent;
    int32_t n_rel;
    int32_t d;
    int32_t scorer; /* enum okge_scorer */
} okge_tables;

/* Positive labels of the batch as coordinates (row in [0,B), column in [0,N) = candidate position),
 * SORTED BY COLUMN (ties in any order).  Replaces the dense (B,N) fp32 label tensor the reference's
 * collate function builds (dataset.py:885-932); labels are {0,1} there. */
typedef struct okge_positives {
    const int32_t *col;
    const int32_t *row;
    int32_t        nnz;
} okge_positives;

int         okge_abi_version(void);
const char *okge_last_error(void);

/* ---- scoring (evaluation and the API-compatible *_prefix_score methods) ---------------------------
 * Replaces RelationScorer.po_prefix_score / sp_prefix_score -> _score(prefix=True)
 * (model.py:52-74, ComplEx :198-229, DistMult :268-274) together with the embedder calls they make
 * (model.py:455-510): gathers and (optionally) drops out the prefix rows, folds them into one query
 * row each, and writes scores[B][ld_scores] = query . candidate for every candidate. */
int okge_score_prefixes(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                        float *scores, int64_t ld_scores, void *workspace, size_t workspace_bytes,
                        void *stream);

/* ---- fused training step: forward + loss + backward ------------------------------------------------
 * Replaces AddLossModule.forward (trainer.py:48-113) followed by
 * `(loss.sum() / normalizer).backward()` (trainer.py:217-234) for
 * scorer in {ComplEx, DistMult} x LookupSimpleRelationEmbedder x loss in {bce, kl}.
 *   loss_out  : device double[1]; receives the SUMMED loss (what AddLossModule returns as `result`)
 *   dE, dR    : dense gradient buffers (n_ent, d) / (n_rel, d); the step ACCUMULATES into them
 *               (like autograd's .grad); they must be zero (or hold earlier accumulation) on entry
 *   scores    : optional [B][ld_scores] output of all_outputs (NULL to skip -- training never needs it)
 *   normalizer: the reference divides the loss by B*N before backward (dataset.py:935)
 *   flags     : OKGE_TRAIN_GRADS_ZERO -- the caller guarantees dE and dR are all-zero on entry (fresh
 *               zero_grad, trainer.py:229-232); candidate rows are then stored instead of read-modify-written
 * No (B,N) label or score tensor is read; positives come as coordinates. */
#define OKGE_TRAIN_GRADS_ZERO 1
/*               OKGE_TRAIN_LOSS_ONLY  -- forward + loss only (validation loss under torch.no_grad(),
 *               trainer.py:363-369): dE/dR are not touched and may be NULL */
#define OKGE_TRAIN_LOSS_ONLY 2
/*               OKGE_TRAIN_UNIQUE_CANDIDATES -- cand->ids names every entity at most once (true for the lists
 *               okge_collate_batch emits): their gradient rows are written without atomics.  Without the flag an
 *               explicit id list may repeat entities (precompute_batch_shared_inputs takes any list). */
#define OKGE_TRAIN_UNIQUE_CANDIDATES 4
/*               OKGE_TRAIN_DISTINCT_PREFIX_ROWS -- every prefix entity id and every prefix relation id of the batch occurs
 *               ONCE and no prefix entity is also a candidate (the "virtual tables" of already encoded rows that the
 *               token-pooled embedder and the embedder variants score: row = position, model.py:762-786 / :463-479):
 *               their gradient rows are STORED, not accumulated -- no float atomics, and the caller need not clear them. */
#define OKGE_TRAIN_DISTINCT_PREFIX_ROWS 8
/*               OKGE_TRAIN_CLEAR_GRADS -- dE and dR may hold anything on entry: the call stores the candidate rows of dE (as
 *               with OKGE_TRAIN_GRADS_ZERO) and itself clears what it only accumulates into -- all of dR, the rows of dE
 *               outside the candidate range -- inside its first launch.  What zero_grad + a fresh .grad buffer are in the
 *               reference (trainer.py:229-234) without two fill launches per step.  Contiguous candidate range
 *               (cand->ids == NULL), unsharded table. */
#define OKGE_TRAIN_CLEAR_GRADS 16
int okge_train_forward_backward(const okge_tables *t, const okge_prefix_batch *batch,
                                const okge_candidates *cand, const okge_positives *pos,
                                int32_t loss_kind, float label_smoothing, double normalizer, int32_t flags,
                                double *loss_out, float *dE, float *dR,
                                float *scores, int64_t ld_scores,
                                void *workspace, size_t workspace_bytes, void *stream);

/* ---- the same step in three phases, for an entity table row-sharded across devices -------------------------
 * (the reference has no counterpart: its only multi-device mechanism is nn.DataParallel, trainer.py:143-145.)
 * Rank r holds entity rows [ent_lo, ent_hi) (okge_tables.E points at its slice, n_ent = ent_hi - ent_lo) plus
 * the whole relation table.  One step on every rank, with two small all-reduces supplied by the caller (RCCL):
 *   1. okge_encode_queries  : the masked prefix entity rows (Q == NULL) of the prefixes whose entity this rank
 *                             owns, zeros elsewhere                 -> all-reduce(sum) of ent_rows  (B x d floats)
 *      okge_fold_queries    : every rank folds them with its replicated relation rows into the query block
 *                             (passing Q to okge_encode_queries instead computes the owned query rows at once;
 *                             Q would then need its own all-reduce)
 *   2. okge_train_tiles     : score/loss/dCand against the LOCAL candidates, dE of the local rows, partial dQ
 *                                                                   -> all-reduce(sum) of dQ (and of the loss)
 *   3. okge_prefix_backward : chain rule; entity gradients scattered by the owner, relation gradients formed
 *                             identically on every rank from ent_rows.
 * cand_col0 = position of the first local candidate in the un-sharded candidate list: positives keep their
 * global columns and dropout masks are identical to the single-device run.
 * KL loss: softmax runs over ALL candidates, so between 1 and 2 each rank calls okge_row_logsumexp on its local
 * candidates, the B values are exchanged (all-gather, log-sum-exp over ranks) and the result goes to
 * okge_train_tiles as `row_lse` (null for BCE).
 * Evaluation: okge_score_queries on the local candidates, okge_group_true_scores
 * -> all-reduce(max), okge_rank_counts -> all-reduce(sum); rank = #greater + #equal / 2 (dataset.py:441-446). */
typedef struct okge_shard {
    int32_t ent_lo, ent_hi;
    int32_t cand_col0;
    int32_t _pad;
} okge_shard;

int64_t okge_query_ld(int32_t d);      /* leading dimension (floats) of Q / ent_rows / dQ blocks for slot size d */
int32_t okge_query_rows(int32_t B);    /* rows those blocks must have (B rounded up to the 64-row chunk)         */

int okge_encode_queries(const okge_tables *t, const okge_shard *shard, const okge_prefix_batch *batch,
                        float *Q, int64_t ldq, float *ent_rows, void *stream);
int okge_fold_queries(const okge_tables *t, const okge_prefix_batch *batch, const float *ent_rows, int64_t ldq,
                      float *Q, void *stream);
int okge_train_tiles(const okge_tables *t, const okge_shard *shard, const float *Q, int64_t ldq, int32_t B,
                     const okge_candidates *local_cand, const okge_positives *pos, int32_t loss_kind,
                     float label_smoothing, double normalizer, int32_t n_cand_global, int32_t flags,
                     const float *row_lse, double *loss_out, float *dE, float *dQ, void *workspace,
                     size_t workspace_bytes, void *stream);
/* scores[b][j] = Q[b] . dropout(E_local[candidate j]) for the local candidates (evaluation / KL statistics) */
int okge_score_queries(const okge_tables *t, const okge_shard *shard, const float *Q, int64_t ldq, int32_t B,
                       const okge_candidates *local_cand, float *scores, int64_t ld_scores, void *stream);
/* row_lse[b] = log sum_j exp(score[b][j]) over the local candidates, without materialising the scores
 * (the log_softmax denominator of trainer.py:99-101, per shard) */
int okge_row_logsumexp(const okge_tables *t, const okge_shard *shard, const float *Q, int64_t ldq, int32_t B,
                       const okge_candidates *local_cand, float *row_lse, void *workspace, size_t workspace_bytes,
                       void *stream);
int okge_prefix_backward(const okge_tables *t, const okge_shard *shard, const okge_prefix_batch *batch,
                         const float *dQ, int64_t ldq, const float *ent_rows, float *dE, float *dR, void *stream);
/* The same with the relation and / or entity gradients formed by SORTED SEGMENTS instead of float atomics: every batch row's
 * gradient rows are stored into grad_rows ([2][okge_query_rows(B)][ldq] scratch: relation rows, then entity rows) and one
 * workgroup per distinct relation / entity adds its rows up in a fixed order and does ONE read-modify-write of the table row
 * (reproducible; no atomic traffic: 1.6 M float atomics per step at the 8-GPU FB15k-237 shape, 4096 batch rows on 237
 * relation rows).  The plans are host work on ids the host already has (sharded.make_row_segments): x_order[B] = batch rows
 * (po rows first, as everywhere) sorted stably by relation id / prefix entity id, x_seg_ptr[n_seg + 1] = bounds of the runs of
 * equal ids; either plan may be absent (NULL, 0: that table keeps the atomics).  Needs d % 8 == 0 (ComplEx) or d % 4 == 0
 * (DistMult).  embedding_dense_backward of the two tables, trainer.py:234. */
int okge_prefix_backward_segmented(const okge_tables *t, const okge_shard *shard, const okge_prefix_batch *batch,
                                   const float *dQ, int64_t ldq, const float *ent_rows, const int32_t *rel_order,
                                   const int32_t *rel_seg_ptr, int32_t n_rel_seg, const int32_t *ent_order,
                                   const int32_t *ent_seg_ptr, int32_t n_ent_seg, float *grad_rows, float *dE, float *dR,
                                   void *stream);

/* Bytes of scratch okge_train_forward_backward / okge_train_tiles need for a batch of B rows against N candidates with
 * slot size d (0 on invalid arguments).  Training sweeps the candidates in ranges (default: G^T of one range <= 1 GiB,
 * environment OKGE_GT_MBYTES), so this grows with B x min(N, range), not with B x N: the reference's autograd graph
 * keeps several (B, N) fp32 tensors alive instead (trainer.py:75-106). */
size_t okge_train_workspace_bytes(int32_t B, int32_t N, int32_t d);
/* Scratch of the scoring-only calls (okge_score_prefixes, okge_evaluate_batch): the folded query block. */
size_t okge_score_workspace_bytes(int32_t B, int32_t d);
/* Scratch of okge_row_logsumexp: query block + one range of (max, sum-exp) tile statistics. */
size_t okge_lse_workspace_bytes(int32_t B, int32_t N, int32_t d);

/* ---- embedder --------------------------------------------------------------------------------------
 * Replaces LookupBaseRelationEmbedder._encode for the lookup embedder with batch-norm / projection /
 * normalisation off (model.py:455-480): out[i][:] = dropout(table[ids ? ids[i] : first_id + i][:]).
 * Used by the API-compatible encode_subj/rel/obj, get_all_* and precompute_batch_shared_inputs. */
int okge_encode_rows(const float *table, int32_t table_rows, int32_t d, const int32_t *ids, int32_t first_id,
                     int32_t n, const okge_dropout *drop, float *out, int64_t ld_out, void *stream);

/* ---- token-pooled embedder ------------------------------------------------------------------------
 * Replaces UnigramPoolingRelationEmbedder._encode (model.py:762-786) up to (not including) its final dropout,
 * which the consumers apply while gathering rows (okge_dropout on the virtual tables):
 *   row id -> token_ids[id][0..max_len) (right-padded with 0; TokenBasedRelationEmbedder, model.py:579-597)
 *          -> sum | mean (/(#tokens>0 + 1e-12)) | max over the token embedding rows; padded positions take part
 *             (row 0 of the table is an ordinary row, only its gradient is suppressed: padding_idx)
 *          -> BatchNorm1d(eps, momentum) if bn_weight != NULL: training != 0 uses the statistics of the n rows of
 *             THIS call and updates the running statistics (unbiased variance), else the running statistics.
 * okge_pool_encode : raw[n][ld] = pooled rows (kept for backward), out[n][ld] = normalised rows (== raw allowed
 *                    without batch-norm), saved[4*d] = {mean, rstd, scratch, scratch} of this call.
 * okge_pool_backward: d_out[n][ld] -> batch-norm backward (d_bn_weight / d_bn_bias += this call's sums)
 *                    -> scatter-add into the dense token-table gradient dW (vocab x d; row 0 untouched). */
typedef struct okge_token_embedder {
    const float *W;              /* token embedding table (vocab x d) */
    const int32_t *token_ids;    /* (n_ids x max_len) */
    int32_t vocab, d, n_ids, max_len;
    int32_t pool;                /* 0 = sum, 1 = mean, 2 = max */
    int32_t _pad;
    const float *bn_weight, *bn_bias;      /* NULL: no batch-norm */
    float *bn_running_mean, *bn_running_var;
    float bn_eps, bn_momentum;
} okge_token_embedder;

/* One _encode call of a BATCH of calls.  A training step of the token-pooled models makes five (candidates, po relations,
 * po objects, sp subjects, sp relations: trainer.py:75-91); issued one by one they are 35 small launches per step, as a
 * batch three forward + three backward, each filling the chip.  Batch-norm statistics stay PER CALL; running statistics
 * (forward) and parameter gradients (backward) are updated in the order of the array -- the reference's calls update one
 * module's buffers in sequence.  Forward uses e, ids / first_id, n (> 0), raw, out, ld, saved; backward also d_out, dW,
 * d_bn_weight, d_bn_bias (raw and saved as the forward left them).  Workspace: the sum of okge_pool_workspace_bytes(n, d)
 * over the calls. */
typedef struct okge_pool_call {
    const okge_token_embedder *e;
    const int32_t *ids;
    int32_t first_id, n;
    float *raw, *out;
    int64_t ld;
    float *saved;
    const float *d_out;
    float *dW, *d_bn_weight, *d_bn_bias;
    uint8_t *row_touched;        /* optional: [vocab] bytes; backward: row_touched[t] = touched_stamp for every row t of dW written;
                                    forward (training != 0): for every token row the call reads (padding row 0 included) */
    int32_t touched_stamp;       /* 1..255 (okge_adagrad_multi reads the map with the same stamp) */
    int32_t _pad;
} okge_pool_call;
int okge_pool_encode_calls(const okge_pool_call *calls, int32_t n_calls, int32_t training, void *workspace,
                           size_t workspace_bytes, void *stream);
/* Backward of a batch of calls.  The scatter-add into the token tables' dense gradients
 * (torch.nn.Embedding's backward under model.py:762-771) runs
 *   scatter_state == NULL: with float atomics (any pooling; the sums depend on arrival order in the last bits);
 *   scatter_state != NULL: store-and-sum through an inverted index the kernels build from the batch's token ids -- every
 *     gradient row is added up by one owner in ascending (row, position) order: BIT-REPRODUCIBLE, and ~2x the speed at
 *     BASELINE configs[4].  Needs sum / mean pooling and slot sizes that are a multiple of 4 (else OKGE_ERR_UNSUPPORTED).
 *     scatter_state: okge_pool_scatter_state_bytes(calls) bytes that the caller zeroes ONCE and thereafter only hands to
 *     this function (it leaves them zero; one call at a time per buffer); workspace: okge_pool_backward_workspace_bytes.
 * row_touched / touched_stamp (optional, both paths): a byte per table row for okge_adagrad_multi, so the optimizer
 * sweep skips the gradient rows no token of the batch named. */
int okge_pool_backward_calls(const okge_pool_call *calls, int32_t n_calls, void *workspace, size_t workspace_bytes,
                             void *scatter_state, size_t scatter_state_bytes, void *stream);
size_t okge_pool_scatter_state_bytes(const okge_pool_call *calls, int32_t n_calls);
size_t okge_pool_backward_workspace_bytes(const okge_pool_call *calls, int32_t n_calls);
size_t okge_pool_workspace_bytes(int32_t n, int32_t d);
int okge_pool_encode(const okge_token_embedder *e, const int32_t *ids, int32_t first_id, int32_t n, int32_t training,
                     float *raw, float *out, int64_t ld, float *saved, void *workspace, size_t workspace_bytes,
                     void *stream);
int okge_pool_backward(const okge_token_embedder *e, const int32_t *ids, int32_t first_id, int32_t n,
                       const float *raw, const float *d_out, int64_t ld, float *saved, float *dW,
                       float *d_bn_weight, float *d_bn_bias, void *workspace, size_t workspace_bytes, void *stream);

/* ---- gradients of the plugin methods' scores (a caller's own loss) ----------------------------------------
 * Backward of sp_prefix_score / po_prefix_score / _score(prefix=True) (model.py:52-77, :198-229, :268-274) for a caller that
 * holds the dense (b, n) gradient g of the scores (the reference's autograd walks its four / one matrix products backwards):
 *   q = fold(ent, rel)   sp: [e1 r1 - e2 r2, e2 r1 + e1 r2]   po: [e1 r1 + e2 r2, e2 r1 - e1 r2]   DistMult: e * r
 *   d_cand [n][d] = g^T . q        dq = g . cand -> d_ent, d_rel [b][d] by the transpose of the fold
 * ent / rel / cand are the ENCODED rows the forward saw (dropout already applied); any of the three outputs may be NULL.
 * Both products run on a hand-written exact-fp32 MFMA kernel; the split contraction is reduced in a fixed order
 * (bit-reproducible).  The fused training path (okge_train_forward_backward) never forms g and does not come through here. */
size_t okge_prefix_score_backward_workspace_bytes(int32_t b, int32_t n, int32_t d);
int okge_prefix_score_backward(int32_t scorer, int32_t sp, const float *g, int64_t ld_g, int32_t b, int32_t n, const float *ent,
                               int64_t ld_ent, const float *rel, int64_t ld_rel, const float *cand, int64_t ld_cand, int32_t d,
                               float *d_ent, float *d_rel, float *d_cand, void *workspace, size_t workspace_bytes, void *stream);
/* Backward of okge_encode_rows (torch.nn.Embedding's backward under model.py:455-470): table_grad[id] += sum of the rows of the
 * positions that named id, each multiplied by its dropout mask (`drop` as given to the forward; NULL: none).  `order` = the
 * positions sorted by id (stable): every table row is added up by ONE owner in that order -- no float atomics, bit-reproducible.
 * ids == NULL: position i names row first_id + i.  Row 0 (padding_idx) receives nothing. */
int okge_scatter_rows(const float *rows, int64_t ld, const int32_t *ids, const int32_t *order, int32_t first_id, int32_t n, int32_t d,
                      const okge_dropout *drop, float *table_grad, int32_t table_rows, void *stream);

/* Per-triple scores of ENCODED rows, Hadamard form: RelationScorer.triple_score / forward(subj, rel, obj)
 * (model.py:43-50; ComplEx :231-238  sum s1 r1 o1 + s2 r1 o2 + s1 r2 o2 - s2 r2 o1;  DistMult :276  sum s r o).
 * Inference helper: the reference trains through the prefix path only (trainer.py:59-64). */
int okge_score_triples(int32_t scorer, const float *subj, int64_t ld_subj, const float *rel, int64_t ld_rel,
                       const float *obj, int64_t ld_obj, int32_t n, int32_t d, float *out, void *stream);

/* x[i] *= *alpha_dev for i < n (alpha is a DEVICE fp32 scalar: the upstream gradient autograd hands to the
 * fused loss node, i.e. 1/normalizer of trainer.py:221, without a host synchronisation). */
int okge_scale_inplace(float *x, int64_t n, const float *alpha_dev, void *stream);
/* The same hand-over when the caller already folded a factor into the gradients: the reference Trainer always divides the
 * summed loss by normalizer_loss = B x N (dataset.py:935, trainer.py:221), so AddLossModule runs the fused step with that
 * normalizer and the gradients need NO pass at all when autograd's upstream scalar turns out to be the same fp32 number:
 * g0, g1 *= *alpha_dev / applied, in one launch that reads nothing but the scalar when the ratio is exactly 1
 * (otherwise it rescales: correct for any upstream gradient, only slower). */
int okge_rescale_gradients(float *g0, int64_t n0, float *g1, int64_t n1, const float *alpha_dev, float applied, void *stream);

/* ---- dense Adagrad ----------------------------------------------------------------------------------
 * Replaces torch.optim.Adagrad.step as configured by OptimRegime (utils/optim.py:29,139-160):
 * g += wd*p; sum += g*g; p -= lr * g / (sqrt(sum) + eps), over all n elements.  If zero_grad != 0 the
 * gradient buffer is cleared in the same sweep (replaces optimizer.zero_grad(), trainer.py:229-244). */
int okge_adagrad_step(float *p, float *g, float *state_sum, int64_t n, float lr, float weight_decay,
                      float eps, int32_t zero_grad, void *stream);
/* The same update on two parameter tensors (entity and relation table) in ONE launch.
 * zero_grad: 0 = keep both gradient buffers, 1 = clear both, 2 = clear only g1 -- for the 1-vs-all step, whose next
 * okge_train_forward_backward(OKGE_TRAIN_GRADS_ZERO) overwrites every candidate row of dE anyway (rows below the first
 * candidate are never written and stay zero), so clearing 4*|E|*d bytes per step would be wasted traffic. */
int okge_adagrad_step2(float *p0, float *g0, float *sum0, int64_t n0, float *p1, float *g1, float *sum1,
                       int64_t n1, float lr, float weight_decay, float eps, int32_t zero_grad, void *stream);

/* The same update on up to four tensors in ONE launch (token tables + batch-norm parameters of the token-pooled models).
 * row_touched (optional, needs zero_grad and n % row_len == 0, row_len % 4 == 0): a byte per row of row_len floats; a row
 * whose byte differs from touched_stamp holds an all-zero gradient BY CONTRACT (okge_pool_backward_calls stamps every row it
 * writes) and its gradient is neither read nor cleared -- the row still takes the reference's weight-decay-only update
 * (utils/optim.py:139-160 applies wd to all rows).  The map is not erased: use another stamp (1..255) for the next update. */
typedef struct okge_adagrad_tensor {
    float *p, *g, *state_sum;
    int64_t n;
    const uint8_t *row_touched;
    int32_t row_len, touched_stamp, zero_grad;
    int32_t rows;                /* 0: all rows; 1: only the rows WITHOUT the stamp (weight-decay-only update, the gradient is not read);
                                    2: only the rows WITH the stamp.  1 then 2 = 0, row for row: the rows no token of the batch names can
                                    take their update while the step's matrix kernels run (okge_pool_encode_calls stamps the rows the
                                    forward reads, so the sweep of the others may start right behind it on another stream) */
} okge_adagrad_tensor;
int okge_adagrad_multi(const okge_adagrad_tensor *tensors, int32_t n_tensors, float lr, float weight_decay, float eps,
                       void *stream);

/* ---- dense Adagrad with the weight-decay-only updates deferred ("lazy decay") ---------------------------------------------
 * The reference's optimizer reaches EVERY row of a table in every step: with weight_decay != 0 (1e-10 in all its configs)
 * even a row no gradient reached moves by its own decay term (utils/optim.py:139-160, dense gradients).  For a token table
 * that is most rows (85 % at BASELINE configs[4]): a read-modify-write of the whole table and its accumulator per step.
 * Such an update depends on that row's (p, state_sum) alone, so it can be applied later -- all pending steps at once, in
 * registers, the same operations in the same order -- provided it has happened before anything READS the row:
 *   row_steps[r]    optimizer steps row r has seen;  counters[0] = steps taken (T),  counters[1] = scratch (both start 0)
 *   okge_pool_catch_up_calls   before the pooling forward: the rows the batch's tokens name are brought to T
 *   okge_adagrad_lazy(OKGE_LAZY_STEP)   rows carrying touched_stamp: their pending steps, then this step with their
 *                   gradient (cleared; the map byte goes back to 0); rows with r % window == T % window: their pending
 *                   steps and this one; tensors without row_steps (batch-norm parameters): every element; then T += 1
 *   okge_adagrad_lazy(OKGE_LAZY_FLUSH)  every row to T -- before evaluation, checkpoints or any other reader of the tables
 * After a flush the tables are BIT-IDENTICAL to window = 1, i.e. to okge_adagrad_multi / the reference's order of
 * operations.  lr, weight_decay and eps must not change while steps are pending (flush first).  T lives on the device so
 * that a captured HIP graph replays correctly. */
#define OKGE_LAZY_STEP 0
#define OKGE_LAZY_FLUSH 1
typedef struct okge_lazy_tensor {
    float *p, *g, *state_sum;    /* (rows, row_len) */
    int64_t rows;
    int32_t *row_steps;          /* [rows], or NULL: a plain dense tensor of rows * row_len floats, updated every step */
    uint8_t *row_touched;        /* [rows] map the pooling backward stamps (okge_pool_call.row_touched), or NULL */
    int32_t row_len, touched_stamp;
} okge_lazy_tensor;
int okge_adagrad_lazy(const okge_lazy_tensor *tensors, int32_t n_tensors, int32_t *counters, int32_t window, int32_t mode, float lr,
                      float weight_decay, float eps, void *stream);
/* calls: the batch's okge_pool_encode_calls list; tables: the lazy tensors (matched to a call by p == e->W) */
int okge_pool_catch_up_calls(const okge_pool_call *calls, int32_t n_calls, const okge_lazy_tensor *tables, int32_t n_tables,
                             const int32_t *counters, float lr, float weight_decay, float eps, void *stream);

/* ---- the whole step in one call: okge_train_forward_backward + the dense Adagrad update of both tables ------------------
 * (Trainer.compute_one_batch's training branch end to end, trainer.py:217-257 with utils/optim.py:139-160.)  Same arithmetic,
 * element for element, as okge_train_forward_backward followed by okge_adagrad_step2(zero_grad = 2 or 1) -- tables and
 * accumulators end up bit-identical -- but the update rides in the step's own launches: the sweep over the entity rows that no
 * prefix of the batch names (their gradient is final once the tile kernel has run) runs in extra workgroups of the
 * prefix-backward launch, which is latency-bound and leaves the memory system idle; one small launch finishes the <= B prefix
 * entity rows and the relation table (gradient cleared).  prefix_flags: n_ent int32 words the caller zeroes ONCE and thereafter
 * only hands to this function (it leaves them zero).  zero_entity_grad: clear dE as okge_adagrad_step does; 0 when the next step
 * overwrites every candidate row anyway (1-vs-all with OKGE_TRAIN_GRADS_ZERO). */
typedef struct okge_adagrad {
    float *sum_E, *sum_R;
    float lr, weight_decay, eps;
    int32_t zero_entity_grad;
    int32_t *prefix_flags;
} okge_adagrad;
int okge_train_step(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand, const okge_positives *pos,
                    int32_t loss_kind, float label_smoothing, double normalizer, int32_t flags, const okge_adagrad *opt,
                    double *loss_out, float *dE, float *dR, void *workspace, size_t workspace_bytes, void *stream);

/* ---- batch producer (HOST pointers; no device work) ---------------------------------------------------
 * Replaces OneToNMentionRelationDataset_collate_func (dataset.py:724-940) and the packed answer-group decoding
 * it uses (utils/misc.py:72-89).  The dataset is the reference's three int32 tensors (dataset.py:567-710):
 *   prefixes [P][7] = a, b, this_start, this_end, all_start, all_end, slot   (slot 0: (rel, obj) = po prefix,
 *                                                                             slot 2: (subj, rel) = sp prefix)
 *   seen_entities       packed answer groups of this split: for k groups [b_0+L .. b_k+L, 0, ids...], L = k+2
 *   all_splits_entities answers over train+valid+test (the evaluation filter)
 * A batch = `rows` (indices into prefixes, in sampler order).  Output, in caller-allocated host buffers:
 *   po_rel/po_obj [n_po], sp_subj/sp_rel [n_sp]   rows keep batch order within a slot, po rows come first
 *   pos_col/pos_row [nnz]   the label tensor as unique coordinates sorted by (col, row) = okge_positives
 *   cand_ids [n_cand]       batch-shared mode: answer ids in first-seen order (training: this split's answers,
 *                           evaluation: all splits'), filled up to min_size_batch_labels with entities sampled
 *                           without replacement (seen ones removed).  DIVERGENCE: the reference appends the
 *                           numpy-sampled fill-up ids in the iteration order of a Python set; here a
 *                           splitmix64(seed) stream in sampling order.  1-vs-all mode: ids offset..n_entities-1,
 *                           not written.
 *   row_ptr/grp_ptr/ids, filt_ptr/filt_col   evaluation only: label_ids and filter_mask as okge_filtered_ranks takes
 *   normalizer_loss = B*N (dataset.py:935), normalizer_metric = nnz (dataset.py:934).
 * Returns OKGE_ERR_WORKSPACE (sizes needed are then in the descriptor) if a capacity is too small. */
typedef struct okge_prefix_table {
    const int32_t *prefixes;
    int64_t n_prefixes;
    const int32_t *seen_entities;
    int64_t n_seen;
    const int32_t *all_splits_entities;
    int64_t n_all;
    int32_t n_entities;      /* entity vocabulary size incl. the reserved ids */
    int32_t entity_offset;   /* first real entity id (2) */
} okge_prefix_table;

typedef struct okge_collated {
    int64_t cap_rows, cap_pos, cap_cand, cap_groups, cap_ids, cap_filter;   /* capacities, set by the caller */
    int32_t *po_rel, *po_obj, *sp_subj, *sp_rel;                            /* [cap_rows] each */
    int32_t *pos_row, *pos_col;                                             /* [cap_pos] */
    int32_t *cand_ids;                                                      /* [cap_cand], batch-shared only */
    int64_t *row_ptr;                                                       /* [cap_rows + 1], evaluation only */
    int64_t *grp_ptr;                                                       /* [cap_groups + 1] */
    int32_t *ids;                                                           /* [cap_ids] */
    int64_t *filt_ptr;                                                      /* [cap_rows + 1] */
    int32_t *filt_col;                                                      /* [cap_filter] */
    int32_t n_po, n_sp;                                                     /* sizes, filled by the call */
    int64_t nnz, n_cand, n_groups, n_ids, n_filter;
    double normalizer_loss, normalizer_metric;
} okge_collated;

int okge_collate_batch(const okge_prefix_table *table, const int64_t *rows, int32_t B, int32_t is_training,
                       int32_t use_batch_shared_entities, int32_t min_size_batch_labels, uint64_t seed,
                       okge_collated *out);

/* n_batches batches of B prefixes each (rows = n_batches * B indices, batch k = rows[k*B .. (k+1)*B)) in one call; all
 * arrays of all batches are laid out in ONE caller-allocated int32 arena (pin it: the whole group crosses PCIe in one
 * copy) and layout[k] says where batch k's arrays start (int32 element offsets; the three int64 arrays sit on even
 * offsets) and how long they are.  arena_used returns the elements used, or -- with OKGE_ERR_WORKSPACE -- the elements
 * needed so far.  Sufficient capacity: sum over all rows of 6 * (this_end - this_start) + (all_end - all_start) + 10,
 * plus n_batches * (candidate capacity + 64).  Same per-batch semantics as okge_collate_batch (batch k uses a seed derived
 * from `seed` and k). */
typedef struct okge_arena_batch {
    int64_t off_po_rel, off_po_obj, off_sp_subj, off_sp_rel, off_pos_row, off_pos_col, off_cand;
    int64_t off_row_ptr, off_grp_ptr, off_ids, off_filt_ptr, off_filt_col;
    int64_t nnz, n_cand, n_groups, n_ids, n_filter;
    int32_t n_po, n_sp;
    double normalizer_loss, normalizer_metric;
} okge_arena_batch;

int okge_collate_batches(const okge_prefix_table *table, const int64_t *rows, int32_t n_batches, int32_t B,
                         int32_t is_training, int32_t use_batch_shared_entities, int32_t min_size_batch_labels,
                         uint64_t seed, int32_t *arena, int64_t arena_cap, okge_arena_batch *layout,
                         int64_t *arena_used);

/* ---- dataset loader (HOST; text files -> the tensors okge_collate_batch reads) --------------------------
 * Replaces OneToNMentionRelationDataset._collect_seen_triples / merge_all_splits_triples / create_data_tensors
 * (dataset.py:480-710) for the 5-column id format  s \t p \t o \t subj-mention-ids \t obj-mention-ids
 * (utils/map_dataset_to_ids.py:11-17), without the jsonl / pickle intermediates.  Split 0 = train (rows carry
 * all_start = all_end = 0; answer lists longer than max_size_prefix_label > 1 are cut into several rows),
 * 1 = valid, 2 = test.  Kept reference behaviour: string sort keys, the last prefix in sort order of every
 * (file, direction) is dropped (dataset.py:501-518 never flushes it), sp_o rows before po_s rows.
 * Divergences: ids inside one all-splits slice are ascending (reference: a Python set's iteration order);
 * no uninitialised tail rows with max_size_prefix_label (dataset.py:628-640 over-allocates). */
typedef struct okge_dataset okge_dataset;
int okge_dataset_open(const char *train_path, const char *valid_path, const char *test_path,
                      int32_t max_size_prefix_label, okge_dataset **out);
int okge_dataset_sizes(const okge_dataset *ds, int32_t split, int64_t *n_prefixes, int64_t *n_seen, int64_t *n_all,
                       int32_t *max_entity_id, int32_t *max_relation_id);
int okge_dataset_copy(const okge_dataset *ds, int32_t split, int32_t *prefixes /* [P][7] */,
                      int32_t *seen_entities, int32_t *all_splits_entities);   /* NULL pointers are skipped */
void okge_dataset_close(okge_dataset *ds);

/* ---- filtered ranks ---------------------------------------------------------------------------------
 * Replaces OneToNMentionRelationDataset.compute_metrics' rank rule (dataset.py:423-446):
 * for row b and each of its answer groups g: true = max_{j in g} scores[b][j];
 * scores'[b][j] = filter(b,j) ? -1e8 : scores[b][j];
 * ranks[g] = #(scores' > true) + (#(scores' == true)) / 2.
 * Filters are CSR (filt_ptr[B+1], filt_col) over candidate positions -- the reference's dense bool
 * filter mask (dataset.py:927) as coordinates; groups are CSR of CSR: row b owns groups
 * row_ptr[b]..row_ptr[b+1], group g owns ids[grp_ptr[g]..grp_ptr[g+1]].  ranks: int64[n_groups]. */
int okge_filtered_ranks(const float *scores, int64_t ld_scores, int32_t B, int32_t N,
                        const int64_t *filt_ptr, const int32_t *filt_col,
                        const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                        int64_t *ranks, void *stream);
/* acc[7] (DEVICE doubles) += {#groups, sum 1/(rank+1), sum rank, #rank<1, #rank<3, #rank<10, #rank<50}: the MRR / MR /
 * Hits@k meters of compute_metrics (dataset.py:447-452, utils/metrics.py) accumulated on the device. */
int okge_rank_metrics(const int64_t *ranks, int64_t n, double *acc, void *stream);
/* One evaluation batch in ONE call, pipelined over two streams: okge_score_prefixes on `stream`, then
 * okge_filtered_ranks + okge_rank_metrics on `rank_stream` (may equal `stream`).  The library orders the two with its
 * own events and makes a later call that reuses the same `scores` buffer wait until that buffer's ranks are counted, so
 * a caller alternating two score buffers gets batch i's ranking overlapped with batch i+1's scoring (scoring is
 * compute-bound, ranking a chain of memory round trips).  Replaces Trainer.evaluate's per-batch body
 * (trainer.py:258-272, 363-369). */
int okge_evaluate_batch(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                        const int64_t *filt_ptr, const int32_t *filt_col, const int64_t *row_ptr,
                        const int64_t *grp_ptr, const int32_t *ids, int64_t n_groups, float *scores, int64_t ld_scores,
                        int64_t *ranks, double *acc, void *workspace, size_t workspace_bytes, void *stream,
                        void *rank_stream);
/* One evaluation batch WITHOUT the (B, N) score block (slot sizes up to 512, eval mode: no dropout): replaces
 * Trainer.evaluate's per-batch body (trainer.py:258-272) + compute_metrics (dataset.py:423-453) by three launches on
 * `stream`: (1) the folded queries and the POINT scores the rank rule needs -- every answer group's true score and the
 * score under every filter entry -- as scalar fma chains in the tile kernel's summation order (bit-equal to the scores
 * okge_score_prefixes writes); (2) one sweep of the candidate tiles that compares the score block in registers with the
 * rows' true scores and adds {#greater, #equal} to per-group counters; (3) ranks (filter entries counted as -1e8) and
 * the seven meters added to acc[7].  ranks[] is bit-equal to okge_score_prefixes + okge_filtered_ranks.
 * n_filter = filt_ptr[B], n_groups = row_ptr[B] (host copies).  Returns OKGE_ERR_UNSUPPORTED for d > 512, dropout or a
 * candidate table: use okge_evaluate_batch there. */
size_t okge_eval_workspace_bytes(int32_t B, int32_t N, int32_t d, int64_t n_groups, int64_t n_filter);
int okge_evaluate_fused(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                        const int64_t *filt_ptr, const int32_t *filt_col, int64_t n_filter, const int64_t *row_ptr,
                        const int64_t *grp_ptr, const int32_t *ids, int64_t n_groups, int64_t *ranks, double *acc,
                        void *workspace, size_t workspace_bytes, void *stream);
/* CANDIDATE-SHARDED fused evaluation (entity table row-sharded over the GPUs of a node, SURVEY.md section 8e: exact ranks need
 * integer counts summed over the shards, not a per-shard top-k): the three launches of okge_evaluate_fused one at a time, on
 * THIS rank's candidates, with the two exchanges between them left to the caller (RCCL):
 *   phase 1  point scores.  The queries arrive FOLDED in `Q` ([B][ldq], okge_fold_queries on the exchanged entity rows).
 *            `cand` = the local candidates (rows of the local table `t`); they are the global candidate columns
 *            sh->cand_col0 .. + cand->n - 1 of n_cand_global.  ids / filt_col hold GLOBAL columns, identical on every rank.
 *            true_scores[g] (the sweep's sorted group numbering, same on every rank) = max over the group's ids THIS rank
 *            holds, -inf if none;                      -> caller: all-reduce(MAX) over the ranks
 *   phase 2  the tile sweep over the local candidates against the (global) true scores
 *   phase 4  counts[g] = {#greater, #equal} of this rank: sweep counts + the filter correction of the filter columns it
 *            holds (original group numbering);          -> caller: all-reduce(SUM); rank = #greater + #equal / 2
 * Point scores use the tile kernel's summation order, every candidate is scored on exactly one rank: the summed counts,
 * hence the ranks, are bit-equal to okge_evaluate_fused on the unsharded table (tests/test_sharded.py).
 * Replaces dataset.py:423-453 for sharded tables without the (B, N / world) score block (5.1 GB per rank and batch at the
 * 2.5 M-entity shape).  Same workspace as okge_evaluate_fused (okge_eval_workspace_bytes with the LOCAL candidate count); the
 * caller orders 1 -> 2 -> 4 of a batch.  Slot sizes up to 256, eval mode. */
int okge_evaluate_fused_shard(int32_t phase, const okge_tables *t, const okge_shard *sh, const float *Q, int64_t ldq, int32_t B,
                              const okge_candidates *cand, int32_t n_cand_global, const int64_t *filt_ptr, const int32_t *filt_col,
                              int64_t n_filter, const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                              int64_t n_groups, float *true_scores, int64_t *counts, void *workspace, size_t workspace_bytes,
                              void *stream);
/* A RUN of evaluation batches in one host call: Trainer.evaluate's loop (trainer.py:363-369) over n_batches batches.
 * Batch i runs on streams[i % n_streams] (1 to 4 different streams); each stream is an independent chain with two launches
 * per batch -- [points i] [sweep i] [ranks i + points i+S] [sweep i+S] ... (the ranks + meters of a batch and the point
 * scores of the chain's next batch are independent and share a launch) -- and NO dependency on the other chains, so the
 * device fills one chain's small latency-bound launches, and the CUs a sweep's tile grid leaves empty, with the other
 * chains' work.  Cross-stream waits happen once per CALL (the other streams join behind streams[0] at the start,
 * streams[0] waits for them at the end); a wait per batch costs ~10 us of queue latency on this hardware, more than the
 * small kernels themselves.  Batches rotate over 2 n_streams slots of `workspace` (each >= the largest batch's
 * okge_eval_workspace_bytes, rounded up to 256 bytes); batch i writes its ranks at ranks + rank_offset[i] (the exclusive
 * prefix sum of n_groups keeps them all; 2 n_streams rotating scratch regions are enough when only acc[7] is wanted).
 * On return all work has been ISSUED, streams[0] is ordered behind the other streams' share, and nothing has been
 * synchronised with the host. */
typedef struct okge_eval_batch {
    okge_prefix_batch batch;
    okge_candidates   cand;
    const int64_t    *filt_ptr;    /* [B + 1] */
    const int32_t    *filt_col;    /* [n_filter] (NULL when n_filter = 0) */
    int64_t           n_filter;
    const int64_t    *row_ptr;     /* [B + 1] */
    const int64_t    *grp_ptr;     /* [n_groups + 1] */
    const int32_t    *ids;
    int64_t           n_groups;
    int64_t           rank_offset; /* first element of this batch's ranks in ranks[] */
} okge_eval_batch;
int okge_evaluate_fused_batches(const okge_tables *t, const okge_eval_batch *batches, int32_t n_batches, int64_t *ranks,
                                double *acc, void *workspace, size_t workspace_bytes, void *const *streams, int32_t n_streams);
/* The same rank rule with the candidate columns [col0, col0 + n_local) of every row held by this rank
 * (scores: B x n_local; filter columns and group ids stay positions in the FULL candidate list):
 * true_out[g] = max over the group's ids inside the local range (-inf if none)       -> all-reduce(max)
 * counts[g]   = {#(scores' > true[g]), #(scores' == true[g])} over the local columns -> all-reduce(sum) */
int okge_group_true_scores(const float *scores, int64_t ld_scores, int32_t B, int32_t col0, int32_t n_local,
                           const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids, float *true_out,
                           void *stream);
int okge_rank_counts(const float *scores, int64_t ld_scores, int32_t B, int32_t col0, int32_t n_local,
                     const int64_t *filt_ptr, const int32_t *filt_col, const int64_t *row_ptr,
                     const float *true_scores, int64_t *counts, void *stream);

/* ---- id safety ----------------------------------------------------------------------------------------
 * Ids live in device memory; the kernels check every row index they form from one against its table, substitute row 0
 * (the padding row) for an index outside it and count the event in a device word.  okge_id_errors copies the count to
 * the host (a synchronising copy: call it where the reference would have raised -- end of an epoch, before a
 * checkpoint, in tests) and clears it.  The reference fails inside torch.nn.Embedding instead (model.py:457-460). */
int okge_id_errors(int64_t *n_out);

/* ---- gradient clipping ----------------------------------------------------------------------------------
 * torch.nn.utils.clip_grad_norm_(parameters, max_norm) over the two dense gradient tensors, as Trainer.compute_one_batch
 * applies it before optimizer.step() when args["grad_clip"] > 0 (trainer.py:236-240): both are scaled in place by
 * min(1, max_norm / (||g||_2 + 1e-6)); the norm (fp32 like torch's, stored as a double) goes to norm_out_dev if given.
 * workspace: 8448 bytes. */
int okge_clip_grad_norm(float *g0, int64_t n0, float *g1, int64_t n1, float max_norm, double *norm_out_dev, void *workspace,
                        size_t workspace_bytes, void *stream);

/* out[b] = log sum_r exp(parts[r][b]): the global log_softmax denominator of the sharded KL loss from the all-gathered
 * per-shard row log-sum-exps of okge_row_logsumexp (trainer.py:99-101). */
int okge_merge_logsumexp(const float *parts, int32_t world, int32_t B, float *out, void *stream);

/* ---- measurement ------------------------------------------------------------------------------------
 * When enabled, every kernel launch of the calls above is bracketed by HIP events on its stream.
 * okge_timing_collect synchronises the events and returns per-kernel totals since the last reset:
 * names[i] (static strings), total_ms[i], launches[i]; returns the number of kernels (<= cap). */
int okge_timing_enable(int32_t on);
int okge_timing_reset(void);
int okge_timing_collect(const char **names, double *total_ms, int64_t *launches, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* OKGE_H */
