// Kernel argument blocks and launcher prototypes shared by the .hip translation units.
#pragma once
#include <string>
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include "okge_device.h"

namespace okge {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE setting: every launcher keeps one of these as a
// function-local static and opts the kernel in once per device (and again if a larger size is asked for), keyed by
// hipGetDevice() -- one process per GPU is the normal deployment, but a process driving two devices must work too.
struct LdsOptIn {
    std::atomic<size_t> bytes[64];
    LdsOptIn() { for (auto &b : bytes) b.store(0, std::memory_order_relaxed); }
};
inline hipError_t ensure_dynamic_lds(LdsOptIn &state, const void *kernel, size_t shmem)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (shmem <= state.bytes[dev].load(std::memory_order_acquire)) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);      // (idempotent: a race repeats it)
    if (e == hipSuccess) state.bytes[dev].store(shmem, std::memory_order_release);
    return e;
}

constexpr int NT = 64;             // candidate rows per tile
constexpr int BC = 64;             // batch rows per chunk
constexpr int LDG = 68;            // leading dimension of the 64x64 G / X tile in LDS (4*odd)
constexpr int FUSED_THREADS = 512; // 8 waves, two per SIMD (fused_tile_kernel: score / stats / count sweep)
constexpr int POS_CACHE = 512;     // positives of one candidate tile cached in LDS (more spill to global reads)

enum { MODE_TRAIN_BCE = 0, MODE_SCORE = 1, MODE_STATS = 2, MODE_TRAIN_KL = 3, MODE_COUNT = 4 };
enum { LOSS_BCE = 0, LOSS_KL = 1 };
enum { SC_COMPLEX = 0, SC_DISTMULT = 1 };

struct FusedArgs {
    const float   *E;          // entity table (n_ent, d)
    const int32_t *cand_ids;   // nullptr => cand_first + position
    const float   *Q;          // folded queries [Bpad][ldq], zero padded to ldq >= D16
    const int32_t *pos_col, *pos_row;
    const int32_t *tile_ptr;   // [tiles + 1] offsets into pos_* per candidate tile
    const float   *row_lse, *row_ysum;   // KL
    float         *G;          // [Bpad][ldg]  dLoss/dX / normalizer, row-major, ldg = 64 * tiles
    float         *Cm;         // [64 * tiles][16*KB]  masked (dropped-out) candidate rows, for dq_kernel
    float         *dE;
    float         *dC_slab;    // [gridDim.y][32 * ktiles][16*KB] partial candidate gradients when the batch is split over blockIdx.y
    double        *loss_partial;
    float         *X;          // score mode
    float         *stats;      // stats mode: float2 [tiles][Bpad]
    // count mode (fused evaluation): per answer group of every row, how many of the tile's scores are greater than /
    // equal to the group's true score -> atomically added to rk_counts[group][2]
    const int64_t *rk_row_ptr; // [B + 1] groups of each row
    const float   *rk_true;    // [n_groups]
    int32_t       *rk_counts;  // [n_groups][2]   (atomics; used when the slab would be too large)
    uint32_t      *rk_slab;    // [tiles][n_groups] packed (#greater | #equal << 16) of every tile: plain coalesced stores
    int64_t        rk_ngroups;
    unsigned long long *stamps_dbg;   // diagnostic build (-DOKGE_STAMPS) only
    int64_t        ldx;
    DropDev        drop_c;
    int32_t        d, KB, LDK, cand_first, N, B, Bpad, ldq, ldg, nnz, b_per_block, loss_kind, x_vec_ok, grads_zero, cand_exclusive;
    float          y_pos, y_neg, inv_norm;
    int32_t        cand_col0;  // global column (candidate position) of local candidate 0: positives, dropout keys
    int64_t        n_table_rows;   // rows of the table `E` points to: candidate ids are checked against it (checked_row)
    int           *id_err;         // device word counting out-of-range ids
    int32_t        loss_only;  // forward + loss only: no G store, no dC product, no write-back
    int32_t        sk_tiles;   // fused_tile64k_kernel: > 0 = stream-K launch over this many candidate tiles (grid = workgroups)
};

struct DqArgs {
    const float   *G;
    const float   *Cm;         // masked candidate rows written by the tile kernel
    float         *slab;       // [nsplit][Bpad][ldq]
    int32_t        d, KB, LDK, N, Bpad, ldq, ldg, nsplit;
    int32_t        accumulate; // add to the slabs instead of overwriting them (candidate ranges after the first)
    int32_t        waves8;     // dq8_kernel: one 8-wave workgroup per CU, contraction split over two wave groups (d <= 256)
};

struct PrefixDev {
    const int32_t *po_rel, *po_obj, *sp_subj, *sp_rel;
    int32_t        n_po, n_sp;
    int32_t        ent_lo, ent_hi;   // global entity ids [ent_lo, ent_hi) live in the local table (row = id - ent_lo)
    int32_t        n_rel;            // rows of the relation table; whole_table: ent range IS the table (an id outside is an error,
    int32_t        whole_table;      // not another rank's row)
    int           *id_err;           // device word counting out-of-range ids
    DropDev        drop_po_ent, drop_po_rel, drop_sp_ent, drop_sp_rel;
};

size_t     fused_shmem_bytes(int LDK);
size_t     dq_shmem_bytes(int LDK);
hipError_t launch_fused(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st);
hipError_t launch_dq(const DqArgs &a, int grid_x, hipStream_t st);
hipError_t launch_fused32(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st);
hipError_t launch_fused64(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st);
hipError_t launch_fused64k(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st);   // slot sizes above 256

// up to four float regions cleared by extra workgroups of encode_queries_kernel (OKGE_TRAIN_CLEAR_GRADS: dR, dE in front of
// and behind the candidates; the KL loss' per-row label mass before it is counted)
constexpr int CLEAR_REGIONS = 4;
struct ClearSpec { float *p[CLEAR_REGIONS]; int64_t n[CLEAR_REGIONS]; int32_t *prefix_flags; };   // prefix_flags: [n_ent] words, set to 1 for every prefix entity row (okge_train_step)
hipError_t launch_encode_queries(const float *E, const float *R, int d, int scorer, const PrefixDev &p, float *Q,
                                 int ldq, int Bpad, float *ent_rows, const int32_t *pos_col, int nnz, int32_t *tile_ptr,
                                 int tiles, int tile_w, int cand_col0, hipStream_t st, const ClearSpec *clear = nullptr);
hipError_t launch_fold_queries(const float *R, int d, int scorer, const PrefixDev &p, const float *ent_rows, float *Q, int ldq,
                               int Bpad, hipStream_t st);
hipError_t launch_slab_reduce(const float *slab, int nsplit, int64_t n, float *out, const double *loss_partials,
                              int n_partials, double *loss_out, hipStream_t st);
// okge_train_step: the dense Adagrad update folded into the step's last launches.  The sweep over the entity rows NO prefix of
// the batch names (their gradient is final once the tile kernel has run) rides in extra workgroups of the prefix-backward launch --
// a latency-bound launch of B one-row workgroups that leaves the memory system idle --, and a small launch finishes the <= B
// flagged rows (they receive prefix gradients in that launch) and the relation table.
struct AdagradFuse {
    float   *E, *sumE, *dE, *R, *sumR, *dR;
    int32_t *flags;                    // [n_ent]: 1 = a prefix entity of this batch (set by the encode launch, cleared by the finish)
    int64_t  n_ent;
    int32_t  n_rel, d, zero_dE, on;
    float    lr, wd, eps;
};
hipError_t launch_prefix_backward(const float *E, const float *R, int d, int scorer, const PrefixDev &p,
                                  const float *slab, int nsplit, int Bpad, int ldq, const float *ent_rows, float *dE,
                                  float *dR, const double *loss_partials, int n_partials, double *loss_out,
                                  hipStream_t st, const int32_t *rel_order = nullptr, const int32_t *rel_seg_ptr = nullptr,
                                  int n_rel_seg = 0, const int32_t *ent_order = nullptr, const int32_t *ent_seg_ptr = nullptr,
                                  int n_ent_seg = 0, float *grad_rows = nullptr, int distinct = 0, const AdagradFuse *fuse = nullptr);
hipError_t launch_adagrad_finish(const AdagradFuse &af, const PrefixDev &p, hipStream_t st);
hipError_t launch_loss_reduce(const double *partials, int n, double *loss_out, hipStream_t st);
hipError_t launch_kl_count_pos(const int32_t *pos_row, int nnz, int Bpad, float *row_ysum, hipStream_t st);
hipError_t launch_kl_row_lse(const float *stats, int tiles, int B, int Bpad, float *run, int first, int last, float *row_lse,
                             hipStream_t st, const int32_t *pos_row = nullptr, int nnz = 0, float *row_ysum = nullptr);
hipError_t launch_encode_rows(const float *table, int64_t table_rows, int d, const int32_t *ids, int first_id, int n,
                              const DropDev &drop, float *out, int64_t ld_out, int *id_err, hipStream_t st);
hipError_t launch_scale(float *x, int64_t n, const float *alpha_dev, hipStream_t st);
hipError_t launch_rescale2(float *x0, int64_t n0, float *x1, int64_t n1, const float *alpha_dev, float applied, hipStream_t st);
hipError_t launch_adagrad(float *p, float *g, float *sum, int64_t n, float lr, float wd, float eps, int zero_grad,
                          hipStream_t st);
hipError_t launch_adagrad2(float *p0, float *g0, float *s0, int64_t n0, float *p1, float *g1, float *s1, int64_t n1,
                           float lr, float wd, float eps, int zero_grad, hipStream_t st);
// torch.nn.utils.clip_grad_norm_ over two gradient tensors: coef = min(1, max_norm / (||g|| + 1e-6)) -> coef_dev (fp32),
// ||g|| -> norm_dev (double); the caller scales with launch_scale (trainer.py:236-240)
hipError_t launch_clip_coef(const float *g0, int64_t n0, const float *g1, int64_t n1, float max_norm, double *partial,
                            int n_partial, float *coef_dev, double *norm_dev, hipStream_t st);
// log-sum-exp over `world` per-shard row log-sum-exps: out[b] = log sum_r exp(parts[r][b])  (sharded KL loss)
hipError_t launch_merge_lse(const float *parts, int world, int B, float *out, hipStream_t st);
// backward of the prefix scores from a dense (b, n) gradient block, and the scatter of encoded-row gradients (okge_gemm.hip)
size_t score_backward_workspace_bytes(int b, int n, int d);
hipError_t launch_score_backward(int scorer, int sp, const float *G, int64_t ld_g, int b, int n, const float *ent, int64_t ld_ent,
                                 const float *rel, int64_t ld_rel, const float *cand, int64_t ld_cand, int d, float *d_ent,
                                 float *d_rel, float *d_cand, void *workspace, hipStream_t st);
hipError_t launch_scatter_rows(const float *rows, int64_t ld, const int32_t *ids, const int32_t *order, int first_id, int n, int d,
                               const DropDev &drop, float *table, int64_t table_rows, int *id_err, hipStream_t st);
// error text for okge_last_error(), shared by the translation units of the C ABI (defined in okge_api.hip)
int report_error(int code, const std::string &msg);

// token-pooled embedder (okge_pool.hip): one _encode call of a batch (UnigramPoolingRelationEmbedder._encode, model.py:762-786)
constexpr int POOL_MAX_CALLS = 8;
struct PoolCall {
    const float   *W;                       // token table (vocab x d)
    const int32_t *tokens;                  // (n_ids x L) token ids
    const int32_t *ids;                     // rows of the call (or first_id + i)
    float         *raw, *out;               // [n][ld] pooled rows / normalised rows (out == raw without batch-norm)
    float         *saved;                   // [4 d] {mean, rstd, scratch, scratch} of this call; nullptr: no training-mode batch-norm
    const float   *dY;                      // backward: [n][ld] gradient of `out`
    const float   *bn_weight, *bn_bias;     // nullptr: no batch-norm
    float         *run_mean, *run_var;
    float         *d_weight, *d_bias, *dW;  // backward targets
    float         *partial;                 // [ceil(n / 32)][2][d] scratch of this call
    uint8_t       *touched;                 // backward: [vocab] bytes, touched[token] = touched_stamp for every row of dW written (or nullptr)
    float         *sumW;                    // catch-up (okge_pool_catch_up_calls): Adagrad accumulator of W and
    int32_t       *steps;                   //   [vocab] optimizer steps each row of W has seen
    int64_t        ld;
    int32_t        d, L, first_id, n, pool, n_ids, vocab, touched_stamp;
    float          eps, momentum;
};
size_t pool_workspace_bytes(int n, int d);
hipError_t launch_pool_encode_calls(const PoolCall *calls, int n_calls, int training, int *id_err, hipStream_t st);
// rows of W the calls' tokens name: their pending decay-only Adagrad steps (okge_adagrad_lazy), before the forward reads them
hipError_t launch_pool_catch_up(const PoolCall *calls, int n_calls, const int32_t *counters, float lr, float wd, float eps, int *id_err,
                                hipStream_t st);
// state == nullptr: the token-table gradient is scattered with float atomics; otherwise (pool sum / mean, slot sizes that are
// a multiple of 4): store-and-sum through a device-built inverted index, bit-reproducible (okge_pool.hip, "scatter plan")
size_t pool_scatter_state_bytes(const PoolCall *calls, int n_calls);
size_t pool_scatter_workspace_bytes(const PoolCall *calls, int n_calls);
hipError_t launch_pool_backward_calls(const PoolCall *calls, int n_calls, int *id_err, hipStream_t st, void *state = nullptr,
                                      size_t state_bytes = 0, void *scratch = nullptr, size_t scratch_bytes = 0);
// dense Adagrad over up to four tensors in one launch; a tensor may come with a touched-row byte map (rows whose byte differs
// from `stamp` hold an all-zero gradient by contract: it is neither read nor cleared)
constexpr int ADAGRAD_MAX_SEGS = 4;
struct AdagradSegM { float *p, *g, *s; int64_t n; const uint8_t *touched; int32_t row_len, stamp, zero_grad, rows; };   // rows: 0 all, 1 unstamped only, 2 stamped only
hipError_t launch_adagrad_multi(const AdagradSegM *segs, int n_segs, float lr, float wd, float eps, hipStream_t st);
// deferred weight-decay-only updates (okge_adagrad_lazy; okge_misc.hip): steps == nullptr: a plain dense tensor of rows * row_len floats
constexpr int LAZY_STEP = 0, LAZY_FLUSH = 1;
struct LazySeg { float *p, *g, *s; int32_t *steps; uint8_t *touched; int64_t rows; int32_t row_len, stamp; };
hipError_t launch_adagrad_lazy(const LazySeg *segs, int n_segs, int32_t *counters, int window, int mode, float lr, float wd,
                               float eps, hipStream_t st);
hipError_t launch_dc_reduce(const float *slab, int nsplit, int rows_pad, int D16, int N, int d, const int32_t *cand_ids,
                            int cand_first, int exclusive, int grads_zero, float *dE, int64_t table_rows, int *id_err,
                            hipStream_t st);
// sums the partial candidate-gradient slabs of a stream-K launch of fused_tile64k_kernel (slab 2p / 2p+1 = first / last
// segment of workgroup p; tiles covered by one whole segment were stored by the tile kernel itself)
hipError_t launch_dc_reduce_streamk(const float *slab, int tiles, int chunks_per_tile, int workgroups, int D16, int N, int d,
                                    const int32_t *cand_ids, int cand_first, int exclusive, int grads_zero, float *dE,
                                    int64_t table_rows, int *id_err, hipStream_t st);
hipError_t launch_score_triples(const float *S, int64_t lds_, const float *Rr, int64_t ldr, const float *O, int64_t ldo,
                                int n, int d, int scorer, float *out, hipStream_t st);
hipError_t launch_rank_metrics(const int64_t *ranks, int64_t n, double *acc, hipStream_t st);
// fused evaluation (okge_evaluate_fused): point scores in the tile kernel's summation order, then ranks from the counts
// fused evaluation: the two small kernels around the tile sweep (okge_misc.hip)
struct EvalPointsArgs {
    const float   *E, *R;
    PrefixDev      p;
    float         *Q;                      // [Bpad][ldq] folded queries at their SORTED positions
    const int32_t *cand_ids;
    const int64_t *row_ptr, *grp_ptr, *filt_ptr;
    const int32_t *ids, *filt_col;
    float         *true_out, *filt_x;      // [n_groups] (sorted numbering), [n_filter]
    int64_t       *row_ptr_sorted, *gshift;
    int32_t       *group_row;              // [n_groups] row of every group (original numbering), for eval_ranks
    int64_t        table_rows;
    int32_t        d, scorer, ldq, KB, Bpad, cand_first, n_cand;
    // candidate-sharded evaluation (okge_evaluate_fused_shard): the queries arrive folded (original row order), the local
    // candidates are the global columns col_lo .. col_lo + n_cand - 1 of n_cand_global; ids / filter columns stay global
    const float   *Q_in;                   // [B][ldq] or nullptr (fold from E / R: the single-device path)
    int32_t        col_lo, n_cand_global;
};
struct EvalRanksArgs {
    const int32_t  *counts;                // [n_groups][2] (atomics path) or nullptr
    const uint32_t *slab;                  // [tiles][n_groups] packed counts or nullptr
    const float    *true_scores, *filt_x;
    const int64_t  *filt_ptr, *gshift;
    const int32_t  *group_row;
    int64_t        *ranks;
    double         *acc;
    int64_t        *counts_out;            // sharded: [n_groups][2] {#greater, #equal} of THIS shard instead of ranks + meters
    int64_t         n_groups;
    int32_t         tiles, B;
};
// either part may be absent (nullptr): points of one batch and ranks of ANOTHER share a launch in a run of batches
hipError_t launch_eval_side(const EvalPointsArgs *pts, const EvalRanksArgs *rk, hipStream_t st);
hipError_t launch_ranks(const float *scores, int64_t ld, int B, int N, const int64_t *filt_ptr,
                        const int32_t *filt_col, const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                        int64_t *ranks, int col0, const float *true_in, float *true_out, int64_t *counts_out,
                        hipStream_t st);

}  // namespace okge
