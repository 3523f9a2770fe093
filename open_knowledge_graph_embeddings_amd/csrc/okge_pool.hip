// Token-pooled embedder: UnigramPoolingRelationEmbedder._encode (openkge/model.py:762-786)
//   id -> token ids (|vocab| x L table, right-padded with 0) -> sum / mean / max of the token embedding rows
//      -> [BatchNorm1d, training: statistics of THIS call's rows; evaluation: running statistics] -> rows
// and its backward (batch-norm backward + scatter-add into the token table's dense gradient).  Dropout, the last
// step of _encode, is applied by the consumers (the tile kernels drop rows as they gather them).
// All of it is HBM-bound gather / reduce work: one workgroup per row with 16-byte column accesses for the gathers,
// row-block partial sums + a double-precision finish for the column statistics (no float atomics on statistics).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

namespace {

constexpr int POOL_SUM = 0, POOL_MEAN = 1, POOL_MAX = 2;
constexpr int STAT_ROWS = 32;          // rows per partial-sum workgroup

__device__ __forceinline__ int row_id(const int32_t *ids, int first_id, int i, int n_ids, int *id_err)
{
    return (int)checked_row(ids ? ids[i] : first_id + i, n_ids, id_err);      // row of the token-id table
}

// out[i][k] = pool_t W[tok(i,t)][k].  Padded positions (token 0) take part: the table's row 0 is an ordinary row
// whose gradient is suppressed (padding_idx), not a zero row (model.py:660-661 re-initialises the whole weight).
__global__ __launch_bounds__(128) void pool_rows_kernel(const float *__restrict__ W, int d, const int32_t *__restrict__ tokens,
                                                        int L, const int32_t *__restrict__ ids, int first_id, int pool,
                                                        float *__restrict__ out, int64_t ld, int n_ids, int *__restrict__ id_err)
{
    const int i = blockIdx.x;
    const int32_t *tok = tokens + (size_t)row_id(ids, first_id, i, n_ids, threadIdx.x ? nullptr : id_err) * L;
    float inv = 1.f;
    if (pool == POOL_MEAN) {
        int len = 0;
        for (int t = 0; t < L; ++t) len += tok[t] > 0;
        inv = 1.f / ((float)len + 1e-12f);
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float acc = pool == POOL_MAX ? -INFINITY : 0.f;
        for (int t = 0; t < L; ++t) {
            const float w = W[(size_t)tok[t] * d + k];
            acc = pool == POOL_MAX ? fmaxf(acc, w) : acc + w;
        }
        out[(size_t)i * ld + k] = pool == POOL_MEAN ? acc * inv : acc;       // torch divides: sum / (len + 1e-12)
    }
}

// MODE 0: partial[b][0][k] = mean of the block's rows, partial[b][1][k] = their sum of squared deviations from it
//         (two sweeps over 64 rows that stay in cache; merged without cancellation by col_finish_kernel<0>)
// MODE 2: partial[b][0][k] = sum_i dy,  partial[b][1][k] = sum_i dy * xhat      (xhat = (x - mean) * rstd)
template <int MODE>
__global__ __launch_bounds__(256) void col_partial_kernel(const float *__restrict__ X, int64_t ldx, const float *__restrict__ DY,
                                                          int64_t lddy, int n, int d, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, float *__restrict__ partial)
{
    const int r0 = blockIdx.x * STAT_ROWS, r1 = min(n, r0 + STAT_ROWS);
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float a = 0.f, b = 0.f;
        if (MODE == 0) {
            for (int i = r0; i < r1; ++i) a += X[(size_t)i * ldx + k];
            a /= (float)(r1 - r0);
            for (int i = r0; i < r1; ++i) {
                const float dx = X[(size_t)i * ldx + k] - a;
                b += dx * dx;
            }
        } else {
            const float m = mean[k], rs = rstd[k];
            for (int i = r0; i < r1; ++i) {
                const float dy = DY[(size_t)i * lddy + k];
                a += dy;
                b += dy * ((X[(size_t)i * ldx + k] - m) * rs);
            }
        }
        partial[((size_t)blockIdx.x * 2 + 0) * d + k] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * d + k] = b;
    }
}

// finish in double: STEP 0 -> mean and rstd of all n rows from the per-block (mean, M2) pairs (+ running statistics);
// STEP 2 -> this call's (dbias, dweight) into saved[2..3] and accumulated into the parameter gradients
template <int STEP>
__global__ __launch_bounds__(256) void col_finish_kernel(const float *__restrict__ partial, int blocks, int n, int d, float eps,
                                                         float momentum, float *__restrict__ saved, float *__restrict__ run_mean,
                                                         float *__restrict__ run_var, float *__restrict__ d_weight,
                                                         float *__restrict__ d_bias)
{
    // 64 columns per workgroup, 4 threads per column each folding a quarter of the row-block partials
    __shared__ double red[2][4][64];
    const int c = threadIdx.x & 63, grp = threadIdx.x >> 6, k = blockIdx.x * 64 + c;
    const bool live = k < d;
    double a = 0.0, b = 0.0;
    if (STEP == 0) {
        // parallel-variance merge: mean = sum n_b mean_b / n,  M2 = sum [M2_b + n_b (mean_b - mean)^2]
        if (live)
            for (int p = grp; p < blocks; p += 4)
                a += (double)partial[((size_t)p * 2 + 0) * d + k] * (double)min(STAT_ROWS, n - p * STAT_ROWS);
        red[0][grp][c] = a;
        __syncthreads();
        const double mean = (red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]) / n;
        if (live)
            for (int p = grp; p < blocks; p += 4) {
                const double dm = (double)partial[((size_t)p * 2 + 0) * d + k] - mean;
                b += (double)partial[((size_t)p * 2 + 1) * d + k] + dm * dm * (double)min(STAT_ROWS, n - p * STAT_ROWS);
            }
        red[1][grp][c] = b;
        __syncthreads();
        if (!live || grp != 0) return;
        const double var = (red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]) / n;   // biased: normalises the batch
        saved[k] = (float)mean;
        saved[d + k] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {                                             // running statistics use the unbiased variance
            run_mean[k] += momentum * ((float)mean - run_mean[k]);
            run_var[k] += momentum * ((float)(var * n / (n > 1 ? n - 1 : 1)) - run_var[k]);
        }
    } else {
        if (live)
            for (int p = grp; p < blocks; p += 4) {
                a += partial[((size_t)p * 2 + 0) * d + k];
                b += partial[((size_t)p * 2 + 1) * d + k];
            }
        red[0][grp][c] = a;
        red[1][grp][c] = b;
        __syncthreads();
        if (!live || grp != 0) return;
        a = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
        b = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
        saved[2 * d + k] = (float)a;
        saved[3 * d + k] = (float)b;
        d_bias[k] += (float)a;
        d_weight[k] += (float)b;
    }
}

// y = (x - mean) * rstd * weight + bias     (training: saved = this call's statistics; evaluation: running ones)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float *__restrict__ X, int64_t ldx, int n, int d,
                                                       const float *__restrict__ mean, const float *__restrict__ rstd_or_var,
                                                       int is_var, float eps, const float *__restrict__ weight,
                                                       const float *__restrict__ bias, float *__restrict__ Y, int64_t ldy)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)n * d) return;
    const int i = (int)(idx / d), k = (int)(idx % d);
    const float rs = is_var ? 1.0f / sqrtf(rstd_or_var[k] + eps) : rstd_or_var[k];
    Y[(size_t)i * ldy + k] = (X[(size_t)i * ldx + k] - mean[k]) * rs * weight[k] + bias[k];
}

// dx = weight * rstd * (dy - dbias/n - xhat * dweight/n)   (or dx = dy without batch-norm), then scattered into the
// token table gradient: sum -> every token of the row, mean -> scaled by 1/(len + 1e-12), max -> the first token that
// attains the maximum of its column.  Token 0 (padding_idx) receives nothing.
// One workgroup takes POOL_BWD_ROWS rows, one thread per column.  The reference's token ids are frequency-ranked
// (index_mapper.py:95-108 writes the maps sorted by count; BOS = 2 and EOS = 3 sit in every row), so the first
// HOT_TOKENS ids would take thousands of same-address atomics per step: their rows are accumulated in LDS (a thread
// owns its column, so plain read-modify-write) and flushed with one atomic per touched (token, column) per workgroup.
constexpr int POOL_BWD_ROWS = 16, HOT_TOKENS = 32, POOL_MAX_LEN = 64;

__global__ __launch_bounds__(256) void pool_backward_kernel(const float *__restrict__ W, int d, const int32_t *__restrict__ tokens,
                                                            int L, const int32_t *__restrict__ ids, int first_id, int pool,
                                                            const float *__restrict__ X, int64_t ldx,
                                                            const float *__restrict__ DY, int64_t lddy, int n,
                                                            const float *__restrict__ saved, const float *__restrict__ weight,
                                                            float *__restrict__ dW, int n_ids)
{
    extern __shared__ float hot[];                      // [HOT_TOKENS][d]
    __shared__ uint32_t hot_seen;
    __shared__ int32_t toks[POOL_BWD_ROWS][POOL_MAX_LEN];
    __shared__ float inv_len[POOL_BWD_ROWS];
    const int r0 = blockIdx.x * POOL_BWD_ROWS, nr = min(n, r0 + POOL_BWD_ROWS) - r0;
    for (int i = threadIdx.x; i < HOT_TOKENS * d; i += blockDim.x) hot[i] = 0.f;
    for (int i = threadIdx.x; i < nr * L; i += blockDim.x)
        toks[i / L][i % L] = tokens[(size_t)row_id(ids, first_id, r0 + i / L, n_ids, nullptr) * L + i % L];
    if (threadIdx.x == 0) hot_seen = 0;
    __syncthreads();
    if (threadIdx.x < nr) {
        int len = 0;
        for (int t = 0; t < L; ++t) len += toks[threadIdx.x][t] > 0;
        inv_len[threadIdx.x] = pool == POOL_MEAN ? 1.f / ((float)len + 1e-12f) : 1.f;
    }
    __syncthreads();
    const float inv_n = 1.f / (float)n;
    uint32_t seen = 0;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        // all of this column's gradients first (independent loads in flight), then the scatter
        float g[POOL_BWD_ROWS];
#pragma unroll
        for (int r = 0; r < POOL_BWD_ROWS; ++r) g[r] = r < nr ? DY[(size_t)(r0 + r) * lddy + k] : 0.f;
        if (saved) {
            const float m = saved[k], rs = saved[d + k], db = saved[2 * d + k], dw = saved[3 * d + k], wk = weight[k];
#pragma unroll
            for (int r = 0; r < POOL_BWD_ROWS; ++r) {
                const float x = r < nr ? X[(size_t)(r0 + r) * ldx + k] : 0.f;
                g[r] = wk * rs * (g[r] - db * inv_n - (x - m) * rs * dw * inv_n);
            }
        }
#pragma unroll
        for (int r = 0; r < POOL_BWD_ROWS; ++r) {
            if (r >= nr) continue;
            if (pool == POOL_MAX) {
                int best = 0;
                float bw = -INFINITY;
                for (int t = 0; t < L; ++t) {
                    const float w = W[(size_t)toks[r][t] * d + k];
                    if (w > bw) { bw = w; best = t; }
                }
                const int tk = toks[r][best];
                if (tk == 0) continue;
                if (tk < HOT_TOKENS) { hot[tk * d + k] += g[r]; seen |= 1u << tk; }
                else atomicAdd(dW + (size_t)tk * d + k, g[r]);
            } else {
                const float gg = g[r] * inv_len[r];
                for (int t = 0; t < L; ++t) {
                    const int tk = toks[r][t];
                    if (tk == 0) continue;
                    if (tk < HOT_TOKENS) { hot[tk * d + k] += gg; seen |= 1u << tk; }
                    else atomicAdd(dW + (size_t)tk * d + k, gg);
                }
            }
        }
    }
    if (seen) atomicOr(&hot_seen, seen);
    __syncthreads();
    const uint32_t all = hot_seen;
    for (int tk = 1; tk < HOT_TOKENS; ++tk) {
        if (!(all >> tk & 1u)) continue;
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            const float v = hot[tk * d + k];
            if (v != 0.f) atomicAdd(dW + (size_t)tk * d + k, v);
        }
    }
}

}  // namespace

size_t pool_workspace_bytes(int n, int d)
{
    const size_t blocks = (size_t)(n + STAT_ROWS - 1) / STAT_ROWS;
    return blocks * 2 * (size_t)d * sizeof(float);
}

hipError_t launch_pool_rows(const float *W, int d, const int32_t *tokens, int L, const int32_t *ids, int first_id, int n,
                            int pool, float *out, int64_t ld, int n_ids, int *id_err, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pool_rows_kernel, dim3(n), dim3(128), 0, st, W, d, tokens, L, ids, first_id, pool, out, ld, n_ids, id_err);
    return hipGetLastError();
}

// training-mode statistics of the n rows of X: saved[0..d) = mean, saved[d..2d) = rstd; running stats updated
hipError_t launch_bn_stats(const float *X, int64_t ldx, int n, int d, float eps, float momentum, float *saved,
                           float *run_mean, float *run_var, float *partial, hipStream_t st)
{
    const int blocks = (n + STAT_ROWS - 1) / STAT_ROWS, fb = (d + 63) / 64;
    hipLaunchKernelGGL(col_partial_kernel<0>, dim3(blocks), dim3(256), 0, st, X, ldx, nullptr, 0, n, d, nullptr, nullptr, partial);
    hipLaunchKernelGGL(col_finish_kernel<0>, dim3(fb), dim3(256), 0, st, partial, blocks, n, d, eps, momentum, saved, run_mean,
                       run_var, nullptr, nullptr);
    return hipGetLastError();
}

hipError_t launch_bn_apply(const float *X, int64_t ldx, int n, int d, const float *mean, const float *rstd_or_var, int is_var,
                           float eps, const float *weight, const float *bias, float *Y, int64_t ldy, hipStream_t st)
{
    const int64_t total = (int64_t)n * d;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, ldx, n, d, mean, rstd_or_var,
                       is_var, eps, weight, bias, Y, ldy);
    return hipGetLastError();
}

hipError_t launch_pool_backward(const float *W, int d, const int32_t *tokens, int L, const int32_t *ids, int first_id, int n,
                                int pool, const float *X, int64_t ldx, const float *DY, int64_t lddy, float *saved,
                                const float *weight, float *d_weight, float *d_bias, float *dW, float *partial, int n_ids,
                                hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    if (saved) {
        const int blocks = (n + STAT_ROWS - 1) / STAT_ROWS, fb = (d + 63) / 64;
        hipLaunchKernelGGL(col_partial_kernel<2>, dim3(blocks), dim3(256), 0, st, X, ldx, DY, lddy, n, d, saved, saved + d, partial);
        hipLaunchKernelGGL(col_finish_kernel<2>, dim3(fb), dim3(256), 0, st, partial, blocks, n, d, 0.f, 0.f, saved, nullptr,
                           nullptr, d_weight, d_bias);
    }
    hipLaunchKernelGGL(pool_backward_kernel, dim3((n + POOL_BWD_ROWS - 1) / POOL_BWD_ROWS), dim3(256),
                       sizeof(float) * HOT_TOKENS * d, st, W, d, tokens, L, ids, first_id, pool, X, ldx, DY, lddy, n, saved,
                       weight, dW, n_ids);
    return hipGetLastError();
}

}  // namespace okge
