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
constexpr int STAT_ROWS = 32;          // rows per pooling / partial-sum workgroup
constexpr int POOL_BWD_ROWS = 16, HOT_TOKENS = 32, POOL_MAX_LEN = 64;
constexpr int BN_ROWS = 16;            // rows per workgroup of bn_apply_kernel

// A step of the token-pooled models makes FIVE _encode calls (candidates, po relations, po objects, sp subjects, sp
// relations: trainer.py:75-91) and as many backward passes.  Run one after the other they were 35 small launches per step,
// each a chain of dependent loads on a few hundred workgroups (0.45 ms of a 1.2 ms step at BASELINE configs[4]).  Every
// kernel here therefore takes a BATCH of calls: a workgroup finds its call from the cumulative block counts, so one step is
// three launches forward (pool + partial statistics, finish, apply) and three backward, each filling the chip.
struct PoolBatch {
    PoolCall c[POOL_MAX_CALLS];
    int32_t  n_calls;
    int32_t  cum[POOL_MAX_CALLS + 1];      // cumulative workgroup counts of the kernel being launched
    int     *id_err;
};

__device__ __forceinline__ int locate_call(const PoolBatch &pb, int blk, int &local)
{
    int c = 0;
#pragma unroll
    for (int i = 1; i < POOL_MAX_CALLS; ++i)
        if (i < pb.n_calls && blk >= pb.cum[i]) c = i;
    local = blk - pb.cum[c];
    return c;
}

__device__ __forceinline__ int row_id(const int32_t *ids, int first_id, int i, int n_ids, int *id_err)
{
    return (int)checked_row(ids ? ids[i] : first_id + i, n_ids, id_err);      // row of the token-id table
}

// raw[i][k] = pool_t W[tok(i,t)][k] for the 32 rows of the workgroup, and -- training-mode batch-norm -- the block's partial
// column statistics: partial[b][0][k] = mean of the block's rows, partial[b][1][k] = their sum of squared deviations from it
// (merged without cancellation by bn_finish_kernel<0>).  Thread = (column quad cq = tid & 63, row group rg = tid >> 6):
// 16-byte loads of four columns for the group's 8 rows -- 80 gathers per thread at 10 tokens instead of 320 four-byte
// ones with a thread per column (the launch is a chain of gathers on two waves per SIMD: 95 -> 50 us at configs[4]) --,
// the four groups' (mean, M2) merged exactly (Chan) through LDS.  Padded positions (token 0) take part: the table's row 0 is
// an ordinary row whose gradient is suppressed (padding_idx), not a zero row (model.py:660-661 re-initialises the whole
// weight).  Slot sizes that are not a multiple of 4 take the scalar path (thread = column).
__global__ __launch_bounds__(256) void pool_stats_kernel(const PoolBatch pb, int training)
{
    __shared__ int32_t toks[STAT_ROWS][POOL_MAX_LEN];
    __shared__ float inv_len[STAT_ROWS];
    extern __shared__ float grp_stats[];                // [2][4 row groups][d]: per-group mean, M2
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    const int L = q.L, d = q.d, r0 = lb * STAT_ROWS, nr = min(q.n, r0 + STAT_ROWS) - r0;
    for (int i = threadIdx.x; i < nr * L; i += blockDim.x)
        toks[i / L][i % L] = q.tokens[(size_t)row_id(q.ids, q.first_id, r0 + i / L, q.n_ids, (i % L) ? nullptr : pb.id_err) * L + i % L];
    __syncthreads();
    if (threadIdx.x < nr) {
        int len = 0;
        for (int t = 0; t < L; ++t) len += toks[threadIdx.x][t] > 0;
        inv_len[threadIdx.x] = q.pool == POOL_MEAN ? 1.f / ((float)len + 1e-12f) : 1.f;     // torch divides: sum / (len + 1e-12)
    }
    __syncthreads();
    const bool stats = training && q.saved;
    const bool copy = !q.bn_weight && q.out != q.raw;
    const bool vec = (d & 3) == 0 && (q.ld & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(q.W) | reinterpret_cast<uintptr_t>(q.raw) | reinterpret_cast<uintptr_t>(q.out)) & 15) == 0;
    if (vec) {
        constexpr int RG = 8;                            // rows per row group
        const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, rbase = RG * rg;
        const int cnt = max(0, min(nr - rbase, RG));     // this group's live rows
        for (int k = 4 * cq; k < d; k += 256) {
            float4 v[RG];
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < cnt) {
                    const float ini = q.pool == POOL_MAX ? -INFINITY : 0.f;
                    float4 acc = make_float4(ini, ini, ini, ini);
                    for (int t = 0; t < L; ++t) {
                        const float4 w = *reinterpret_cast<const float4 *>(q.W + (size_t)toks[rbase + r][t] * d + k);
                        if (q.pool == POOL_MAX) { acc.x = fmaxf(acc.x, w.x); acc.y = fmaxf(acc.y, w.y); acc.z = fmaxf(acc.z, w.z); acc.w = fmaxf(acc.w, w.w); }
                        else { acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w; }
                    }
                    if (q.pool == POOL_MEAN) { const float il = inv_len[rbase + r]; acc.x *= il; acc.y *= il; acc.z *= il; acc.w *= il; }
                    v[r] = acc;
                    *reinterpret_cast<float4 *>(q.raw + (size_t)(r0 + rbase + r) * q.ld + k) = acc;
                    if (copy) *reinterpret_cast<float4 *>(q.out + (size_t)(r0 + rbase + r) * q.ld + k) = acc;
                    a.x += acc.x; a.y += acc.y; a.z += acc.z; a.w += acc.w;
                }
            }
            if (stats) {
                const float ic = cnt ? 1.f / (float)cnt : 0.f;
                a.x *= ic; a.y *= ic; a.z *= ic; a.w *= ic;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < RG; ++r)
                    if (r < cnt) {
                        const float dx = v[r].x - a.x, dy = v[r].y - a.y, dz = v[r].z - a.z, dw = v[r].w - a.w;
                        b.x += dx * dx; b.y += dy * dy; b.z += dz * dz; b.w += dw * dw;
                    }
                *reinterpret_cast<float4 *>(grp_stats + (0 * 4 + rg) * d + k) = a;
                *reinterpret_cast<float4 *>(grp_stats + (1 * 4 + rg) * d + k) = b;
            }
        }
        if (!stats) return;
        __syncthreads();
        // merge the four groups' (count, mean, M2) per column: mean = sum n_g mean_g / n, M2 = sum [M2_g + n_g (mean_g - mean)^2]
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            float mean = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) mean += grp_stats[(0 * 4 + g) * d + k] * (float)max(0, min(nr - RG * g, RG));
            mean /= (float)nr;
            float m2 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float ng = (float)max(0, min(nr - RG * g, RG)), dm = grp_stats[(0 * 4 + g) * d + k] - mean;
                m2 += grp_stats[(1 * 4 + g) * d + k] + ng * dm * dm;
            }
            q.partial[((size_t)lb * 2 + 0) * d + k] = mean;
            q.partial[((size_t)lb * 2 + 1) * d + k] = m2;
        }
        return;
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float v[STAT_ROWS];
        float a = 0.f;
#pragma unroll 4
        for (int r = 0; r < STAT_ROWS; ++r) {
            v[r] = 0.f;
            if (r < nr) {
                float acc = q.pool == POOL_MAX ? -INFINITY : 0.f;
                for (int t = 0; t < L; ++t) {
                    const float w = q.W[(size_t)toks[r][t] * d + k];
                    acc = q.pool == POOL_MAX ? fmaxf(acc, w) : acc + w;
                }
                v[r] = q.pool == POOL_MEAN ? acc * inv_len[r] : acc;
                q.raw[(size_t)(r0 + r) * q.ld + k] = v[r];
                if (copy) q.out[(size_t)(r0 + r) * q.ld + k] = v[r];
                a += v[r];
            }
        }
        if (stats) {
            a /= (float)nr;
            float b = 0.f;
#pragma unroll 4
            for (int r = 0; r < STAT_ROWS; ++r) {
                const float dx = v[r] - a;
                b += r < nr ? dx * dx : 0.f;
            }
            q.partial[((size_t)lb * 2 + 0) * d + k] = a;
            q.partial[((size_t)lb * 2 + 1) * d + k] = b;
        }
    }
}

// backward partials: partial[b][0][k] = sum_i dy,  partial[b][1][k] = sum_i dy * xhat      (xhat = (x - mean) * rstd)
__global__ __launch_bounds__(256) void bn_partial2_kernel(const PoolBatch pb)
{
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    const int d = q.d, r0 = lb * STAT_ROWS, r1 = min(q.n, r0 + STAT_ROWS);
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float a = 0.f, b = 0.f;
        const float m = q.saved[k], rs = q.saved[d + k];
        for (int i = r0; i < r1; ++i) {
            const float dy = q.dY[(size_t)i * q.ld + k];
            a += dy;
            b += dy * ((q.raw[(size_t)i * q.ld + k] - m) * rs);
        }
        q.partial[((size_t)lb * 2 + 0) * d + k] = a;
        q.partial[((size_t)lb * 2 + 1) * d + k] = b;
    }
}

// finish in double, one workgroup per (call, 16 columns), 16 threads per column each folding a sixteenth of the row-block
// partials (4 per column and 64 columns per workgroup made this a 20-workgroup launch of 64-step dependent chains: 34 us):
// STEP 0 -> saved = {mean, rstd, unbiased variance (for the running statistics, applied in call order by bn_apply_kernel's
//           last workgroup)} of all n rows from the per-block (mean, M2) pairs;
// STEP 2 -> this call's (dbias, dweight) into saved[2..3] (added to the parameter gradients in call order by
//           pool_backward_kernel's last workgroup)
constexpr int FIN_COLS = 16, FIN_PARTS = 16;
template <int STEP>
__global__ __launch_bounds__(256) void bn_finish_kernel(const PoolBatch pb)
{
    __shared__ double red[2][FIN_PARTS][FIN_COLS];
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    const int n = q.n, d = q.d, blocks = (n + STAT_ROWS - 1) / STAT_ROWS;
    const float *partial = q.partial;
    float *saved = q.saved;
    const int c = threadIdx.x & (FIN_COLS - 1), grp = threadIdx.x / FIN_COLS, k = lb * FIN_COLS + c;
    const bool live = k < d;
    auto fold = [&](int which) {                      // fixed order: the result does not depend on scheduling
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < FIN_PARTS; ++g) t += red[which][g][c];
        return t;
    };
    double a = 0.0, b = 0.0;
    if (STEP == 0) {
        // parallel-variance merge: mean = sum n_b mean_b / n,  M2 = sum [M2_b + n_b (mean_b - mean)^2]
        if (live)
            for (int p = grp; p < blocks; p += FIN_PARTS)
                a += (double)partial[((size_t)p * 2 + 0) * d + k] * (double)min(STAT_ROWS, n - p * STAT_ROWS);
        red[0][grp][c] = a;
        __syncthreads();
        const double mean = fold(0) / n;
        if (live)
            for (int p = grp; p < blocks; p += FIN_PARTS) {
                const double dm = (double)partial[((size_t)p * 2 + 0) * d + k] - mean;
                b += (double)partial[((size_t)p * 2 + 1) * d + k] + dm * dm * (double)min(STAT_ROWS, n - p * STAT_ROWS);
            }
        red[1][grp][c] = b;
        __syncthreads();
        if (!live || grp != 0) return;
        const double var = fold(1) / n;                                  // biased: normalises the batch
        saved[k] = (float)mean;
        saved[d + k] = (float)(1.0 / sqrt(var + (double)q.eps));
        saved[2 * d + k] = (float)(var * n / (n > 1 ? n - 1 : 1));       // running statistics use the unbiased variance
    } else {
        if (live)
            for (int p = grp; p < blocks; p += FIN_PARTS) {
                a += partial[((size_t)p * 2 + 0) * d + k];
                b += partial[((size_t)p * 2 + 1) * d + k];
            }
        red[0][grp][c] = a;
        red[1][grp][c] = b;
        __syncthreads();
        if (!live || grp != 0) return;
        saved[2 * d + k] = (float)fold(0);
        saved[3 * d + k] = (float)fold(1);
    }
}

// y = (x - mean) * rstd * weight + bias     (training: saved = this call's statistics; evaluation: running ones).
// The LAST workgroup of a training launch instead updates the running statistics, call after call in the order given --
// the reference's _encode calls update one module's buffers in sequence (momentum 0.1: the order matters).
__global__ __launch_bounds__(256) void bn_apply_kernel(const PoolBatch pb, int training)
{
    if ((int)blockIdx.x == pb.cum[pb.n_calls]) {
        for (int c = 0; c < pb.n_calls; ++c) {
            const PoolCall &q = pb.c[c];
            if (!q.saved || !q.run_mean) continue;
            for (int k = threadIdx.x; k < q.d; k += blockDim.x) {
                q.run_mean[k] += q.momentum * (q.saved[k] - q.run_mean[k]);
                q.run_var[k] += q.momentum * (q.saved[2 * q.d + k] - q.run_var[k]);
            }
            __syncthreads();             // (calls of one slot share the buffers: finish a call before the next reads them)
        }
        return;
    }
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    // a workgroup takes BN_ROWS rows; thread = column (quad): the per-column factors are loaded once per workgroup
    const int d = q.d, r0 = lb * BN_ROWS, r1 = min(q.n, r0 + BN_ROWS);
    const bool vec = (d & 3) == 0 && (q.ld & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(q.raw) | reinterpret_cast<uintptr_t>(q.out)) & 15) == 0;
    if (vec) {
        for (int k = 4 * threadIdx.x; k < d; k += 4 * blockDim.x) {
            float rs[4], mu[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mu[e] = training ? q.saved[k + e] : q.run_mean[k + e];
                rs[e] = training ? q.saved[d + k + e] : 1.0f / sqrtf(q.run_var[k + e] + q.eps);
            }
            const float4 wt = *reinterpret_cast<const float4 *>(q.bn_weight + k), bias = *reinterpret_cast<const float4 *>(q.bn_bias + k);
            for (int i = r0; i < r1; ++i) {
                const float4 x = *reinterpret_cast<const float4 *>(q.raw + (size_t)i * q.ld + k);
                float4 o;                         // (x - mean) * rstd * weight + bias, the scalar path's order: same bits
                o.x = (x.x - mu[0]) * rs[0] * wt.x + bias.x;
                o.y = (x.y - mu[1]) * rs[1] * wt.y + bias.y;
                o.z = (x.z - mu[2]) * rs[2] * wt.z + bias.z;
                o.w = (x.w - mu[3]) * rs[3] * wt.w + bias.w;
                *reinterpret_cast<float4 *>(q.out + (size_t)i * q.ld + k) = o;
            }
        }
        return;
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const float mean = training ? q.saved[k] : q.run_mean[k];
        const float rs = training ? q.saved[d + k] : 1.0f / sqrtf(q.run_var[k] + q.eps);
        for (int i = r0; i < r1; ++i)
            q.out[(size_t)i * q.ld + k] = (q.raw[(size_t)i * q.ld + k] - mean) * rs * q.bn_weight[k] + q.bn_bias[k];
    }
}

// dx = weight * rstd * (dy - dbias/n - xhat * dweight/n)   (or dx = dy without batch-norm), then scattered into the
// token table gradient: sum -> every token of the row, mean -> scaled by 1/(len + 1e-12), max -> the first token that
// attains the maximum of its column.  Token 0 (padding_idx) receives nothing.
// One workgroup takes POOL_BWD_ROWS rows, one thread per column.  The reference's token ids are frequency-ranked
// (index_mapper.py:95-108 writes the maps sorted by count; BOS = 2 and EOS = 3 sit in every row), so the first
// HOT_TOKENS ids would take thousands of same-address atomics per step: their rows are accumulated in LDS (a thread
// owns its column, so plain read-modify-write) and flushed with one atomic per touched (token, column) per workgroup.
// The LAST workgroup adds the calls' batch-norm parameter gradients (bn_finish_kernel<2>) up in call order.
__global__ __launch_bounds__(256) void pool_backward_kernel(const PoolBatch pb)
{
    extern __shared__ float hot[];                      // [HOT_TOKENS][d]
    __shared__ uint32_t hot_seen;
    __shared__ int32_t toks[POOL_BWD_ROWS][POOL_MAX_LEN];
    __shared__ float inv_len[POOL_BWD_ROWS];
    if ((int)blockIdx.x == pb.cum[pb.n_calls]) {
        for (int c = 0; c < pb.n_calls; ++c) {
            const PoolCall &q = pb.c[c];
            if (!q.saved) continue;
            for (int k = threadIdx.x; k < q.d; k += blockDim.x) {
                q.d_bias[k] += q.saved[2 * q.d + k];
                q.d_weight[k] += q.saved[3 * q.d + k];
            }
            __syncthreads();
        }
        return;
    }
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    const int d = q.d, L = q.L, n = q.n, pool = q.pool;
    const float *saved = q.saved, *X = q.raw, *DY = q.dY, *W = q.W;
    float *dW = q.dW;
    const int64_t ldx = q.ld, lddy = q.ld;
    const int r0 = lb * POOL_BWD_ROWS, nr = min(n, r0 + POOL_BWD_ROWS) - r0;
    for (int i = threadIdx.x; i < HOT_TOKENS * d; i += blockDim.x) hot[i] = 0.f;
    for (int i = threadIdx.x; i < nr * L; i += blockDim.x)
        toks[i / L][i % L] = q.tokens[(size_t)row_id(q.ids, q.first_id, r0 + i / L, q.n_ids, nullptr) * L + i % L];
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
            const float m = saved[k], rs = saved[d + k], db = saved[2 * d + k], dw = saved[3 * d + k], wk = q.bn_weight[k];
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

template <typename F>
PoolBatch make_batch(const PoolCall *calls, int n_calls, int *id_err, F blocks_of)
{
    PoolBatch pb;
    pb.n_calls = n_calls;
    pb.id_err = id_err;
    pb.cum[0] = 0;
    for (int i = 0; i < POOL_MAX_CALLS; ++i) {
        if (i < n_calls) pb.c[i] = calls[i];
        pb.cum[i + 1] = pb.cum[i] + (i < n_calls ? blocks_of(calls[i]) : 0);
    }
    return pb;
}

}  // namespace

size_t pool_workspace_bytes(int n, int d)
{
    const size_t blocks = (size_t)(n + STAT_ROWS - 1) / STAT_ROWS;
    return (blocks * 2 * (size_t)d * sizeof(float) + 255) / 256 * 256;
}

// forward of a batch of _encode calls (each with n > 0): pooled rows -> raw; batch-norm (calls with `saved`): training-mode
// statistics of THAT call's rows into saved[0..2d), running statistics updated in call order, normalised rows -> out;
// evaluation mode: running statistics.  calls[i].partial must hold pool_workspace_bytes(n_i, d_i).
hipError_t launch_pool_encode_calls(const PoolCall *calls, int n_calls, int training, int *id_err, hipStream_t st)
{
    if (n_calls <= 0) return hipSuccess;
    if (n_calls > POOL_MAX_CALLS) return hipErrorInvalidValue;
    {
        const PoolBatch pb = make_batch(calls, n_calls, id_err, [](const PoolCall &q) { return (q.n + STAT_ROWS - 1) / STAT_ROWS; });
        int dmax = 0;
        for (int i = 0; i < n_calls; ++i) dmax = calls[i].d > dmax ? calls[i].d : dmax;
        hipLaunchKernelGGL(pool_stats_kernel, dim3(pb.cum[n_calls]), dim3(256), sizeof(float) * 2 * 4 * dmax, st, pb, training);
    }
    bool any_bn = false;
    for (int i = 0; i < n_calls; ++i) any_bn = any_bn || calls[i].bn_weight != nullptr;
    if (!any_bn) return hipGetLastError();
    // calls without batch-norm drop out of the two batch-norm launches
    PoolCall bn[POOL_MAX_CALLS];
    int nb = 0;
    for (int i = 0; i < n_calls; ++i)
        if (calls[i].bn_weight) bn[nb++] = calls[i];
    if (training) {
        const PoolBatch pf = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.d + FIN_COLS - 1) / FIN_COLS; });
        hipLaunchKernelGGL(bn_finish_kernel<0>, dim3(pf.cum[nb]), dim3(256), 0, st, pf);
    }
    const PoolBatch pa = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.n + BN_ROWS - 1) / BN_ROWS; });
    hipLaunchKernelGGL(bn_apply_kernel, dim3(pa.cum[nb] + (training ? 1 : 0)), dim3(256), 0, st, pa, training);
    return hipGetLastError();
}

hipError_t launch_pool_backward_calls(const PoolCall *calls, int n_calls, int *id_err, hipStream_t st)
{
    if (n_calls <= 0) return hipSuccess;
    if (n_calls > POOL_MAX_CALLS) return hipErrorInvalidValue;
    PoolCall bn[POOL_MAX_CALLS];
    int nb = 0, dmax = 0;
    for (int i = 0; i < n_calls; ++i) {
        if (calls[i].saved) bn[nb++] = calls[i];
        dmax = calls[i].d > dmax ? calls[i].d : dmax;
    }
    if (nb) {
        const PoolBatch p1 = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.n + STAT_ROWS - 1) / STAT_ROWS; });
        hipLaunchKernelGGL(bn_partial2_kernel, dim3(p1.cum[nb]), dim3(256), 0, st, p1);
        const PoolBatch p2 = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.d + FIN_COLS - 1) / FIN_COLS; });
        hipLaunchKernelGGL(bn_finish_kernel<2>, dim3(p2.cum[nb]), dim3(256), 0, st, p2);
    }
    const PoolBatch p3 = make_batch(calls, n_calls, id_err, [](const PoolCall &q) { return (q.n + POOL_BWD_ROWS - 1) / POOL_BWD_ROWS; });
    hipLaunchKernelGGL(pool_backward_kernel, dim3(p3.cum[n_calls] + (nb ? 1 : 0)), dim3(256), sizeof(float) * HOT_TOKENS * dmax, st, p3);
    return hipGetLastError();
}

}  // namespace okge
