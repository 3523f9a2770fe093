// Backward of the prefix scores for callers that hold the (b, N) gradient block: the plugin methods sp_prefix_score /
// po_prefix_score / _score called with gradients enabled (a user's own loss on the scores; openkge/model.py:52-77, :198-229,
// :268-274 -- the reference gets these gradients from ATen's autograd through its four / one matrix products).
//   dQ = G . C        (b x n) . (n x d)   -> chain rule of the folded query -> d_ent, d_rel
//   dC = G^T . Q      (n x b) . (b x d)
// Both products on ONE hand-written exact-fp32 MFMA kernel (v_mfma_f32_16x16x4_f32); the fused training path
// (okge_train_forward_backward: G never leaves the chip) does not come through here.
//
// gemm_f32_kernel<TA>: C[M][N] = op(A) . B, B = [K][N] row-major, A = [M][K] (TA = false) or [K][M] (TA = true) row-major.
// Workgroup = 4 waves = a 64 x 64 output tile, wave (wm, wn) a 32 x 32 quarter = 2 x 2 MFMA blocks; K in chunks of 16 staged
// through LDS (next chunk's global loads in registers while the current one is multiplied); blockIdx.z splits K, every split
// writes its own slab and splitk_reduce_kernel adds the slabs in split order (no float atomics: reproducible).
// Also here: the scatter of encoded-row gradients into a table's dense gradient by SORTED ids (EncodeRowsFn.backward).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>

#include "okge_device.h"
#include "okge_eval_device.h"
#include "okge_kernels.h"

namespace okge {

namespace {

constexpr int GM = 64, GN = 64, GK = 16;
constexpr int LDA_N = GK + 4;          // A tile [64 m][16 k] (TA = false): 4 * odd, the operand read (16 rows x 4 columns) hits 64 banks
constexpr int LDT = 80;                // [16 k][64] tiles (B, and A when TA): rows 16 banks apart, 4 rows x 16 columns hit 64 banks

template <bool TA>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, int64_t lda, const float *__restrict__ B, int64_t ldb,
                                                       float *__restrict__ C, int64_t ldc, int M, int N, int K, int k_per_split,
                                                       int64_t slab_stride)
{
    __shared__ float As[TA ? GK * LDT : GM * LDA_N];
    __shared__ float Bs[GK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wm = w >> 1, wn = w & 1;
    const int m0 = blockIdx.x * GM, n0 = blockIdx.y * GN;
    const int k_lo = blockIdx.z * k_per_split, k_hi = min(K, k_lo + k_per_split);
    float *Cz = C + (size_t)blockIdx.z * slab_stride;
    v4f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    float ra[4], rb[4];
    auto load = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (TA) {                                    // A[k][m]: rows of 64 consecutive m
                const int kk = (tid >> 6) + 4 * j, m = m0 + (tid & 63);
                ra[j] = (k0 + kk < k_hi && m < M) ? A[(size_t)(k0 + kk) * lda + m] : 0.f;
            } else {                                     // A[m][k]: 16 consecutive k of 16 rows per wave-instruction
                const int m = m0 + (tid >> 4) + 16 * j, kk = tid & 15;
                ra[j] = (m < M && k0 + kk < k_hi) ? A[(size_t)m * lda + k0 + kk] : 0.f;
            }
            const int kk = (tid >> 6) + 4 * j, n = n0 + (tid & 63);
            rb[j] = (k0 + kk < k_hi && n < N) ? B[(size_t)(k0 + kk) * ldb + n] : 0.f;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (TA) As[((tid >> 6) + 4 * j) * LDT + (tid & 63)] = ra[j];
            else As[((tid >> 4) + 16 * j) * LDA_N + (tid & 15)] = ra[j];
            Bs[((tid >> 6) + 4 * j) * LDT + (tid & 63)] = rb[j];
        }
    };
    if (k_lo < k_hi) load(k_lo);
    for (int k0 = k_lo; k0 < k_hi; k0 += GK) {
        __syncthreads();                                 // the previous chunk has been multiplied
        stage();
        __syncthreads();
        if (k0 + GK < k_hi) load(k0 + GK);
#pragma unroll
        for (int k4 = 0; k4 < GK; k4 += 4) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int m = 32 * wm + 16 * i + (lane & 15), kk = k4 + (lane >> 4);
                a[i] = TA ? As[kk * LDT + m] : As[m * LDA_N + kk];
                b[i] = Bs[kk * LDT + 32 * wn + 16 * i + (lane & 15)];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a[i], b[j], acc[i][j]);
        }
    }
    // result register r of lane l: C[m = 4 (l >> 4) + r][n = l & 15] of the block
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wm + 16 * i + 4 * (lane >> 4) + r, n = n0 + 32 * wn + 16 * j + (lane & 15);
                if (m < M && n < N) Cz[(size_t)m * ldc + n] = acc[i][j][r];
            }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float *__restrict__ slab, int splits, int64_t slab_stride, int64_t n,
                                                            float *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = 0.f;
        for (int s = 0; s < splits; ++s) v += slab[(size_t)s * slab_stride + i];
        out[i] = v;
    }
}

// q = fold(ent, rel): the operand with score = q . cand^T (model.py:205-216 regrouped, :268-274), same roundings as the forward
__global__ __launch_bounds__(128) void fold_rows_kernel(int scorer, int sp, const float *__restrict__ ent, int64_t ld_e,
                                                        const float *__restrict__ rel, int64_t ld_r, int d, float *__restrict__ q)
{
    const int b = blockIdx.x;
    const float *e = ent + (size_t)b * ld_e, *r = rel + (size_t)b * ld_r;
    float *o = q + (size_t)b * d;
    if (scorer == SC_DISTMULT) {
        for (int k = threadIdx.x; k < d; k += blockDim.x) o[k] = __fmul_rn(e[k], r[k]);
        return;
    }
    const int h = d >> 1;
    for (int k = threadIdx.x; k < h; k += blockDim.x) fold_complex(sp != 0, e[k], e[h + k], r[k], r[h + k], o[k], o[h + k]);
}

// (d_ent, d_rel) from dq: the transpose of the fold
__global__ __launch_bounds__(128) void fold_backward_kernel(int scorer, int sp, const float *__restrict__ ent, int64_t ld_e,
                                                            const float *__restrict__ rel, int64_t ld_r, const float *__restrict__ dq,
                                                            int d, float *__restrict__ d_ent, float *__restrict__ d_rel)
{
    const int b = blockIdx.x;
    const float *e = ent + (size_t)b * ld_e, *r = rel + (size_t)b * ld_r, *g = dq + (size_t)b * d;
    float *de = d_ent ? d_ent + (size_t)b * d : nullptr, *dr = d_rel ? d_rel + (size_t)b * d : nullptr;
    if (scorer == SC_DISTMULT) {
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            if (de) de[k] = g[k] * r[k];
            if (dr) dr[k] = g[k] * e[k];
        }
        return;
    }
    const int h = d >> 1;
    for (int k = threadIdx.x; k < h; k += blockDim.x) {
        const float e1 = e[k], e2 = e[h + k], r1 = r[k], r2 = r[h + k], g1 = g[k], g2 = g[h + k];
        if (sp) {            // q1 = e1 r1 - e2 r2, q2 = e2 r1 + e1 r2
            if (de) { de[k] = g1 * r1 + g2 * r2; de[h + k] = g2 * r1 - g1 * r2; }
            if (dr) { dr[k] = g1 * e1 + g2 * e2; dr[h + k] = g2 * e1 - g1 * e2; }
        } else {             // q1 = e1 r1 + e2 r2, q2 = e2 r1 - e1 r2
            if (de) { de[k] = g1 * r1 - g2 * r2; de[h + k] = g1 * r2 + g2 * r1; }
            if (dr) { dr[k] = g1 * e1 + g2 * e2; dr[h + k] = g1 * e2 - g2 * e1; }
        }
    }
}

// table[id] += sum of the (masked) gradient rows of the positions that named id, in the order of `order` (positions sorted by
// id, stable: a fixed order, no float atomics).  One workgroup per sorted position; the first position of a run of equal ids
// owns the run.  ids == nullptr: position i names row first_id + i (a run of one).  Row 0 is the padding row: no gradient.
__global__ __launch_bounds__(128) void scatter_rows_kernel(const float *__restrict__ rows, int64_t ld, const int32_t *__restrict__ ids,
                                                           const int32_t *__restrict__ order, int first_id, int n, int d, const DropDev drop,
                                                           float *__restrict__ table, int64_t table_rows, int *__restrict__ id_err)
{
    const int i = blockIdx.x;
    auto pos_of = [&](int j) { return order ? min(max(order[j], 0), n - 1) : j; };
    auto id_of = [&](int j) { return ids ? ids[pos_of(j)] : first_id + j; };
    const int id = id_of(i);
    if (ids && i > 0 && id_of(i - 1) == id) return;
    const int64_t row = checked_row(id, table_rows, threadIdx.x == 0 ? id_err : nullptr);
    if (row == 0) return;
    int hi = i + 1;
    if (ids)
        while (hi < n && id_of(hi) == id) ++hi;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float acc = 0.f;
        for (int j = i; j < hi; ++j) {
            const int pos = pos_of(j);
            acc += rows[(size_t)pos * ld + k] * drop_mult1(drop, (uint32_t)pos, k, d);
        }
        table[row * d + k] += acc;
    }
}

}  // namespace

size_t score_backward_workspace_bytes(int b, int n, int d)
{
    // q [b][d] + dq [b][d] + the split-K slabs of dQ = G . C (at most 64 splits of [b][d])
    return (size_t)(2 + 64) * (size_t)b * d * sizeof(float) + 512;
}

hipError_t launch_score_backward(int scorer, int sp, const float *G, int64_t ld_g, int b, int n, const float *ent, int64_t ld_ent,
                                 const float *rel, int64_t ld_rel, const float *cand, int64_t ld_cand, int d, float *d_ent,
                                 float *d_rel, float *d_cand, void *workspace, hipStream_t st)
{
    float *q = static_cast<float *>(workspace), *dq = q + (size_t)b * d, *slab = dq + (size_t)b * d;
    if (d_cand) {                                        // dC = G^T . Q: A = G read as [K = b][M = n]
        hipLaunchKernelGGL(fold_rows_kernel, dim3(b), dim3(128), 0, st, scorer, sp, ent, ld_ent, rel, ld_rel, d, q);
        hipLaunchKernelGGL(gemm_f32_kernel<true>, dim3((n + GM - 1) / GM, (d + GN - 1) / GN, 1), dim3(256), 0, st, G, ld_g, q, (int64_t)d,
                           d_cand, (int64_t)d, n, d, b, b, (int64_t)0);
    }
    if (d_ent || d_rel) {                                // dQ = G . C: few output tiles, a long contraction: split it
        const int tiles = ((b + GM - 1) / GM) * ((d + GN - 1) / GN);
        int splits = std::max(1, std::min(64, std::min(1024 / std::max(tiles, 1), (n + 4 * GK - 1) / (4 * GK))));
        int k_per = ((n + splits - 1) / splits + GK - 1) / GK * GK;
        splits = (n + k_per - 1) / k_per;
        float *out = splits > 1 ? slab : dq;
        hipLaunchKernelGGL(gemm_f32_kernel<false>, dim3((b + GM - 1) / GM, (d + GN - 1) / GN, splits), dim3(256), 0, st, G, ld_g, cand, ld_cand,
                           out, (int64_t)d, b, d, n, k_per, (int64_t)b * d);
        if (splits > 1) {
            const int64_t total = (int64_t)b * d;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)std::min<int64_t>(1024, (total + 255) / 256)), dim3(256), 0, st, slab,
                               splits, total, total, dq);
        }
        hipLaunchKernelGGL(fold_backward_kernel, dim3(b), dim3(128), 0, st, scorer, sp, ent, ld_ent, rel, ld_rel, dq, d, d_ent, d_rel);
    }
    return hipGetLastError();
}

hipError_t launch_scatter_rows(const float *rows, int64_t ld, const int32_t *ids, const int32_t *order, int first_id, int n, int d,
                               const DropDev &drop, float *table, int64_t table_rows, int *id_err, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(128), 0, st, rows, ld, ids, order, first_id, n, d, drop, table, table_rows, id_err);
    return hipGetLastError();
}

}  // namespace okge
