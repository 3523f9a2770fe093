// Fused prefix-scoring kernels for gfx950 (MI355X).
//
//   fused_tile_kernel<KBMAX, MODE>
//     One workgroup owns a tile of NT=64 candidate entities (gathered + dropped-out once into LDS) and
//     sweeps the batch's folded query rows in chunks of BC=64:
//        X  = Q_chunk . C_tile^T            (v_mfma_f32_16x16x4_f32, exact fp32)       [all modes]
//        G  = dLoss/dX / normalizer, loss   (BCE / KL epilogue in registers)           [train]
//        dC += G^T . Q_chunk                (accumulators stay in registers)           [train]
//     and leaves G^T (N x B) in HBM for the query-gradient kernel.  Replaces the reference's
//     encode_obj(candidates) + 4 mm + cat + BCEWithLogits/log_softmax+KLDiv forward and the mm/sigmoid
//     half of autograd's backward (openkge/model.py:198-229,268-274; openkge/trainer.py:75-106,234).
//     MODE_SCORE writes X (evaluation / *_prefix_score); MODE_STATS writes per-row (max, sum-exp)
//     partials for the KL loss' log_softmax.
//
//   dq_kernel<KBMAX>
//     dQ = G . C : one workgroup per (64-row batch block, candidate range); partial slabs are summed by
//     prefix_backward_kernel (okge_misc.hip).
//
// LDS tiles use leading dimension D16+4 (4*odd floats): conflict-free ds_read_b128 along k for the
// score product and conflict-free ds_read_b32 along rows for the two gradient products (okge_device.h).
#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

// ---- candidate tile: gather rows of E, apply dropout, park in LDS as Cs[NT][LDK] (zero padded) -----
__device__ __forceinline__ void load_cand_tile(float *Cs, const float *__restrict__ E, int d, int KB, int LDK,
                                               const int32_t *__restrict__ cand_ids, int cand_first, int N,
                                               int n0, const DropDev &drop, bool vec_ok, int tid)
{
    const int r = tid >> 2, q = tid & 3;
    const int n = n0 + r;
    const bool valid = n < N;
    int64_t cid = 0;
    if (valid) cid = cand_ids ? (int64_t)cand_ids[n] : (int64_t)cand_first + n;
    const float *row = E + cid * d;
    for (int it = 0; it < KB; ++it) {
        const int q4 = q + 4 * it, k = 4 * q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid && k < d) {
            if (vec_ok) {
                v = *reinterpret_cast<const float4 *>(row + k);
            } else {
                v.x = row[k];
                if (k + 1 < d) v.y = row[k + 1];
                if (k + 2 < d) v.z = row[k + 2];
                if (k + 3 < d) v.w = row[k + 3];
            }
            const float4 m = drop_mult4(drop, (uint32_t)n, q4, d);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        *reinterpret_cast<float4 *>(Cs + r * LDK + k) = v;
    }
}

__device__ __forceinline__ int lower_bound_i32(const int32_t *__restrict__ a, int n, int key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int KBMAX, int MODE>
__global__ __launch_bounds__(FUSED_THREADS) void fused_tile_kernel(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LDK = a.LDK, KB = a.KB, d = a.d;
    float *Cs = reinterpret_cast<float *>(smem);              // [NT][LDK]
    float *Qs = Cs + NT * LDK;                                // [BC][LDK]   (end: dC staging)
    float *GT = Qs + BC * LDK;                                // [NT][LDG]   G^T tile (score mode: X[BC][LDG])
    uint32_t *ybits = reinterpret_cast<uint32_t *>(GT + NT * LDG);   // [BC][2] label bits of the chunk
    double *red = reinterpret_cast<double *>(ybits + BC * 2);        // [4]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int n0 = blockIdx.x * NT;
    const int b_begin = blockIdx.y * a.b_per_block;
    const int b_end = min(a.B, b_begin + a.b_per_block);
    const bool vec_ok = (d & 3) == 0;

    load_cand_tile(Cs, a.E, d, KB, LDK, a.cand_ids, a.cand_first, a.N, n0, a.drop_c, vec_ok, tid);
    if (tid < BC * 2) ybits[tid] = 0u;

    int pos_lo = 0, pos_hi = 0;
    if (MODE == MODE_TRAIN) {
        pos_lo = lower_bound_i32(a.pos_col, a.nnz, n0);
        pos_hi = lower_bound_i32(a.pos_col, a.nnz, n0 + NT);
    }

    // register-staged query chunk: thread (row r = tid/4, quarter q = tid%4) holds KB float4 of that row
    const int qr = tid >> 2, qq = tid & 3;
    float4 qreg[KBMAX];
    auto fetch_chunk = [&](int b0) {
        const int b = b0 + qr;
        const float *src = a.Q + (size_t)b * a.ldq + 4 * qq;
#pragma unroll
        for (int it = 0; it < KBMAX; ++it) {
            if (it < KB) {
                qreg[it] = (b < b_end) ? *reinterpret_cast<const float4 *>(src + 16 * it)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    fetch_chunk(b_begin);

    v4f dc[KBMAX];                                   // dC[n = 16w + 4s + i][k = 16kb + c]
#pragma unroll
    for (int kb = 0; kb < KBMAX; ++kb) dc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;

    for (int b0 = b_begin; b0 < b_end; b0 += BC) {
        // ---- phase A: park the prefetched chunk in LDS, set label bits, prefetch the next chunk -----
#pragma unroll
        for (int it = 0; it < KBMAX; ++it)
            if (it < KB) *reinterpret_cast<float4 *>(Qs + qr * LDK + 4 * qq + 16 * it) = qreg[it];
        if (MODE == MODE_TRAIN) {
            for (int p = pos_lo + tid; p < pos_hi; p += FUSED_THREADS) {
                const int row = a.pos_row[p] - b0;
                if (row >= 0 && row < BC) {
                    const int col = a.pos_col[p] - n0;
                    atomicOr(&ybits[row * 2 + (col >> 5)], 1u << (col & 31));
                }
            }
        }
        if (b0 + BC < b_end) fetch_chunk(b0 + BC);
        __syncthreads();

        // ---- phase B: X = Q_chunk . C^T ; wave w owns rows 16w..16w+15, all four 16-wide n blocks ---
        v4f x[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) x[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            const float *qa = Qs + (16 * w + c) * LDK + 4 * s;
            const float *cb = Cs + c * LDK + 4 * s;
            for (int r = 0; r < KB; ++r) {
                const float4 av = *reinterpret_cast<const float4 *>(qa + 16 * r);
                float4 bv[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    bv[nb] = *reinterpret_cast<const float4 *>(cb + 16 * nb * LDK + 16 * r);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) x[nb] = mfma16(av.x, bv[nb].x, x[nb]);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) x[nb] = mfma16(av.y, bv[nb].y, x[nb]);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) x[nb] = mfma16(av.z, bv[nb].z, x[nb]);
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) x[nb] = mfma16(av.w, bv[nb].w, x[nb]);
            }
        }
        // lane holds X[b = b0 + 16w + 4s + i][n = n0 + 16nb + c] in x[nb][i]

        if (MODE == MODE_SCORE) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) GT[(16 * w + 4 * s + i) * LDG + 16 * nb + c] = x[nb][i];
            __syncthreads();
            for (int idx = tid; idx < BC * 16; idx += FUSED_THREADS) {
                const int r = idx >> 4, c4 = idx & 15;
                const int b = b0 + r, n = n0 + 4 * c4;
                if (b < b_end) {
                    const float4 v = *reinterpret_cast<const float4 *>(GT + r * LDG + 4 * c4);
                    float *dst = a.X + (size_t)b * a.ldx + n;
                    if (a.x_vec_ok && n + 3 < a.N) {
                        *reinterpret_cast<float4 *>(dst) = v;
                    } else {
                        if (n + 0 < a.N) dst[0] = v.x;
                        if (n + 1 < a.N) dst[1] = v.y;
                        if (n + 2 < a.N) dst[2] = v.z;
                        if (n + 3 < a.N) dst[3] = v.w;
                    }
                }
            }
            __syncthreads();
            continue;
        }

        if (MODE == MODE_STATS) {
            // per row of the chunk: max and sum-exp over this tile's candidates
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m = -INFINITY;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    if (n0 + 16 * nb + c < a.N) m = fmaxf(m, x[nb][i]);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
                float se = 0.f;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    if (n0 + 16 * nb + c < a.N) se += __expf(x[nb][i] - m);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) se += __shfl_xor(se, o);
                const int b = b0 + 16 * w + 4 * s + i;
                if (c == 0 && b < b_end) {
                    float2 *dst = reinterpret_cast<float2 *>(a.stats) + (size_t)blockIdx.x * a.Bpad + b;
                    *dst = make_float2(m, se);
                }
            }
            __syncthreads();
            continue;
        }

        // ---- train epilogue: G = dLoss/dX / normalizer, loss; G^T tile to LDS -----------------------
        {
            uint32_t yw[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint2 t = *reinterpret_cast<const uint2 *>(ybits + (16 * w + 4 * s + i) * 2);
                yw[i][0] = t.x; yw[i][1] = t.y;
            }
            float lse[4] = {0.f, 0.f, 0.f, 0.f}, ysum[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.loss_kind == LOSS_KL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = min(b0 + 16 * w + 4 * s + i, a.B - 1);
                    lse[i] = a.row_lse[b];
                    ysum[i] = a.row_ysum[b];
                }
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int nl = 16 * nb + c;
                const bool nvalid = n0 + nl < a.N;
                float4 gv;
                float *gp = &gv.x;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = x[nb][i];
                    const bool valid = nvalid && (b0 + 16 * w + 4 * s + i < b_end);
                    const bool pos = (yw[i][nl >> 5] >> (nl & 31)) & 1u;
                    float g, l;
                    if (a.loss_kind == LOSS_BCE) {
                        // BCEWithLogits: max(x,0) - x*y + log1p(exp(-|x|)); d/dx = sigmoid(x) - y
                        const float y = pos ? a.y_pos : a.y_neg;
                        const float e = __expf(-fabsf(xv));
                        const float rcp = __frcp_rn(1.f + e);
                        const float sig = xv >= 0.f ? rcp : e * rcp;
                        l = fmaxf(xv, 0.f) - xv * y + __logf(1.f + e);
                        g = sig - y;
                    } else {
                        // KLDiv(sum)(log_softmax(x), y), y in {0,1} unnormalised:
                        // loss = -sum_pos log_softmax; d/dx = softmax * sum_n y - y
                        const float lsm = xv - lse[i];
                        const float y = pos ? 1.f : 0.f;
                        l = pos ? -lsm : 0.f;
                        g = __expf(lsm) * ysum[i] - y;
                    }
                    lsum += valid ? l : 0.f;
                    gp[i] = valid ? g * a.inv_norm : 0.f;
                }
                *reinterpret_cast<float4 *>(GT + nl * LDG + 16 * w + 4 * s) = gv;
            }
        }
        __syncthreads();

        // ---- phase C: G^T tile -> HBM (for dq_kernel); dC += G^T . Q_chunk --------------------------
        for (int idx = tid; idx < NT * 16; idx += FUSED_THREADS) {
            const int r = idx >> 4, c4 = idx & 15;
            if (n0 + r < a.N)
                *reinterpret_cast<float4 *>(a.GT + (size_t)(n0 + r) * a.ldgt + b0 + 4 * c4) =
                    *reinterpret_cast<const float4 *>(GT + r * LDG + 4 * c4);
        }
        {
            const float *ga = GT + (16 * w + c) * LDG + 4 * s;    // A[i = n][slot] = G[b = 16tq + 4s + j][n]
            const float *qb = Qs + 4 * s * LDK + c;               // B[slot][k]     = Q[b][16kb + c]
#pragma unroll
            for (int tq = 0; tq < BC / 16; ++tq) {
                const float4 gq = *reinterpret_cast<const float4 *>(ga + 16 * tq);
                const float gj[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float *qrow = qb + (16 * tq + j) * LDK;
#pragma unroll
                    for (int kb = 0; kb < KBMAX; ++kb)
                        if (kb < KB) dc[kb] = mfma16(gj[j], qrow[16 * kb], dc[kb]);
                }
            }
        }
        if (tid < BC * 2) ybits[tid] = 0u;
        __syncthreads();
    }

    if (MODE != MODE_TRAIN) return;

    // ---- dC epilogue: stage through LDS, apply the candidates' dropout mask, add into dE rows --------
    float *stage = Qs;
#pragma unroll
    for (int kb = 0; kb < KBMAX; ++kb)
        if (kb < KB)
#pragma unroll
            for (int i = 0; i < 4; ++i) stage[(16 * w + 4 * s + i) * LDK + 16 * kb + c] = dc[kb][i];
    {
        const double ls = wave_sum((double)lsum);
        if (lane == 0) red[w] = ls;
    }
    __syncthreads();
    if (tid == 0) a.loss_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    {
        const int r = tid >> 2, q = tid & 3;
        const int n = n0 + r;
        if (n < a.N) {
            const int64_t cid = a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n;
            float *drow = a.dE + cid * d;
            const bool exclusive = gridDim.y == 1;
            for (int it = 0; it < KB; ++it) {
                const int q4 = q + 4 * it, k = 4 * q4;
                if (k >= d) break;
                float4 v = *reinterpret_cast<const float4 *>(stage + r * LDK + k);
                const float4 m = drop_mult4(a.drop_c, (uint32_t)n, q4, d);
                v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
                if (exclusive && vec_ok) {
                    float4 o = *reinterpret_cast<float4 *>(drow + k);
                    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
                    *reinterpret_cast<float4 *>(drow + k) = o;
                } else if (exclusive) {
                    drow[k] += v.x;
                    if (k + 1 < d) drow[k + 1] += v.y;
                    if (k + 2 < d) drow[k + 2] += v.z;
                    if (k + 3 < d) drow[k + 3] += v.w;
                } else {
                    atomicAdd(drow + k, v.x);
                    if (k + 1 < d) atomicAdd(drow + k + 1, v.y);
                    if (k + 2 < d) atomicAdd(drow + k + 2, v.z);
                    if (k + 3 < d) atomicAdd(drow + k + 3, v.w);
                }
            }
        }
    }
}

// ---- dQ = G . C over a candidate range -------------------------------------------------------------
template <int KBMAX>
__global__ __launch_bounds__(FUSED_THREADS) void dq_kernel(const DqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LDK = a.LDK, KB = a.KB, d = a.d;
    float *Cs = reinterpret_cast<float *>(smem);      // [NT][LDK]
    float *Gs = Cs + NT * LDK;                        // [NT (n)][LDG] : G^T tile, 64 batch rows wide

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int split = blockIdx.x % a.nsplit, bblk = blockIdx.x / a.nsplit;
    const int b0 = bblk * BC;
    const int nchunks = (a.N + NT - 1) / NT;
    const int ch_lo = (int)((int64_t)split * nchunks / a.nsplit);
    const int ch_hi = (int)((int64_t)(split + 1) * nchunks / a.nsplit);
    const bool vec_ok = (d & 3) == 0;

    v4f acc[KBMAX];                                   // dQ[b = b0 + 16w + 4s + i][k = 16kb + c]
#pragma unroll
    for (int kb = 0; kb < KBMAX; ++kb) acc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};

    for (int ch = ch_lo; ch < ch_hi; ++ch) {
        const int n0 = ch * NT;
        load_cand_tile(Cs, a.E, d, KB, LDK, a.cand_ids, a.cand_first, a.N, n0, a.drop_c, vec_ok, tid);
        for (int idx = tid; idx < NT * 16; idx += FUSED_THREADS) {
            const int r = idx >> 4, c4 = idx & 15;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n0 + r < a.N)
                v = *reinterpret_cast<const float4 *>(a.GT + (size_t)(n0 + r) * a.ldgt + b0 + 4 * c4);
            *reinterpret_cast<float4 *>(Gs + r * LDG + 4 * c4) = v;
        }
        __syncthreads();
        {
            const float *ga = Gs + 4 * s * LDG + 16 * w + c;   // A[i = b][slot] = G^T[n = 16tq + 4s + j][b]
            const float *cb = Cs + 4 * s * LDK + c;            // B[slot][k]     = C[n][16kb + c]
#pragma unroll
            for (int tq = 0; tq < NT / 16; ++tq) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float av = ga[(16 * tq + j) * LDG];
                    const float *crow = cb + (16 * tq + j) * LDK;
#pragma unroll
                    for (int kb = 0; kb < KBMAX; ++kb)
                        if (kb < KB) acc[kb] = mfma16(av, crow[16 * kb], acc[kb]);
                }
            }
        }
        __syncthreads();
    }
    float *dst = a.slab + ((size_t)split * a.Bpad + b0 + 16 * w + 4 * s) * a.ldq + c;
#pragma unroll
    for (int kb = 0; kb < KBMAX; ++kb)
        if (kb < KB)
#pragma unroll
            for (int i = 0; i < 4; ++i) dst[(size_t)i * a.ldq + 16 * kb] = acc[kb][i];
}

// ---- host-side launchers -----------------------------------------------------------------------------
template <int KBMAX, int MODE>
static hipError_t launch_fused_t(const FusedArgs &a, dim3 grid, size_t shmem, hipStream_t st)
{
    auto k = fused_tile_kernel<KBMAX, MODE>;
    static size_t configured = 0;
    if (shmem > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        configured = shmem;
    }
    hipLaunchKernelGGL(k, grid, dim3(FUSED_THREADS), shmem, st, a);
    return hipGetLastError();
}

template <int KBMAX>
static hipError_t launch_fused_m(int mode, const FusedArgs &a, dim3 grid, size_t shmem, hipStream_t st)
{
    switch (mode) {
        case MODE_TRAIN: return launch_fused_t<KBMAX, MODE_TRAIN>(a, grid, shmem, st);
        case MODE_SCORE: return launch_fused_t<KBMAX, MODE_SCORE>(a, grid, shmem, st);
        default:         return launch_fused_t<KBMAX, MODE_STATS>(a, grid, shmem, st);
    }
}

size_t fused_shmem_bytes(int LDK)
{
    return (size_t)(NT + BC) * LDK * sizeof(float) + (size_t)NT * LDG * sizeof(float) + BC * 2 * sizeof(uint32_t) +
           4 * sizeof(double);
}

size_t dq_shmem_bytes(int LDK)
{
    return (size_t)NT * LDK * sizeof(float) + (size_t)NT * LDG * sizeof(float);
}

hipError_t launch_fused(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st)
{
    const dim3 grid(grid_x, grid_y);
    const size_t shmem = fused_shmem_bytes(a.LDK);
    if (a.KB <= 4)  return launch_fused_m<4>(mode, a, grid, shmem, st);
    if (a.KB <= 8)  return launch_fused_m<8>(mode, a, grid, shmem, st);
    if (a.KB <= 13) return launch_fused_m<13>(mode, a, grid, shmem, st);
    if (a.KB <= 16) return launch_fused_m<16>(mode, a, grid, shmem, st);
    return hipErrorInvalidValue;
}

template <int KBMAX>
static hipError_t launch_dq_t(const DqArgs &a, int grid_x, size_t shmem, hipStream_t st)
{
    auto k = dq_kernel<KBMAX>;
    static size_t configured = 0;
    if (shmem > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
        configured = shmem;
    }
    hipLaunchKernelGGL(k, dim3(grid_x), dim3(FUSED_THREADS), shmem, st, a);
    return hipGetLastError();
}

hipError_t launch_dq(const DqArgs &a, int grid_x, hipStream_t st)
{
    const size_t shmem = dq_shmem_bytes(a.LDK);
    if (a.KB <= 4)  return launch_dq_t<4>(a, grid_x, shmem, st);
    if (a.KB <= 8)  return launch_dq_t<8>(a, grid_x, shmem, st);
    if (a.KB <= 13) return launch_dq_t<13>(a, grid_x, shmem, st);
    if (a.KB <= 16) return launch_dq_t<16>(a, grid_x, shmem, st);
    return hipErrorInvalidValue;
}

}  // namespace okge
