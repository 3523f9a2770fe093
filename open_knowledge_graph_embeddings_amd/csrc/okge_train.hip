// Fused prefix-scoring kernels for gfx950 (MI355X).
//
//   fused_tile_kernel<KB, MODE>
//     One workgroup owns a tile of NT=64 candidate entities (gathered + dropped-out once into LDS) and
//     sweeps the batch's folded query rows in chunks of BC=64:
//        X  = Q_chunk . C_tile^T            (v_mfma_f32_16x16x4_f32, exact fp32)       [all modes]
//        G  = dLoss/dX / normalizer, loss   (BCE / KL epilogue in registers)           [train]
//        dC += G^T . Q_chunk                (accumulators stay in registers)           [train]
//     and leaves G (B x N, row-major) in HBM for the query-gradient kernel.  Replaces the reference's
//     encode_obj(candidates) + 4 mm + cat + BCEWithLogits/log_softmax+KLDiv forward and the mm/sigmoid
//     half of autograd's backward (openkge/model.py:198-229,268-274; openkge/trainer.py:75-106,234).
//     MODE_SCORE writes X (evaluation / *_prefix_score); MODE_STATS writes per-row (max, sum-exp)
//     partials for the KL loss' log_softmax.
//
//   dq_kernel<KB>
//     dQ = G . C : one workgroup per (64-row batch block, candidate range); partial slabs are summed by
//     prefix_backward_kernel (okge_misc.hip).
//
// KB = padded slot size / 16 is a compile-time constant so every operand read is unconditional.
//
// Operand feeding.  v_mfma_f32_16x16x4_f32 takes ONE fp32 of A and of B per lane (lane l: row/col l&15,
// k-slot l>>4).  Which actual contraction index a (step, slot) pair means is free as long as A and B agree:
//   * score product (contract over k): slot s, step j of round r  <->  k = 16r + 4s + j, so each lane reads
//     4 consecutive k of its row with one ds_read_b128 and feeds 4 MFMAs;
//   * gradient products (contract over batch rows / candidates): slot s, step t <-> row 16s + t.  The B
//     operand (Q or C rows, LDK = 4*odd floats) is read with ds_read_b128 as 4 consecutive COLUMNS of that
//     row; the 4 values feed 4 MFMAs whose output blocks hold the permuted columns 64*kq + 4*c + e
//     (e = 0..3).  Rows 16 apart start 16*LDK*4 bytes = a multiple of 256 B apart, so the four slots of a
//     ds_read_b128 lane group hit 16 distinct 16-byte LDS slots (conflict-free).  Leftover 16-column blocks
//     (KB mod 4) use ds_read_b32 with natural columns.
#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

// ---- candidate tile: gather rows of E, apply dropout, park in LDS as Cs[NT][LDK] (zero padded) -----
// Thread (tid>>3, tid&7) handles rows tid>>3 and 32 + tid>>3, octets (8 columns) tid&7, 8 + tid&7, ...
// keepb (LDS, [NT][32] bytes) caches the keep flags for the dC epilogue; Cm (global [NT][16*KB]) receives the
// masked tile for dq_kernel.  Either may be nullptr.
template <int KB>
__device__ __forceinline__ void load_cand_tile(float *Cs, uint8_t *keepb, float *Cm, const float *__restrict__ E,
                                               int d, const int32_t *__restrict__ cand_ids, int cand_first, int N,
                                               int n0, const DropDev &drop, bool vec_ok, int tid)
{
    constexpr int LDK = lds_ld(16 * KB), NO = 2 * KB, NOIT = (NO + 7) / 8;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = (tid >> 3) + 32 * pass;
        const int n = n0 + r;
        const bool valid = n < N;
        int64_t cid = 0;
        if (valid) cid = cand_ids ? (int64_t)cand_ids[n] : (int64_t)cand_first + n;
        const float *row = E + cid * d;
        float4 v0[NOIT], v1[NOIT];
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = (tid & 7) + 8 * it, k = 8 * o;
            v0[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            v1[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (o < NO && valid && k < d) {
                if (vec_ok) {
                    v0[it] = *reinterpret_cast<const float4 *>(row + k);
                    if (k + 4 < d) v1[it] = *reinterpret_cast<const float4 *>(row + k + 4);
                } else {
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = (k + e < d) ? row[k + e] : 0.f;
                    v0[it] = make_float4(t[0], t[1], t[2], t[3]);
                    v1[it] = make_float4(t[4], t[5], t[6], t[7]);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = (tid & 7) + 8 * it, k = 8 * o;
            if (o < NO) {
                uint32_t bits = 0xFFu;
                if (drop.enabled) {
                    bits = (valid && k < d) ? drop_keep8(drop, (uint32_t)n, o, d) : 0u;
                    apply_keep4(v0[it], bits & 15u, drop.scale);
                    apply_keep4(v1[it], bits >> 4, drop.scale);
                }
                *reinterpret_cast<float4 *>(Cs + r * LDK + k) = v0[it];
                *reinterpret_cast<float4 *>(Cs + r * LDK + k + 4) = v1[it];
                if (keepb) keepb[r * 32 + o] = (uint8_t)bits;
                if (Cm) {
                    *reinterpret_cast<float4 *>(Cm + (size_t)r * (16 * KB) + k) = v0[it];
                    *reinterpret_cast<float4 *>(Cm + (size_t)r * (16 * KB) + k + 4) = v1[it];
                }
            }
        }
    }
}

// acc[kbi] += A^T-style gradient product over 64 contraction rows held in LDS.
//   arow : &A[0][out_row(lane)] element for contraction row 0, slot offset included by caller = base + (16*s)*lda
//   brow : &B[16*s][0] + lane column offset handled here
// Contraction row of (slot s, step t) is 16*s + t.  A is read 4 steps at a time when A_VEC (A stored with
// the contraction index contiguous), else one ds_read_b32 per step.
template <int KB, bool A_VEC>
__device__ __forceinline__ void grad_product(v4f (&acc)[KB], const float *a_base, int lda, const float *b_base,
                                             int c)
{
    constexpr int LDK = lds_ld(16 * KB);
    constexpr int KQ = KB / 4, KR = KB % 4;
    // b_base points at B[16*s][0]
#pragma unroll
    for (int t4 = 0; t4 < 4; ++t4) {
        float av[4];
        if (A_VEC) {
            const float4 a4 = *reinterpret_cast<const float4 *>(a_base + 4 * t4);   // A[out][16s + 4t4 ..+3]
            av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = a_base[(4 * t4 + j) * lda];          // A[16s + t][out]
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float *brow = b_base + (4 * t4 + j) * LDK;
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const float4 b4 = *reinterpret_cast<const float4 *>(brow + 64 * kq + 4 * c);
                acc[4 * kq + 0] = mfma16(av[j], b4.x, acc[4 * kq + 0]);
                acc[4 * kq + 1] = mfma16(av[j], b4.y, acc[4 * kq + 1]);
                acc[4 * kq + 2] = mfma16(av[j], b4.z, acc[4 * kq + 2]);
                acc[4 * kq + 3] = mfma16(av[j], b4.w, acc[4 * kq + 3]);
            }
#pragma unroll
            for (int r = 0; r < KR; ++r)
                acc[4 * KQ + r] = mfma16(av[j], brow[64 * KQ + 16 * r + c], acc[4 * KQ + r]);
        }
    }
}

// Column held by lane column-index c of accumulator block kbi (see "Operand feeding" above).
template <int KB>
__device__ __forceinline__ int grad_col(int kbi, int c)
{
    constexpr int KQ = KB / 4;
    return kbi < 4 * KQ ? 64 * (kbi >> 2) + 4 * c + (kbi & 3) : 64 * KQ + 16 * (kbi - 4 * KQ) + c;
}

#ifdef OKGE_STAMPS
// Diagnostic build only (make stamps): per-phase shader-cycle totals per wave -> a.stamps_dbg.
#define STAMP(var)                                                                                  \
    do {                                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                \
        __builtin_amdgcn_sched_barrier(0);                                                          \
    } while (0)
#define STAMP_ACC(i, t_new, t_old) (stamp_acc[i] += (t_new) - (t_old))
#else
#define STAMP(var) do { } while (0)
#define STAMP_ACC(i, t_new, t_old) do { } while (0)
#endif

template <int KB, int MODE>
__global__ __launch_bounds__(FUSED_THREADS) void fused_tile_kernel(const FusedArgs a)
{
#ifdef OKGE_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0, t6 = 0;
    STAMP(t0);
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDK = lds_ld(16 * KB);
    constexpr int KQ = KB / 4, KR = KB % 4;
    const int d = a.d;
    float *Cs = reinterpret_cast<float *>(smem);              // [NT][LDK]
    float *Qs = Cs + NT * LDK;                                // [BC][LDK]   (end: dC staging)
    float *Gs = Qs + BC * LDK;                                // [BC][LDG]   G tile (score mode: X tile)
    uint32_t *ybits = reinterpret_cast<uint32_t *>(Gs + BC * LDG);   // [BC][2] label bits of the chunk
    double *red = reinterpret_cast<double *>(ybits + BC * 2);        // [4]
    uint8_t *keepb = reinterpret_cast<uint8_t *>(red + 4);           // [NT][32] dropout keep flags of the tile
    constexpr bool TRAIN = MODE == MODE_TRAIN_BCE || MODE == MODE_TRAIN_KL;

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int n0 = blockIdx.x * NT;
    const int b_begin = blockIdx.y * a.b_per_block;
    const int b_end = min(a.B, b_begin + a.b_per_block);
    const bool vec_ok = (d & 3) == 0;

    // register-staged query chunk: thread (row r = tid/4, quarter q = tid%4) holds KB float4 of that row
    const int qr = tid >> 2, qq = tid & 3;
    float4 qreg[KB];
    auto fetch_chunk = [&](int b0) {
        const int b = b0 + qr;
        const float *src = a.Q + (size_t)b * a.ldq + 4 * qq;
#pragma unroll
        for (int it = 0; it < KB; ++it)
            qreg[it] = (b < b_end) ? *reinterpret_cast<const float4 *>(src + 16 * it)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    fetch_chunk(b_begin);

    load_cand_tile<KB>(Cs, TRAIN ? keepb : nullptr,
                       (TRAIN && blockIdx.y == 0) ? a.Cm + (size_t)n0 * (16 * KB) : nullptr, a.E, d, a.cand_ids,
                       a.cand_first, a.N, n0, a.drop_c, vec_ok, tid);
    if (tid < BC * 2) ybits[tid] = 0u;

    int pos_lo = 0, pos_hi = 0;
    if (TRAIN) {            // positives of this tile: [tile_ptr[t], tile_ptr[t+1]) (built by encode_queries_kernel)
        pos_lo = a.tile_ptr[blockIdx.x];
        pos_hi = a.tile_ptr[blockIdx.x + 1];
    }

    v4f dc[KB];                                      // dC[n = 16w + 4s + i][k = grad_col(kbi, c)]
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) dc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    STAMP(t1);
    STAMP_ACC(0, t1, t0);                             // [0] prologue: cand tile gather + dropout

    for (int b0 = b_begin; b0 < b_end; b0 += BC) {
        STAMP(t1);
        // ---- phase A: park the prefetched chunk in LDS, set label bits, prefetch the next chunk -----
#pragma unroll
        for (int it = 0; it < KB; ++it) *reinterpret_cast<float4 *>(Qs + qr * LDK + 4 * qq + 16 * it) = qreg[it];
        if (TRAIN) {
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
        STAMP(t2);
        STAMP_ACC(1, t2, t1);                         // [1] phase A: Q chunk -> LDS, label bits, barrier

        // ---- phase B: X = Q_chunk . C^T ; wave w owns rows 16w..16w+15, all four 16-wide n blocks ---
        v4f x[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) x[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            const float *qa = Qs + (16 * w + c) * LDK + 4 * s;
            const float *cb = Cs + c * LDK + 4 * s;
#pragma unroll
            for (int r = 0; r < KB; ++r) {
                const float4 av = *reinterpret_cast<const float4 *>(qa + 16 * r);
                float4 bv[4];
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) bv[nb] = *reinterpret_cast<const float4 *>(cb + 16 * nb * LDK + 16 * r);
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
        STAMP(t3);
        STAMP_ACC(2, t3, t2);                         // [2] score product

        if (MODE == MODE_SCORE) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) Gs[(16 * w + 4 * s + i) * LDG + 16 * nb + c] = x[nb][i];
            __syncthreads();
            for (int idx = tid; idx < BC * 16; idx += FUSED_THREADS) {
                const int r = idx >> 4, c4 = idx & 15;
                const int b = b0 + r, n = n0 + 4 * c4;
                if (b < b_end) {
                    const float4 v = *reinterpret_cast<const float4 *>(Gs + r * LDG + 4 * c4);
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

        // ---- train epilogue: G = dLoss/dX / normalizer, loss; G tile to LDS ---------------------------
        {
            uint32_t yw[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint2 t = *reinterpret_cast<const uint2 *>(ybits + (16 * w + 4 * s + i) * 2);
                yw[i][0] = t.x; yw[i][1] = t.y;
            }
            float lse[4] = {0.f, 0.f, 0.f, 0.f}, ysum[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == MODE_TRAIN_KL) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = min(b0 + 16 * w + 4 * s + i, a.B - 1);
                    lse[i] = a.row_lse[b];
                    ysum[i] = a.row_ysum[b];
                }
            }
            constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int nl = 16 * nb + c;
                const bool nvalid = n0 + nl < a.N;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = x[nb][i];
                    const bool valid = nvalid && (b0 + 16 * w + 4 * s + i < b_end);
                    const bool pos = (yw[i][nl >> 5] >> (nl & 31)) & 1u;
                    float g, l;
                    if (MODE == MODE_TRAIN_BCE) {
                        // BCEWithLogits: max(x,0) - x*y + log1p(exp(-|x|)); d/dx = sigmoid(x) - y
                        // v_exp_f32 / v_rcp_f32 / v_log_f32 (1 ulp each); 1 + e is in (1, 2]
                        const float y = pos ? a.y_pos : a.y_neg;
                        const float e = __builtin_amdgcn_exp2f(-fabsf(xv) * LOG2E);
                        const float ope = 1.f + e;
                        const float rcp = __builtin_amdgcn_rcpf(ope);
                        const float sig = xv >= 0.f ? rcp : e * rcp;
                        l = fmaxf(xv, 0.f) - xv * y + __builtin_amdgcn_logf(ope) * LN2;
                        g = sig - y;
                    } else {
                        // KLDiv(sum)(log_softmax(x), y), y in {0,1} unnormalised:
                        // loss = -sum_pos log_softmax; d/dx = softmax * sum_n y - y
                        const float lsm = xv - lse[i];
                        l = pos ? -lsm : 0.f;
                        g = __builtin_amdgcn_exp2f(lsm * LOG2E) * ysum[i] - (pos ? 1.f : 0.f);
                    }
                    lsum += valid ? l : 0.f;
                    Gs[(16 * w + 4 * s + i) * LDG + nl] = valid ? g * a.inv_norm : 0.f;
                }
            }
        }
        STAMP(t4);
        STAMP_ACC(3, t4, t3);                         // [3] loss epilogue (VALU) + G tile to LDS
        __syncthreads();
        STAMP(t5);
        STAMP_ACC(4, t5, t4);                         // [4] barrier after epilogue

        // ---- phase C: G tile -> HBM (for dq_kernel); dC += G^T . Q_chunk ----------------------------
        for (int idx = tid; idx < BC * 16; idx += FUSED_THREADS) {
            const int r = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4 *>(a.G + (size_t)(b0 + r) * a.ldg + n0 + 4 * c4) =
                *reinterpret_cast<const float4 *>(Gs + r * LDG + 4 * c4);
        }
        STAMP(t6);
        STAMP_ACC(5, t6, t5);                         // [5] G tile LDS -> HBM
        // A[i = n][slot s, step t] = G[b = 16s + t][n = 16w + c] ; B[slot][k] = Q[b = 16s + t][k]
        grad_product<KB, false>(dc, Gs + 16 * s * LDG + 16 * w + c, LDG, Qs + 16 * s * LDK, c);
        STAMP(t1);
        STAMP_ACC(6, t1, t6);                         // [6] dC product
        if (tid < BC * 2) ybits[tid] = 0u;
        __syncthreads();
    }

    if (!TRAIN) return;
    STAMP(t1);

    // ---- dC epilogue: stage through LDS, apply the candidates' dropout mask, add into dE rows --------
    float *stage = Qs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *srow = stage + (16 * w + 4 * s + i) * LDK;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
            *reinterpret_cast<float4 *>(srow + 64 * kq + 4 * c) =
                make_float4(dc[4 * kq][i], dc[4 * kq + 1][i], dc[4 * kq + 2][i], dc[4 * kq + 3][i]);
#pragma unroll
        for (int r = 0; r < KR; ++r) srow[64 * KQ + 16 * r + c] = dc[4 * KQ + r][i];
    }
    {
        const double ls = wave_sum((double)lsum);
        if (lane == 0) red[w] = ls;
    }
    __syncthreads();
    if (tid == 0) a.loss_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    {
        constexpr int NO = 2 * KB, NOIT = (NO + 7) / 8;
        const bool exclusive = gridDim.y == 1;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = (tid >> 3) + 32 * pass;
            const int n = n0 + r;
            if (n >= a.N) continue;
            const int64_t cid = a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n;
            float *drow = a.dE + cid * d;
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = (tid & 7) + 8 * it, k = 8 * o;
                if (o >= NO || k >= d) continue;
                float4 v[2];
                v[0] = *reinterpret_cast<const float4 *>(stage + r * LDK + k);
                v[1] = *reinterpret_cast<const float4 *>(stage + r * LDK + k + 4);
                if (a.drop_c.enabled) {
                    const uint32_t bits = keepb[r * 32 + o];
                    apply_keep4(v[0], bits & 15u, a.drop_c.scale);
                    apply_keep4(v[1], bits >> 4, a.drop_c.scale);
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kk = k + 4 * hh;
                    if (kk >= d) continue;
                    if (exclusive && vec_ok) {
                        float4 o4 = v[hh];
                        if (!a.grads_zero) {
                            const float4 old = *reinterpret_cast<const float4 *>(drow + kk);
                            o4.x += old.x; o4.y += old.y; o4.z += old.z; o4.w += old.w;
                        }
                        *reinterpret_cast<float4 *>(drow + kk) = o4;
                    } else if (exclusive) {
                        drow[kk] += v[hh].x;
                        if (kk + 1 < d) drow[kk + 1] += v[hh].y;
                        if (kk + 2 < d) drow[kk + 2] += v[hh].z;
                        if (kk + 3 < d) drow[kk + 3] += v[hh].w;
                    } else {
                        atomicAdd(drow + kk, v[hh].x);
                        if (kk + 1 < d) atomicAdd(drow + kk + 1, v[hh].y);
                        if (kk + 2 < d) atomicAdd(drow + kk + 2, v[hh].z);
                        if (kk + 3 < d) atomicAdd(drow + kk + 3, v[hh].w);
                    }
                }
            }
        }
    }
#ifdef OKGE_STAMPS
    STAMP(t2);
    STAMP_ACC(7, t2, t1);                             // [7] dC epilogue (stage, dropout, add into dE)
    if (a.stamps_dbg && lane == 0) {
        unsigned long long *dst = a.stamps_dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 10;
        for (int i = 0; i < 8; ++i) dst[i] = stamp_acc[i];
        dst[8] = t2 - t0;
        dst[9] = t0;
    }
#endif
}

// ---- dQ = G . C over a candidate range -------------------------------------------------------------
template <int KB>
__global__ __launch_bounds__(FUSED_THREADS) void dq_kernel(const DqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDK = lds_ld(16 * KB);
    constexpr int KQ = KB / 4, KR = KB % 4;
    float *Cs = reinterpret_cast<float *>(smem);      // [NT][LDK]
    float *Gs = Cs + NT * LDK;                        // [BC (b)][LDG] : G tile, 64 candidates wide

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int split = blockIdx.x % a.nsplit, bblk = blockIdx.x / a.nsplit;
    const int b0 = bblk * BC;
    const int nchunks = (a.N + NT - 1) / NT;
    const int ch_lo = (int)((int64_t)split * nchunks / a.nsplit);
    const int ch_hi = (int)((int64_t)(split + 1) * nchunks / a.nsplit);

    v4f acc[KB];                                      // dQ[b = b0 + 16w + 4s + i][k = grad_col(kbi, c)]
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};

    // register-staged prefetch: the next chunk's G tile and masked candidate tile are in flight during the MFMAs
    constexpr int NO = 2 * KB, NOIT = (NO + 7) / 8;
    float4 gv[4], cv0[2][NOIT], cv1[2][NOIT];
    auto prefetch = [&](int ch) {
        const int n0 = ch * NT;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * FUSED_THREADS, r = idx >> 4, c4 = idx & 15;
            gv[it] = *reinterpret_cast<const float4 *>(a.G + (size_t)(b0 + r) * a.ldg + n0 + 4 * c4);
        }
        const float *cm = a.Cm + (size_t)n0 * (16 * KB);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = (tid >> 3) + 32 * pass;
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = (tid & 7) + 8 * it;
                if (o < NO) {
                    cv0[pass][it] = *reinterpret_cast<const float4 *>(cm + (size_t)r * (16 * KB) + 8 * o);
                    cv1[pass][it] = *reinterpret_cast<const float4 *>(cm + (size_t)r * (16 * KB) + 8 * o + 4);
                }
            }
        }
    };
    if (ch_lo < ch_hi) prefetch(ch_lo);
    for (int ch = ch_lo; ch < ch_hi; ++ch) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * FUSED_THREADS, r = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4 *>(Gs + r * LDG + 4 * c4) = gv[it];
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = (tid >> 3) + 32 * pass;
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = (tid & 7) + 8 * it;
                if (o < NO) {
                    *reinterpret_cast<float4 *>(Cs + r * LDK + 8 * o) = cv0[pass][it];
                    *reinterpret_cast<float4 *>(Cs + r * LDK + 8 * o + 4) = cv1[pass][it];
                }
            }
        }
        __syncthreads();
        if (ch + 1 < ch_hi) prefetch(ch + 1);
        // A[i = b][slot s, step t] = G[b = 16w + c][n = 16s + t] ; B[slot][k] = C[n = 16s + t][k]
        grad_product<KB, true>(acc, Gs + (16 * w + c) * LDG + 16 * s, LDG, Cs + 16 * s * LDK, c);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *dst = a.slab + ((size_t)split * a.Bpad + b0 + 16 * w + 4 * s + i) * a.ldq;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
            *reinterpret_cast<float4 *>(dst + 64 * kq + 4 * c) =
                make_float4(acc[4 * kq][i], acc[4 * kq + 1][i], acc[4 * kq + 2][i], acc[4 * kq + 3][i]);
#pragma unroll
        for (int r = 0; r < KR; ++r) dst[64 * KQ + 16 * r + c] = acc[4 * KQ + r][i];
    }
}

// ---- host-side launchers -----------------------------------------------------------------------------
template <int KB, int MODE>
static hipError_t launch_fused_t(const FusedArgs &a, dim3 grid, size_t shmem, hipStream_t st)
{
    auto k = fused_tile_kernel<KB, MODE>;
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

template <int KB>
static hipError_t launch_fused_m(int mode, const FusedArgs &a, dim3 grid, size_t shmem, hipStream_t st)
{
    switch (mode) {
        case MODE_TRAIN_BCE: return launch_fused_t<KB, MODE_TRAIN_BCE>(a, grid, shmem, st);
        case MODE_TRAIN_KL:  return launch_fused_t<KB, MODE_TRAIN_KL>(a, grid, shmem, st);
        case MODE_SCORE:     return launch_fused_t<KB, MODE_SCORE>(a, grid, shmem, st);
        default:             return launch_fused_t<KB, MODE_STATS>(a, grid, shmem, st);
    }
}

size_t fused_shmem_bytes(int LDK)
{
    return (size_t)(NT + BC) * LDK * sizeof(float) + (size_t)BC * LDG * sizeof(float) + BC * 2 * sizeof(uint32_t) +
           4 * sizeof(double) + NT * 32;
}

size_t dq_shmem_bytes(int LDK)
{
    return (size_t)NT * LDK * sizeof(float) + (size_t)BC * LDG * sizeof(float);
}

hipError_t launch_fused(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st)
{
    const dim3 grid(grid_x, grid_y);
    const size_t shmem = fused_shmem_bytes(a.LDK);
    switch (a.KB) {
        case 4:  return launch_fused_m<4>(mode, a, grid, shmem, st);
        case 8:  return launch_fused_m<8>(mode, a, grid, shmem, st);
        case 13: return launch_fused_m<13>(mode, a, grid, shmem, st);
        case 16: return launch_fused_m<16>(mode, a, grid, shmem, st);
        default: return hipErrorInvalidValue;
    }
}

template <int KB>
static hipError_t launch_dq_t(const DqArgs &a, int grid_x, size_t shmem, hipStream_t st)
{
    auto k = dq_kernel<KB>;
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
    switch (a.KB) {
        case 4:  return launch_dq_t<4>(a, grid_x, shmem, st);
        case 8:  return launch_dq_t<8>(a, grid_x, shmem, st);
        case 13: return launch_dq_t<13>(a, grid_x, shmem, st);
        case 16: return launch_dq_t<16>(a, grid_x, shmem, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace okge
