// fused_tile32_kernel<KB, MODE>: the fused score -> loss -> dCand tile kernel, cut so that TWO workgroups are
// resident per CU (slot sizes up to 256): 32 candidates x 32 batch rows per step, ~60 KB LDS, <= 256 registers.
// Replaces the reference's encode_obj(candidates) + 4 mm + cat + BCEWithLogits / log_softmax+KLDiv forward and the
// mm / sigmoid half of autograd's backward (openkge/model.py:198-229,268-274,455-510; openkge/trainer.py:75-106,234).
//
// One workgroup (4 waves) owns 32 candidate entities: it gathers and drops them out ONCE into LDS (and hands the
// masked rows to dq_kernel through `Cm`), then sweeps the batch's folded query rows in 32-row chunks.
// Wave roles: nbk = w & 1 (16-candidate block), half = w >> 1 (16-row block of the chunk).
//   score product : the wave's 16x16 block X[rows of half][candidates of nbk], two accumulator chains over k,
//                   operands by ds_read_b128 requested one round ahead (v_mfma_f32_16x16x4_f32, exact fp32).
//   loss epilogue : on the accumulator registers; label bit from an LDS bitmask (the tile's positives are cached
//                   in LDS once), BCE-with-logits or KL-on-log-softmax gradient, scaled by 1/normalizer.
//   dC product    : dC[candidates of nbk][all columns] += G^T . Q over the wave's 16 batch rows.
// Register chaining: the MFMA result layout of the score block (lane column = candidate, register i of slot s =
// batch row 4s+i) is exactly the A-operand layout the dC product needs (M index = candidate, contraction slot s,
// step i = batch row 4s+i), so G = dLoss/dX never goes through LDS and no barrier separates the two products.
// G also leaves for dq_kernel straight from registers (one float4 per lane = 4 consecutive batch rows of one
// candidate) as 64x64 transposed blocks  Gt[(T * nJ + J)][n_local (64)][b_local (64)],  T = 64-candidate chunk,
// J = 64-row block: exactly the LDS image dq_kernel wants.
// The two halves' dC partial sums are added in the write-back, which also applies the cached dropout flags.
//
// MODE_SCORE / MODE_STATS (used for slot sizes above 256, where the 64x64 score tile does not fit LDS) stop after
// the score product and write X, or per-(16-candidate block, row) (max, sum-exp) for the KL loss.
//
// Slot sizes above 256 (KB = 32): the two tiles fill 128 KB of LDS, one workgroup per CU.  It runs 8 waves instead of
// 4 (Tile32Cfg::KS = 2): waves w and w + 4 share a score block, split its contraction and the gradient's columns,
// and exchange the partial block through LDS (one extra barrier per chunk) -- 184 registers instead of 492, two
// waves per SIMD: 180 -> 150 us at the DistMult d=512 / N=10000 shape.
//
// fp32 MFMA shares the SIMD's vector issue with VALU on gfx950, so the loop is MFMA cycles + VALU cycles; what the
// second resident workgroup hides is latency (LDS, barriers, HBM), not arithmetic.
#include <cstdio>
#include <cstdlib>

#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

constexpr int NT32 = 32, BC32 = 32;

template <int KB> struct Tile32Cfg {
    static constexpr int LDK = lds_ld(16 * KB);
    static constexpr int NO = 2 * KB;                         // 8-column octets per row
    static constexpr int KEEP_LD = NO < 32 ? 32 : NO;         // keep-flag bytes per row
    // Slot sizes above 256: the two tiles take 128 KB of LDS, so only ONE workgroup fits a CU.  It then runs 8 waves
    // (KS = 2): waves w and w + 4 own the same 16x16 score block, each contracts half of the columns, the partial
    // blocks are exchanged through LDS, both apply the loss epilogue, and each accumulates the candidate gradient for
    // ITS half of the columns -- 64 accumulator registers instead of 128, two waves per SIMD instead of one.
    static constexpr int KS = KB <= 16 ? 1 : 2;
    static constexpr int THREADS = 256 * KS;
    static constexpr int QG = 8 * KS;                         // staging: column groups per row (THREADS / 32 rows)
    static constexpr int WAVES_PER_SIMD = 2;                  // KB <= 16: two workgroups per CU; above: one of 8 waves
};

#ifdef OKGE_STAMPS
// diagnostic build (tools/build_stamps.sh): workgroup placement + per-chunk phase timeline of wave 0
#define TL_STAMP()                                                                                                 \
    do {                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        unsigned long long t_;                                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                  \
        if (a.stamps_dbg && threadIdx.x == 0 && tl_n < 80)                                                         \
            a.stamps_dbg[(size_t)gridDim.x * gridDim.y * 4 + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 80 + tl_n] = t_; \
        ++tl_n;                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
#else
#define TL_STAMP() do { } while (0)
#endif

template <int KB, int MODE>
__global__ __launch_bounds__(Tile32Cfg<KB>::THREADS, Tile32Cfg<KB>::WAVES_PER_SIMD) void fused_tile32_kernel(const FusedArgs a)
{
#ifdef OKGE_STAMPS
    unsigned long long wg_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0)::"memory");
    int tl_n = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Cfg = Tile32Cfg<KB>;
    constexpr int LDK = Cfg::LDK, NO = Cfg::NO, KEEP_LD = Cfg::KEEP_LD;
    constexpr int KS = Cfg::KS, NTHR = Cfg::THREADS, QG = Cfg::QG;
    constexpr int KQ = KB / 4, KR = KB % 4;
    constexpr int KBW = KB / KS, KQW = KQ / KS;          // contraction blocks / gradient column quads of one wave
    static_assert(KS == 1 || (KB % 8 == 0), "the column split needs whole quads of 16-column blocks per wave");
    constexpr int NOIT = (NO + QG - 1) / QG;
    constexpr int NQ = 4 * KB, NQIT = (NQ + QG - 1) / QG;       // float4 per row
    constexpr bool TRAIN = MODE == MODE_TRAIN_BCE || MODE == MODE_TRAIN_KL;
    const int d = a.d;
    float *Cs = reinterpret_cast<float *>(smem);              // [32][LDK]   (end: dC stage of half 1)
    float *Qs = Cs + NT32 * LDK;                              // [32][LDK]   (end: dC stage of half 0)
    uint32_t *ybits2 = reinterpret_cast<uint32_t *>(Qs + BC32 * LDK);    // [2][32] label bits, double-buffered by chunk
    double *red = reinterpret_cast<double *>(ybits2 + 2 * BC32);         // [4 * KS]
    uint8_t *keepb = reinterpret_cast<uint8_t *>(red + 4 * KS);          // [32][KEEP_LD] keep flags of the tile
    uint32_t *posc = reinterpret_cast<uint32_t *>(keepb + NT32 * KEEP_LD);   // [POS_CACHE] (row << 6 | col)
    v4f *xs = reinterpret_cast<v4f *>(posc + POS_CACHE);                 // KS == 2: [2][4][64] partial score blocks

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int ks = KS == 1 ? 0 : w >> 2, wq = w & 3;     // column half, position in the 2 x 2 block grid
    const int nbk = wq & 1, half = wq >> 1;
    const int n0 = blockIdx.x * NT32;
    const int b_begin = blockIdx.y * a.b_per_block;
    const int b_end = min(a.B, b_begin + a.b_per_block);
    const bool vec_ok = (d & 3) == 0;
    const int r8 = tid / QG, q8 = tid % QG;                   // staging role: row r8, column group q8

    // ---- register-staged query chunk (32 rows x 16*KB): thread holds float4 columns q8 + 8*it of row r8 ------
    v4f qreg[NQIT];
    auto fetch_chunk = [&](int b0) {
        const int b = b0 + r8;
        const float *src = a.Q + (size_t)b * a.ldq;
#pragma unroll
        for (int it = 0; it < NQIT; ++it) {
            const int q = min(q8 + QG * it, NQ - 1);
            qreg[it] = (b < b_end) ? *reinterpret_cast<const v4f *>(src + 4 * q) : (v4f){0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch_chunk(b_begin);

    // ---- candidate tile: gather, dropout, LDS; masked copy for dq_kernel ---------------------------------------
    {
        const int n = n0 + r8;
        const bool valid = n < a.N;
        int64_t cid = 0;
        if (valid) cid = checked_row(a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n, a.n_table_rows, q8 ? nullptr : a.id_err);
        const float *row = a.E + cid * d;
        float *cm = (TRAIN && blockIdx.y == 0 && !a.loss_only) ? a.Cm + (size_t)n * (16 * KB) : nullptr;
        v4f v0[NOIT], v1[NOIT];
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = q8 + QG * it, k = 8 * o;
            v0[it] = (v4f){0.f, 0.f, 0.f, 0.f};
            v1[it] = (v4f){0.f, 0.f, 0.f, 0.f};
            if (o < NO && valid && k < d) {
                if (vec_ok) {
                    v0[it] = *reinterpret_cast<const v4f *>(row + k);
                    if (k + 4 < d) v1[it] = *reinterpret_cast<const v4f *>(row + k + 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (k + e < d) v0[it][e] = row[k + e];
                        if (k + 4 + e < d) v1[it][e] = row[k + 4 + e];
                    }
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = q8 + QG * it, k = 8 * o;
            if (o < NO) {
                uint32_t bits = 0xFFu;
                if (a.drop_c.enabled) {
                    bits = (valid && k < d) ? drop_keep8(a.drop_c, (uint32_t)(n + a.cand_col0), o, d) : 0u;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v0[it][e] *= (bits >> e & 1u) ? a.drop_c.scale : 0.f;
                        v1[it][e] *= (bits >> (4 + e) & 1u) ? a.drop_c.scale : 0.f;
                    }
                }
                *reinterpret_cast<v4f *>(Cs + r8 * LDK + k) = v0[it];
                *reinterpret_cast<v4f *>(Cs + r8 * LDK + k + 4) = v1[it];
                if (TRAIN) keepb[r8 * KEEP_LD + o] = (uint8_t)bits;
                if (cm) {
                    *reinterpret_cast<v4f *>(cm + k) = v0[it];
                    *reinterpret_cast<v4f *>(cm + k + 4) = v1[it];
                }
            }
        }
    }
    // positives of this tile: cached in LDS once (a global re-read per chunk costs an L2 round trip each time)
    int pos_lo = 0, pos_hi = 0, pos_cached = 0;
    if (TRAIN) {
        if (tid < 2 * BC32) ybits2[tid] = 0u;
        pos_lo = a.tile_ptr[blockIdx.x];
        pos_hi = a.tile_ptr[blockIdx.x + 1];
        pos_cached = min(pos_hi - pos_lo, POS_CACHE);
        for (int i = tid; i < pos_cached; i += NTHR)
            posc[i] = ((uint32_t)a.pos_row[pos_lo + i] << 6) | (uint32_t)(a.pos_col[pos_lo + i] - a.cand_col0 - n0);
        __syncthreads();
    }

    v4f dc[KBW];                                     // dC[n = 16nbk + 4s + i][k = grad col(kbi, c)], rows of this half
#pragma unroll                                       // (KS == 2: the 16-column blocks KBW*ks .. of the gradient)
    for (int kb = 0; kb < KBW; ++kb) dc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;

    int par = 0;
    for (int b0 = b_begin; b0 < b_end; b0 += BC32, par ^= 1) {
        // ---- phase A: park the prefetched chunk, set label bits, prefetch the next chunk ------------------------
        // (no barrier separates a wave's epilogue from its dC product, so the label bits are double-buffered: this
        //  chunk's buffer was cleared one chunk ago, the previous chunk's buffer is cleared now)
        uint32_t *ybits = ybits2 + par * BC32;
#pragma unroll
        for (int it = 0; it < NQIT; ++it) {
            const int q = q8 + QG * it;
            if (q < NQ) *reinterpret_cast<v4f *>(Qs + r8 * LDK + 4 * q) = qreg[it];
        }
        if (TRAIN) {
            if (tid < BC32) ybits2[(par ^ 1) * BC32 + tid] = 0u;
            for (int i = tid; i < pos_cached; i += NTHR) {
                const uint32_t v = posc[i];
                const int row = (int)(v >> 6) - b0;
                if (row >= 0 && row < BC32) atomicOr(&ybits[row], 1u << (v & 63u));
            }
            for (int p = pos_lo + POS_CACHE + tid; p < pos_hi; p += NTHR) {      // overflow: rare
                const int row = a.pos_row[p] - b0;
                if (row >= 0 && row < BC32) atomicOr(&ybits[row], 1u << (a.pos_col[p] - a.cand_col0 - n0));
            }
        }
        if (b0 + BC32 < b_end) fetch_chunk(b0 + BC32);
        __syncthreads();
        TL_STAMP();   // [0] start of score product

        // ---- score block (rows 16*half + 4s + i, columns 16*nbk + c) -------------------------------------------
        v4f x0 = (v4f){0.f, 0.f, 0.f, 0.f}, x1 = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            // operands are requested TWO rounds ahead: one round is 4 MFMAs = 128 cycles, about one LDS round trip
            const float *qa = Qs + (16 * half + c) * LDK + 4 * s + 16 * KBW * ks;    // KS == 2: this wave's column half
            const float *cb = Cs + (16 * nbk + c) * LDK + 4 * s + 16 * KBW * ks;
            v4f a0 = *reinterpret_cast<const v4f *>(qa), b0v = *reinterpret_cast<const v4f *>(cb);
            v4f a1 = a0, b1v = b0v;
            if (KBW > 1) {
                a1 = *reinterpret_cast<const v4f *>(qa + 16);
                b1v = *reinterpret_cast<const v4f *>(cb + 16);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, KBW > 1 ? 4 : 2, 0);   // rounds 0 and 1 operands
#pragma unroll
            for (int r = 0; r < KBW; ++r) {
                v4f a2 = a1, b2v = b1v;
                if (r + 2 < KBW) {
                    a2 = *reinterpret_cast<const v4f *>(qa + 16 * (r + 2));
                    b2v = *reinterpret_cast<const v4f *>(cb + 16 * (r + 2));
                }
                x0 = mfma16(a0[0], b0v[0], x0);
                x1 = mfma16(a0[1], b0v[1], x1);
                x0 = mfma16(a0[2], b0v[2], x0);
                x1 = mfma16(a0[3], b0v[3], x1);
                a0 = a1; b0v = b1v;
                a1 = a2; b1v = b2v;
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 ds_read (round r+2)
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA   (round r)
            }
        }
        v4f x = x0 + x1;
        if (KS == 2) {                                   // the other column half's partial block, through LDS
            xs[(ks * 4 + wq) * 64 + lane] = x;
            __syncthreads();
            x += xs[((ks ^ 1) * 4 + wq) * 64 + lane];    // a + b == b + a: both waves hold the same bits
        }
        TL_STAMP();   // [1] end of score product

        if (MODE == MODE_SCORE) {
            const int n = n0 + 16 * nbk + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + 16 * half + 4 * s + i;
                if (ks == 0 && b < b_end && n < a.N) a.X[(size_t)b * a.ldx + n] = x[i];
            }
            __syncthreads();
            continue;
        }
        if (MODE == MODE_STATS) {
            // per row: max and sum-exp over this wave's 16 candidates -> stats[(2*tile + nbk)][row]
            const bool nvalid = n0 + 16 * nbk + c < a.N;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m = nvalid ? x[i] : -INFINITY;
                m = row16_max(m);
                float se = nvalid ? __expf(x[i] - m) : 0.f;
                se = row16_sum(se);
                const int b = b0 + 16 * half + 4 * s + i;
                if (ks == 0 && c == 0 && b < b_end)
                    reinterpret_cast<float2 *>(a.stats)[(size_t)(2 * blockIdx.x + nbk) * a.Bpad + b] = make_float2(m, se);
            }
            __syncthreads();
            continue;
        }

        // B operands (query rows) of the dC product's first step: requested now, consumed after the epilogue
        const float *qb = Qs + (16 * half + 4 * s) * LDK + 64 * KQW * ks;       // KS == 2: this wave's gradient columns
        v4f pb[KQW > 0 ? KQW : 1];
        float pr[KR > 0 ? KR : 1];
#pragma unroll
        for (int kq = 0; kq < KQW; ++kq) pb[kq] = *reinterpret_cast<const v4f *>(qb + 64 * kq + 4 * c);
#pragma unroll
        for (int r = 0; r < KR; ++r) pr[r] = qb[64 * KQ + 16 * r + c];          // (KR > 0 only with KS == 1)

        // ---- loss epilogue: G = dLoss/dX / normalizer, kept in registers --------------------------------------
        v4f g4;
        {
            constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
            const int nl = 16 * nbk + c;
            const bool nvalid = n0 + nl < a.N;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int bl = 16 * half + 4 * s + i;
                const float xv = x[i];
                const bool valid = nvalid && (b0 + bl < b_end);
                const bool pos = (ybits[bl] >> nl) & 1u;
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
                    // KLDiv(sum)(log_softmax(x), y), y in {0,1} unnormalised (trainer.py:99-101):
                    // loss = -sum_pos log_softmax; d/dx = softmax * sum_n y - y
                    const int b = min(b0 + bl, a.B - 1);
                    const float lsm = xv - a.row_lse[b];
                    l = pos ? -lsm : 0.f;
                    g = __builtin_amdgcn_exp2f(lsm * LOG2E) * a.row_ysum[b] - (pos ? 1.f : 0.f);
                }
                lsum += (valid && ks == 0) ? l : 0.f;       // the partner wave sees the same block: count it once
                g4[i] = valid ? g * a.inv_norm : 0.f;
            }
        }
        TL_STAMP();   // [2] start of dC product
        if (!a.loss_only) {
            // ---- G block -> HBM for dq_kernel: Gt[T][J][n_local][b_local], 4 consecutive batch rows per lane ------
            if (ks == 0) {
                const int t = blockIdx.x, j = b0 >> 5;
                const size_t blk = (size_t)(t >> 1) * (a.Bpad >> 6) + (j >> 1);
                const int nl64 = 32 * (t & 1) + 16 * nbk + c, bl64 = 32 * (j & 1) + 16 * half + 4 * s;
                *reinterpret_cast<v4f *>(a.G + blk * 4096 + nl64 * 64 + bl64) = g4;
            }
            // ---- dC += G^T . Q over this wave's 16 batch rows: A operand straight from g4 ------------------------
            // slot s, step t  <->  batch row 16*half + 4s + t ; A = G[row][n = 16nbk + c] = g4[t], B = Q[row][columns]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float av = g4[t];
                v4f nb[KQW > 0 ? KQW : 1];
                float nr[KR > 0 ? KR : 1];
                if (t + 1 < 4) {
                    const float *brow = qb + (t + 1) * LDK;
#pragma unroll
                    for (int kq = 0; kq < KQW; ++kq) nb[kq] = *reinterpret_cast<const v4f *>(brow + 64 * kq + 4 * c);
#pragma unroll
                    for (int r = 0; r < KR; ++r) nr[r] = brow[64 * KQ + 16 * r + c];
                }
#pragma unroll
                for (int kq = 0; kq < KQW; ++kq) {
                    dc[4 * kq + 0] = mfma16(av, pb[kq][0], dc[4 * kq + 0]);
                    dc[4 * kq + 1] = mfma16(av, pb[kq][1], dc[4 * kq + 1]);
                    dc[4 * kq + 2] = mfma16(av, pb[kq][2], dc[4 * kq + 2]);
                    dc[4 * kq + 3] = mfma16(av, pb[kq][3], dc[4 * kq + 3]);
                }
#pragma unroll
                for (int r = 0; r < KR; ++r) dc[4 * KQ + r] = mfma16(av, pr[r], dc[4 * KQ + r]);
                if (t + 1 < 4) {
#pragma unroll
                    for (int kq = 0; kq < KQW; ++kq) pb[kq] = nb[kq];
#pragma unroll
                    for (int r = 0; r < KR; ++r) pr[r] = nr[r];
                    __builtin_amdgcn_sched_group_barrier(0x100, KQW + KR, 1);  // next step's ds_reads first
                }
                __builtin_amdgcn_sched_group_barrier(0x008, KBW, 1);           // then this step's MFMAs
            }
        }
        TL_STAMP();   // [3] end of dC product
        __syncthreads();
    }

    if (TRAIN) {
        // ---- write-back: both halves stage their partial dC (half 0 -> Qs, half 1 -> Cs), then rows are summed,
        //      masked with the cached dropout flags and added into dE ------------------------------------------------
        if (!a.loss_only) {
            float *stage = (half == 0 ? Qs : Cs) + 64 * KQW * ks;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *srow = stage + (16 * nbk + 4 * s + i) * LDK;
#pragma unroll
                for (int kq = 0; kq < KQW; ++kq)
                    *reinterpret_cast<v4f *>(srow + 64 * kq + 4 * c) =
                        (v4f){dc[4 * kq][i], dc[4 * kq + 1][i], dc[4 * kq + 2][i], dc[4 * kq + 3][i]};
#pragma unroll
                for (int r = 0; r < KR; ++r) srow[64 * KQ + 16 * r + c] = dc[4 * KQ + r][i];
            }
        }
        {
            const double ls = wave_sum((double)lsum);
            if (lane == 0) red[w] = ls;
        }
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
#pragma unroll
            for (int i = 0; i < 4 * KS; ++i) tot += red[i];
            a.loss_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
        }
        const int n = n0 + r8;
        if (!a.loss_only && n < a.N) {
            const int64_t cid = checked_row(a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n, a.n_table_rows, nullptr);
            float *drow = a.dE + cid * d;
            const bool exclusive = gridDim.y == 1 && a.cand_exclusive;    // one workgroup per entity row: plain stores
            // batch split over blockIdx.y: every workgroup stores ITS partial rows into a slab (plain 16-byte stores);
            // dc_reduce_kernel sums the slabs into dE.  (Float atomics from gridDim.y workgroups per row cost more than
            // the whole tile sweep: 3.4 M atomics = ~120 us at the 8-rank FB shape.)
            float *srow = gridDim.y > 1 ? a.dC_slab + ((size_t)blockIdx.y * gridDim.x * NT32 + n) * (16 * KB) : nullptr;
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = q8 + QG * it, k = 8 * o;
                if (o >= NO || k >= d) continue;
                v4f v[2];
                v[0] = *reinterpret_cast<const v4f *>(Qs + r8 * LDK + k) + *reinterpret_cast<const v4f *>(Cs + r8 * LDK + k);
                v[1] = *reinterpret_cast<const v4f *>(Qs + r8 * LDK + k + 4) +
                       *reinterpret_cast<const v4f *>(Cs + r8 * LDK + k + 4);
                if (a.drop_c.enabled) {
                    const uint32_t bits = keepb[r8 * KEEP_LD + o];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[0][e] *= (bits >> e & 1u) ? a.drop_c.scale : 0.f;
                        v[1][e] *= (bits >> (4 + e) & 1u) ? a.drop_c.scale : 0.f;
                    }
                }
                if (srow) {                                   // 16*KB columns per slab row: k + 8 <= 16*KB always
                    *reinterpret_cast<v4f *>(srow + k) = v[0];
                    *reinterpret_cast<v4f *>(srow + k + 4) = v[1];
                    continue;
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int kk = k + 4 * hh;
                    if (kk >= d) continue;
                    if (exclusive && vec_ok) {
                        v4f o4 = v[hh];
                        if (!a.grads_zero) o4 += *reinterpret_cast<const v4f *>(drow + kk);
                        *reinterpret_cast<v4f *>(drow + kk) = o4;
                    } else if (exclusive) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (kk + e < d) drow[kk + e] = a.grads_zero ? v[hh][e] : drow[kk + e] + v[hh][e];
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (kk + e < d) atomicAdd(drow + kk + e, v[hh][e]);
                    }
                }
            }
        }
    }
#ifdef OKGE_STAMPS
    if (a.stamps_dbg && tid == 0) {
        unsigned long long wg_t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t1)::"memory");
        unsigned long long *dst = a.stamps_dbg + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4;
        dst[0] = wg_t0;
        dst[1] = wg_t1;
        dst[2] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
        dst[3] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    }
#endif
}

// ---- launcher ---------------------------------------------------------------------------------------------------
template <int KB>
static size_t shmem32()
{
    using Cfg = Tile32Cfg<KB>;
    return (size_t)(NT32 + BC32) * Cfg::LDK * sizeof(float) + 2 * BC32 * sizeof(uint32_t) + 4 * Cfg::KS * sizeof(double) +
           NT32 * Cfg::KEEP_LD + POS_CACHE * sizeof(uint32_t) + (Cfg::KS == 2 ? 2 * 4 * 64 * sizeof(v4f) : 0);
}

template <int KB, int MODE>
static hipError_t launch32_t(const FusedArgs &a, dim3 grid, hipStream_t st)
{
    auto k = fused_tile32_kernel<KB, MODE>;
    const size_t shmem = shmem32<KB>();
    static LdsOptIn lds_opt_in;
    if (hipError_t e = ensure_dynamic_lds(lds_opt_in, reinterpret_cast<const void *>(k), shmem); e != hipSuccess) return e;
    if (std::getenv("OKGE_DEBUG")) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k), Tile32Cfg<KB>::THREADS, shmem);
        fprintf(stderr, "[okge] fused_tile32<%d,%d>: occupancy %d blocks/CU (err %d), dyn LDS %zu, grid %ux%u\n", KB, MODE,
                nb, (int)e, shmem, grid.x, grid.y);
    }
    hipLaunchKernelGGL(k, grid, dim3(Tile32Cfg<KB>::THREADS), shmem, st, a);
    return hipGetLastError();
}

template <int KB>
static hipError_t launch32_m(int mode, const FusedArgs &a, dim3 grid, hipStream_t st)
{
    switch (mode) {
        case MODE_TRAIN_BCE: return launch32_t<KB, MODE_TRAIN_BCE>(a, grid, st);
        case MODE_TRAIN_KL:  return launch32_t<KB, MODE_TRAIN_KL>(a, grid, st);
        case MODE_SCORE:     return launch32_t<KB, MODE_SCORE>(a, grid, st);
        default:             return launch32_t<KB, MODE_STATS>(a, grid, st);
    }
}

// grid_x = number of 32-candidate tiles (even, so every 64-wide chunk dq_kernel reads is written)
hipError_t launch_fused32(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st)
{
    const dim3 grid(grid_x, grid_y);
    switch (a.KB) {
        case 4:  return launch32_m<4>(mode, a, grid, st);
        case 8:  return launch32_m<8>(mode, a, grid, st);
        case 13: return launch32_m<13>(mode, a, grid, st);
        case 16: return launch32_m<16>(mode, a, grid, st);
        case 32: return launch32_m<32>(mode, a, grid, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace okge
