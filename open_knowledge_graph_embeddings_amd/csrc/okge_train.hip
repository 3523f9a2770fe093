// Prefix-scoring kernels for gfx950 (MI355X), 64x64 cut.
//
//   fused_tile_kernel<KB, MODE>   MODE_SCORE | MODE_STATS | MODE_COUNT
//     One workgroup owns a tile of NT=64 candidate entities (gathered + dropped-out once into LDS) and sweeps the
//     batch's folded query rows in chunks of BC=64:  X = Q_chunk . C_tile^T  (v_mfma_f32_16x16x4_f32, exact fp32).
//     MODE_SCORE writes X (evaluation / *_prefix_score: openkge/model.py:52-74,198-229,268-274);
//     MODE_STATS writes per-row (max, sum-exp) partials for the KL loss' log_softmax (openkge/trainer.py:99-100).
//     MODE_COUNT (fused evaluation) compares X in registers with each row's true answer scores and adds the tile's
//     {#greater, #equal} to per-group counters: the rank rule of dataset.py:436-446 without the (B, N) score block.
//     The training step (score -> loss -> dCand) is fused_tile32_kernel in okge_train32.hip.
//
//   dq_kernel<KB>
//     dQ = G . C : one workgroup per (64-row batch block, candidate range); partial slabs are summed by
//     prefix_backward_kernel (okge_misc.hip).  G arrives as 64x64 transposed blocks G^T[n][b] (16 KB each,
//     written by fused_tile32_kernel straight from its registers) and is copied to LDS as is.
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
#include <algorithm>
#include <cstdlib>

#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

// ---- candidate tile: gather rows of E, apply dropout, park in LDS as Cs[NT][LDK] (zero padded) -----
// Thread (tid>>3, tid&7) handles rows tid>>3 and 32 + tid>>3, octets (8 columns) tid&7, 8 + tid&7, ...
// keepb (LDS, [NT][32] bytes) caches the keep flags for the dC epilogue; Cm (global [NT][16*KB]) receives the
// masked tile for dq_kernel.  Either may be nullptr.
template <int KB>
__device__ __forceinline__ void load_cand_tile(float *Cs, const float *__restrict__ E, int d, const int32_t *__restrict__ cand_ids,
                                               int cand_first, int N, int n0, const DropDev &drop, bool vec_ok, int tid,
                                               int cand_col0, int64_t table_rows, int *id_err)
{
    // 512 threads: row r = tid / 8, column group tid % 8 (octets o = tid % 8 + 8 it)
    constexpr int LDK = lds_ld(16 * KB), NO = 2 * KB, NOIT = (NO + 7) / 8;
    const int r = tid >> 3, n = n0 + r;
    const bool valid = n < N;
    int64_t cid = 0;
    if (valid) cid = checked_row(cand_ids ? (int64_t)cand_ids[n] : (int64_t)cand_first + n, table_rows, (tid & 7) ? nullptr : id_err);
    const float *row = E + cid * d;
    float4 v0[NOIT], v1[NOIT];
    if (vec_ok) {
        // branch-free: the 2 NOIT 16-byte loads of a thread go out back to back (out-of-range pieces re-read the row's
        // first floats and are zeroed below)
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = (tid & 7) + 8 * it, k = 8 * o;
            const bool in0 = o < NO && valid && k < d, in1 = in0 && k + 4 < d;
            v0[it] = *reinterpret_cast<const float4 *>(row + (in0 ? k : 0));
            v1[it] = *reinterpret_cast<const float4 *>(row + (in1 ? k + 4 : 0));
        }
    } else {
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = (tid & 7) + 8 * it, k = 8 * o;
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = (o < NO && valid && k + e < d) ? row[k + e] : 0.f;
            v0[it] = make_float4(t[0], t[1], t[2], t[3]);
            v1[it] = make_float4(t[4], t[5], t[6], t[7]);
        }
    }
    const uint32_t dstep = drop.enabled ? drop_step(drop) : 0u;
#pragma unroll
    for (int it = 0; it < NOIT; ++it) {
        const int o = (tid & 7) + 8 * it, k = 8 * o;
        if (o < NO) {
            const bool in0 = valid && k < d, in1 = in0 && k + 4 < d;
            uint32_t bits = in0 ? 0xFFu : 0u;
            if (drop.enabled && in0) bits = drop_keep8<true>(drop, (uint32_t)(n + cand_col0), o, d, dstep);
            if (!in0) v0[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!in1) v1[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (drop.enabled) {
                apply_keep4(v0[it], bits & 15u, drop.scale);
                apply_keep4(v1[it], bits >> 4, drop.scale);
            }
            *reinterpret_cast<float4 *>(Cs + r * LDK + k) = v0[it];
            *reinterpret_cast<float4 *>(Cs + r * LDK + k + 4) = v1[it];
        }
    }
}

// acc[kbi] += A^T-style gradient product over 64 contraction rows held in LDS.
//   arow : &A[0][out_row(lane)] element for contraction row 0, slot offset included by caller = base + (16*s)*lda
//   brow : &B[16*s][0] + lane column offset handled here
// Contraction row of (slot s, step t) is 16*s + t.  A is read 4 steps at a time when A_VEC (A stored with
// the contraction index contiguous), else one ds_read_b32 per step.
template <int KB, bool A_VEC, int LDK = lds_ld(16 * KB), int T0 = 0, int T1 = 16>   // LDK: leading dimension of the B tile (wider
__device__ __forceinline__ void grad_product(v4f (&acc)[KB], const float *a_base, int lda, const float *b_base,   // than 16*KB when a
                                             int c)                            // wave takes a column range of it); steps T0 .. T1-1
{
    constexpr int KQ = KB / 4, KR = KB % 4;
    // b_base points at B[16*s][0].  Operands of step t+1 are requested before the KB MFMAs of step t issue.
    v4f pb[KQ > 0 ? KQ : 1], nb[KQ > 0 ? KQ : 1];
    float pr[KR > 0 ? KR : 1], nr[KR > 0 ? KR : 1];
    float pa, na = 0.f;
    auto a_at = [&](int t) { return A_VEC ? a_base[t] : a_base[t * lda]; };   // A[out][16s + t]  |  A[16s + t][out]
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) pb[kq] = *reinterpret_cast<const v4f *>(b_base + T0 * LDK + 64 * kq + 4 * c);
#pragma unroll
    for (int r = 0; r < KR; ++r) pr[r] = b_base[T0 * LDK + 64 * KQ + 16 * r + c];
    pa = a_at(T0);
    __builtin_amdgcn_sched_group_barrier(0x100, KQ + KR + 1, 2);
#pragma unroll
    for (int t = T0; t < T1; ++t) {
        if (t + 1 < T1) {
            const float *brow = b_base + (t + 1) * LDK;
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) nb[kq] = *reinterpret_cast<const v4f *>(brow + 64 * kq + 4 * c);
#pragma unroll
            for (int r = 0; r < KR; ++r) nr[r] = brow[64 * KQ + 16 * r + c];
            na = a_at(t + 1);
        }
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            acc[4 * kq + 0] = mfma16(pa, pb[kq][0], acc[4 * kq + 0]);
            acc[4 * kq + 1] = mfma16(pa, pb[kq][1], acc[4 * kq + 1]);
            acc[4 * kq + 2] = mfma16(pa, pb[kq][2], acc[4 * kq + 2]);
            acc[4 * kq + 3] = mfma16(pa, pb[kq][3], acc[4 * kq + 3]);
        }
#pragma unroll
        for (int r = 0; r < KR; ++r) acc[4 * KQ + r] = mfma16(pa, pr[r], acc[4 * KQ + r]);
        if (t + 1 < T1) {
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) pb[kq] = nb[kq];
#pragma unroll
            for (int r = 0; r < KR; ++r) pr[r] = nr[r];
            pa = na;
            __builtin_amdgcn_sched_group_barrier(0x100, KQ + KR + 1, 2);   // next step's ds_reads first
        }
        __builtin_amdgcn_sched_group_barrier(0x008, KB, 2);                // then this step's MFMAs
    }
}

// Column held by lane column-index c of accumulator block kbi (see "Operand feeding" above).
template <int KB>
__device__ __forceinline__ int grad_col(int kbi, int c)
{
    constexpr int KQ = KB / 4;
    return kbi < 4 * KQ ? 64 * (kbi >> 2) + 4 * c + (kbi & 3) : 64 * KQ + 16 * (kbi - 4 * KQ) + c;
}

// Score / stats / count sweep of one 64-candidate tile over the batch in chunks of 64 rows: ONE 8-wave workgroup per CU
// (the 64 x LDK candidate tile + query chunk fill the LDS), two waves per SIMD.  Wave w: rows 16 (w & 3) .. + 15 of
// the chunk, candidate blocks 2 (w >> 2) and 2 (w >> 2) + 1 (the query operand is shared by the two).  (Round 1-2: four
// waves, one per SIMD, each with all four candidate blocks: nothing overlapped a wave's LDS reads, epilogue VALU and
// barriers, and the sweep sat at 0.43 of the MFMA peak.)
template <int KB, int MODE>
__global__ __launch_bounds__(FUSED_THREADS) void fused_tile_kernel(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDK = lds_ld(16 * KB);
    constexpr int NBW = 2;                                    // candidate blocks per wave
    const int d = a.d;
    float *Cs = reinterpret_cast<float *>(smem);              // [NT][LDK]
    float *Qs = Cs + NT * LDK;                                // [BC][LDK]
    float *Xs = Qs + BC * LDK;                                // [BC][LDG]   X tile staging (score mode), counters, stats partials

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int wq = w & 3, nh = w >> 2;
    const int n0 = blockIdx.x * NT;
    const int b_begin = blockIdx.y * a.b_per_block;
    const int b_end = min(a.B, b_begin + a.b_per_block);
    const bool vec_ok = (d & 3) == 0;

    // register-staged query chunk: thread (row r = tid/8, eighth q = tid%8) holds float4 q + 8 it of that row
    constexpr int NQ = 4 * KB, NQIT = (NQ + 7) / 8;
    const int qr = tid >> 3, qq = tid & 7;
    v4f qreg[NQIT];
    auto fetch_chunk = [&](int b0) {
        const int b = b0 + qr;
        const float *src = a.Q + (size_t)b * a.ldq;
#pragma unroll
        for (int it = 0; it < NQIT; ++it) {
            const int f = min(qq + 8 * it, NQ - 1);           // clamped: surplus lanes reload the last float4
            qreg[it] = (b < b_end) ? *reinterpret_cast<const v4f *>(src + 4 * f) : (v4f){0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch_chunk(b_begin);
    load_cand_tile<KB>(Cs, a.E, d, a.cand_ids, a.cand_first, a.N, n0, a.drop_c, vec_ok, tid, a.cand_col0, a.n_table_rows, a.id_err);

    // MODE_COUNT: the group range of the lane's row is loaded ONE CHUNK AHEAD and the row's first RK_PRE true scores before
    // the score product, so that the counting loop does not wait on dependent global loads
    constexpr int RK_PRE = 4;
    int64_t rk_glo = 0, rk_glo_n = 0, rk_c0 = 0, rk_c1 = 0, rk_c0_n = 0, rk_c1_n = 0;
    int rk_ng = 0, rk_ng_n = 0;
    float rk_t[RK_PRE];
    auto rk_fetch_rows = [&](int b0) {
        if (MODE != MODE_COUNT) return;
        const int b = b0 + 16 * wq + c;
        rk_glo_n = 0; rk_ng_n = 0;
        if (b < b_end) {
            rk_glo_n = a.rk_row_ptr[b];
            rk_ng_n = (int)(a.rk_row_ptr[b + 1] - rk_glo_n);
        }
        rk_c0_n = a.rk_row_ptr[min(b0, a.B)];
        rk_c1_n = a.rk_row_ptr[min(b0 + BC, a.B)];
    };
    rk_fetch_rows(b_begin);

    for (int b0 = b_begin; b0 < b_end; b0 += BC) {
        // ---- phase A: park the prefetched chunk in LDS, prefetch the next chunk -------------------------------
#pragma unroll
        for (int it = 0; it < NQIT; ++it)
            if (qq + 8 * it < NQ) *reinterpret_cast<v4f *>(Qs + qr * LDK + 4 * (qq + 8 * it)) = qreg[it];
        if (b0 + BC < b_end) fetch_chunk(b0 + BC);
        if (MODE == MODE_COUNT) {
            rk_glo = rk_glo_n; rk_ng = rk_ng_n;
#pragma unroll
            for (int jj = 0; jj < RK_PRE; ++jj)
                rk_t[jj] = jj < rk_ng ? a.rk_true[rk_glo + jj] : __builtin_nanf("");      // NaN never compares true
            rk_c0 = rk_c0_n; rk_c1 = rk_c1_n;
            if (b0 + BC < b_end) rk_fetch_rows(b0 + BC);
            // the chunk's packed counters live in LDS (the X staging tile is free in this mode)
            if (a.rk_slab && rk_c1 - rk_c0 <= BC * LDG)
                for (int k = tid; k < (int)(rk_c1 - rk_c0); k += FUSED_THREADS) reinterpret_cast<uint32_t *>(Xs)[k] = 0u;
        }
        __syncthreads();

        // ---- phase B: X = Q_chunk . C^T ; wave w owns rows 16wq..16wq+15 x candidate blocks 2nh, 2nh+1 ----------
        v4f x[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) x[nb] = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            // operands of round r+1 are requested before the MFMAs of round r issue
            const float *qa = Qs + (16 * wq + c) * LDK + 4 * s;
            const float *cb = Cs + (16 * NBW * nh + c) * LDK + 4 * s;
            v4f av = *reinterpret_cast<const v4f *>(qa), bv[NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) bv[nb] = *reinterpret_cast<const v4f *>(cb + 16 * nb * LDK);
            __builtin_amdgcn_sched_group_barrier(0x100, 1 + NBW, 0);
#pragma unroll
            for (int r = 0; r < KB; ++r) {
                v4f an = av, bn[NBW];
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) bn[nb] = bv[nb];
                if (r + 1 < KB) {
                    an = *reinterpret_cast<const v4f *>(qa + 16 * (r + 1));
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) bn[nb] = *reinterpret_cast<const v4f *>(cb + 16 * nb * LDK + 16 * (r + 1));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb)
                        x[nb] = MODE == MODE_COUNT ? mfma16(bv[nb][j], av[j], x[nb])     // X^T block: lane = batch row
                                                   : mfma16(av[j], bv[nb][j], x[nb]);
                av = an;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) bv[nb] = bn[nb];
                __builtin_amdgcn_sched_group_barrier(0x100, 1 + NBW, 0);    // the ds_reads of the next round
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NBW, 0);    // the MFMAs of this round
            }
        }
        // lane holds X[b = b0 + 16wq + 4s + i][n = n0 + 16(2nh + nb) + c] in x[nb][i]
        // (MODE_COUNT, operands swapped: X[b = b0 + 16wq + c][n = n0 + 16(2nh + nb) + 4s + i] -- the same products in the
        //  same order, a * b == b * a, so the same bits -- one batch row per lane)
        const int nb0 = NBW * nh;                              // first candidate block of this wave

        if (MODE == MODE_SCORE) {
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) Xs[(16 * wq + 4 * s + i) * LDG + 16 * (nb0 + nb) + c] = x[nb][i];
            __syncthreads();
            for (int idx = tid; idx < BC * 16; idx += FUSED_THREADS) {
                const int r = idx >> 4, c4 = idx & 15;
                const int b = b0 + r, n = n0 + 4 * c4;
                if (b < b_end) {
                    const v4f v = *reinterpret_cast<const v4f *>(Xs + r * LDG + 4 * c4);
                    float *dst = a.X + (size_t)b * a.ldx + n;
                    if (a.x_vec_ok && n + 3 < a.N) {
                        *reinterpret_cast<v4f *>(dst) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < a.N) dst[e] = v[e];
                    }
                }
            }
            __syncthreads();
        } else if (MODE == MODE_COUNT) {
            // fused evaluation: compare the block against the true scores of each row's answer groups and add the
            // {#greater, #equal} of this tile to the group's counters; the (B, N) score block is never written.
            // (dataset.py:436-446; the filter correction is applied by eval_ranks_block from point scores.)
            // The groups of a chunk's 64 rows are one contiguous index range: their packed counts (#greater | #equal << 16)
            // are summed in LDS and leave as one coalesced store per tile and chunk.  (Global atomics from a few lanes
            // per instruction cost ~100 cycles each and made this sweep 3.5x slower; with the untransposed block, 4 rows
            // x 4 candidates per lane, the per-group compare / cross-lane sum cost 3x the instructions per element.)
            uint32_t *cnt = reinterpret_cast<uint32_t *>(Xs);
            constexpr int CNT_CAP = BC * LDG;
            const int64_t gc_lo = rk_c0, gc_hi = rk_c1;
            const bool in_lds = a.rk_slab && gc_hi - gc_lo <= CNT_CAP;
            uint32_t *slab_row = a.rk_slab ? a.rk_slab + (size_t)blockIdx.x * a.rk_ngroups : nullptr;
            float xm[NBW][4];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                for (int i = 0; i < 4; ++i) xm[nb][i] = n0 + 16 * (nb0 + nb) + 4 * s + i < a.N ? x[nb][i] : -INFINITY;
            const int64_t g_lo = rk_glo;
            const int ng = rk_ng;
            // a row's 64 candidates of this tile sit in eight lanes: c, c+16, c+32, c+48 of the waves (wq, 0) and (wq, 1).
            // Counters in LDS: all eight add atomically.  Otherwise (a chunk with more groups than the LDS tile holds, or
            // the atomics path of very large batches): the wave pair takes turns -- first half stores, second half adds.
            auto count_pass = [&](bool add) {
                for (int j = 0; __builtin_amdgcn_ballot_w64(j < ng) != 0; ++j) {
                    float t = j < ng && j >= RK_PRE ? a.rk_true[g_lo + j] : __builtin_nanf("");
#pragma unroll
                    for (int jj = 0; jj < RK_PRE; ++jj) t = j == jj ? rk_t[jj] : t;
                    int pq[NBW];
                    float mx = -INFINITY;
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) {
                        pq[nb] = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            pq[nb] += xm[nb][i] > t;
                            mx = fmaxf(mx, xm[nb][i] <= t ? xm[nb][i] : -INFINITY);      // largest score not above t
                        }
                    }
                    int pk = pq[0] + pq[1];
                    const bool any_eq = mx == t;
                    if (__builtin_amdgcn_ballot_w64(any_eq) != 0) {      // exact ties are rare (the true answer's own tile)
#pragma unroll
                        for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
                            for (int i = 0; i < 4; ++i) pk += xm[nb][i] == t ? 65536 : 0;
                    }
                    if (j < ng) {
                        if (in_lds) {
                            if (pk) atomicAdd(&cnt[g_lo + j - gc_lo], (uint32_t)pk);
                        } else {
                            pk += __shfl_xor(pk, 16);
                            pk += __shfl_xor(pk, 32);
                            if (s == 0) {
                                if (slab_row) slab_row[g_lo + j] = add ? slab_row[g_lo + j] + (uint32_t)pk : (uint32_t)pk;
                                else {
                                    if (pk & 0xFFFF) atomicAdd(a.rk_counts + 2 * (g_lo + j), pk & 0xFFFF);
                                    if (pk >> 16) atomicAdd(a.rk_counts + 2 * (g_lo + j) + 1, pk >> 16);
                                }
                            }
                        }
                    }
                }
            };
            if (in_lds) {
                count_pass(false);
                __syncthreads();
                for (int k = tid; k < (int)(gc_hi - gc_lo); k += FUSED_THREADS) slab_row[gc_lo + k] = cnt[k];
            } else {
                if (nh == 0) count_pass(false);
                __syncthreads();
                if (nh == 1) count_pass(true);
                __syncthreads();
            }
        } else {
            // MODE_STATS: per row of the chunk, max and sum-exp over this tile's candidates: each wave reduces its 32
            // candidates, the second wave of a row parks its pair in LDS and the first merges (running max / sum-exp)
            float2 *part = reinterpret_cast<float2 *>(Xs);       // [BC]
            float pm[4], pse[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m = -INFINITY;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    if (n0 + 16 * (nb0 + nb) + c < a.N) m = fmaxf(m, x[nb][i]);
                m = row16_max(m);
                float se = 0.f;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb)
                    if (n0 + 16 * (nb0 + nb) + c < a.N) se += __expf(x[nb][i] - m);
                se = row16_sum(se);
                pm[i] = m; pse[i] = se;
                if (nh == 1 && c == 0) part[16 * wq + 4 * s + i] = make_float2(m, se);
            }
            __syncthreads();
            if (nh == 0 && c == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int b = b0 + 16 * wq + 4 * s + i;
                    const float2 o = part[16 * wq + 4 * s + i];
                    const float m = fmaxf(pm[i], o.x);
                    const float se = (pm[i] > -INFINITY ? pse[i] * __expf(pm[i] - m) : 0.f) + (o.x > -INFINITY ? o.y * __expf(o.x - m) : 0.f);
                    if (b < b_end) {
                        float2 *dst = reinterpret_cast<float2 *>(a.stats) + (size_t)blockIdx.x * a.Bpad + b;
                        *dst = make_float2(m, se);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- dQ = G . C over a candidate range -------------------------------------------------------------
// G^T block (64 candidates x 64 batch rows, 16 KB contiguous) and masked
// candidate tile (64 x 16*KB) of one chunk -> registers
// slot sizes above 256: the 132 KB candidate tile leaves room for one workgroup per CU; it runs 8 waves, waves w and
// w + 4 taking the two halves of the output columns (no exchange needed: disjoint outputs, shared G^T operand)
template <int KB> struct DqCfg {
    static constexpr int KS = KB <= 16 ? 1 : 2;
    static constexpr int THREADS = 256 * KS;
    static constexpr int QG = 8 * KS;                 // staging: column groups per row
    static constexpr int NO = 2 * KB, NOIT = (NO + QG - 1) / QG;
    static constexpr int GV = 4 / KS;                 // float4 of the G^T block per thread
};

template <int KB>
__device__ __forceinline__ void dq_prefetch(v4f (&gv)[DqCfg<KB>::GV], v4f (&cv)[4 * DqCfg<KB>::NOIT],
                                            const float *__restrict__ g_blk, const DqArgs &a, int ch, int tid)
{
    using Cfg = DqCfg<KB>;
    constexpr int NO = Cfg::NO, NOIT = Cfg::NOIT, QG = Cfg::QG;
#pragma unroll
    for (int it = 0; it < Cfg::GV; ++it) gv[it] = *reinterpret_cast<const v4f *>(g_blk + (size_t)(tid + it * Cfg::THREADS) * 4);
    const float *cm = a.Cm + (size_t)ch * NT * (16 * KB);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int r = tid / QG + 32 * pass;
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = min(tid % QG + QG * it, NO - 1);   // clamped: surplus lanes reload the last octet
            cv[(2 * pass) * NOIT + it] = *reinterpret_cast<const v4f *>(cm + (size_t)r * (16 * KB) + 8 * o);
            cv[(2 * pass + 1) * NOIT + it] = *reinterpret_cast<const v4f *>(cm + (size_t)r * (16 * KB) + 8 * o + 4);
        }
    }
}

constexpr int LDGT = 68;   // G^T tile leading dimension (16-byte aligned rows; the A operand is one ds_read_b32 per 13 MFMAs,
                           // so its 2-way conflict between slots 16 rows apart is irrelevant)

template <int KB>
__global__ __launch_bounds__(DqCfg<KB>::THREADS) void dq_kernel(const DqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Cfg = DqCfg<KB>;
    constexpr int LDK = lds_ld(16 * KB);
    constexpr int KS = Cfg::KS, NTHR = Cfg::THREADS, QG = Cfg::QG;
    constexpr int KBW = KB / KS;                      // 16-column output blocks of one wave
    constexpr int KQ = KBW / 4, KR = KBW % 4;
    static_assert(KS == 1 || KB % 8 == 0, "the column split needs whole quads of 16-column blocks per wave");
    constexpr int NO = Cfg::NO, NOIT = Cfg::NOIT;
    float *Cs = reinterpret_cast<float *>(smem);      // [NT][LDK]   masked candidate rows
    float *Gt = Cs + NT * LDK;                        // [NT (n)][LDGT] : G^T tile, 64 batch rows wide

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int ks = KS == 1 ? 0 : w >> 2, wq = w & 3;  // column half, 16-row block of the 64 batch rows
    const int split = blockIdx.x % a.nsplit, bblk = blockIdx.x / a.nsplit;
    const int b0 = bblk * BC;
    const int nJ = a.Bpad / BC;
    const int nchunks = (a.N + NT - 1) / NT;
    const int ch_lo = (int)((int64_t)split * nchunks / a.nsplit);
    const int ch_hi = (int)((int64_t)(split + 1) * nchunks / a.nsplit);

    v4f acc[KBW];                                     // dQ[b = b0 + 16wq + 4s + i][k = 16*KBW*ks + grad_col(kbi, c)]
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) acc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};

    // register-staged prefetch: the next chunk's G block and masked candidate tile are in flight during the MFMAs
    v4f gv[Cfg::GV], cv[4 * NOIT];   // cv[(2*pass + half) * NOIT + it]
    auto g_block = [&](int ch) { return a.G + ((size_t)ch * nJ + bblk) * 4096; };
    if (ch_lo < ch_hi) dq_prefetch<KB>(gv, cv, g_block(ch_lo), a, ch_lo, tid);
    for (int ch = ch_lo; ch < ch_hi; ++ch) {
        // G^T block -> LDS (float4 number f: candidate f >> 4, batch rows 4 * (f & 15) ..)
#pragma unroll
        for (int it = 0; it < Cfg::GV; ++it) {
            const int f = tid + it * NTHR;
            *reinterpret_cast<v4f *>(Gt + (f >> 4) * LDGT + 4 * (f & 15)) = gv[it];
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = tid / QG + 32 * pass;
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = tid % QG + QG * it;
                if (o < NO) {
                    *reinterpret_cast<v4f *>(Cs + r * LDK + 8 * o) = cv[(2 * pass) * NOIT + it];
                    *reinterpret_cast<v4f *>(Cs + r * LDK + 8 * o + 4) = cv[(2 * pass + 1) * NOIT + it];
                }
            }
        }
        __syncthreads();
        if (ch + 1 < ch_hi) dq_prefetch<KB>(gv, cv, g_block(ch + 1), a, ch + 1, tid);
        // A[i = b][slot s, step t] = G^T[n = 16s + t][b = 16wq + c] ; B[slot][k] = C[n = 16s + t][k of this wave's half]
        grad_product<KBW, false, LDK>(acc, Gt + 16 * s * LDGT + 16 * wq + c, LDGT, Cs + 16 * s * LDK + 16 * KBW * ks, c);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *dst = a.slab + ((size_t)split * a.Bpad + b0 + 16 * wq + 4 * s + i) * a.ldq + 16 * KBW * ks;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            v4f v = (v4f){acc[4 * kq][i], acc[4 * kq + 1][i], acc[4 * kq + 2][i], acc[4 * kq + 3][i]};
            if (a.accumulate) v += *reinterpret_cast<const v4f *>(dst + 64 * kq + 4 * c);
            *reinterpret_cast<v4f *>(dst + 64 * kq + 4 * c) = v;
        }
#pragma unroll
        for (int r = 0; r < KR; ++r)
            dst[64 * KQ + 16 * r + c] = a.accumulate ? dst[64 * KQ + 16 * r + c] + acc[4 * KQ + r][i] : acc[4 * KQ + r][i];
    }
}

// ---- dQ = G . C, eight waves: the contraction of a 64-candidate chunk split over two wave groups ---------------------
// Slot sizes up to 256.  Same workgroup-level job as dq_kernel (64 batch rows x a candidate range -> one slab), but one
// 8-wave workgroup per CU takes a range TWICE as long: half the slabs (13.6 instead of 27 MB written here and read back
// by the prefix backward at S-FB), half the first-chunk burst at kernel start, prologue / epilogue amortised over twice
// the chunks -- at two waves per SIMD like the two co-resident 4-wave workgroups before.  Wave (wq = w & 3, kh = w >> 2):
// batch rows 16wq.., contraction steps 8kh .. 8kh+7 of every slot (candidates 16s + 8kh + t: slots stay 16 rows apart, the
// B operand reads keep their conflict-free bank pattern); the two halves are added through LDS before the slab store.
// DB (slot sizes up to 208: two tile pairs fit the LDS): chunk ch+1 is parked in the second pair while chunk ch is being
// multiplied -- ONE barrier per chunk and no staging phase in which all eight waves write LDS and request the next chunk
// while the MFMA pipes idle.  The two waves of a SIMD take turns: group kh = 0 parks before its MFMAs, group kh = 1 after.
template <int KB, bool DB>
__global__ __launch_bounds__(512, 2) void dq8_kernel(const DqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDK = lds_ld(16 * KB);
    constexpr int NTHR = 512, QG = 16;                // staging: 16 column groups per row, 32 rows per pass
    constexpr int KQ = KB / 4, KR = KB % 4;
    constexpr int NO = 2 * KB, NOIT = (NO + QG - 1) / QG;
    constexpr int PAIR = NT * LDK + NT * LDGT;        // floats of one (candidate tile, G^T tile) pair
    float *Cs = reinterpret_cast<float *>(smem);      // [NT][LDK]   masked candidate rows (end: the kh = 1 partial sums)
    float *Gt = Cs + NT * LDK;                        // [NT (n)][LDGT] : G^T tile, 64 batch rows wide

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int wq = w & 3, kh = w >> 2;
    const int split = blockIdx.x % a.nsplit, bblk = blockIdx.x / a.nsplit;
    const int b0 = bblk * BC;
    const int nJ = a.Bpad / BC;
    const int nchunks = (a.N + NT - 1) / NT;
    const int ch_lo = (int)((int64_t)split * nchunks / a.nsplit);
    const int ch_hi = (int)((int64_t)(split + 1) * nchunks / a.nsplit);

    v4f acc[KB];                                      // dQ[b = b0 + 16wq + 4s + i][k = grad_col(kbi, c)], this wave's candidates
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};

    // register-staged prefetch of the next chunk: 2 float4 of the G^T block, 2 x NOIT x 2 float4 of the candidate tile
    v4f gv[2], cv[4 * NOIT];
    auto prefetch = [&](int ch) {
        const float *g_blk = a.G + ((size_t)ch * nJ + bblk) * 4096;
        const float *cm = a.Cm + (size_t)ch * NT * (16 * KB);
#pragma unroll
        for (int it = 0; it < 2; ++it) gv[it] = *reinterpret_cast<const v4f *>(g_blk + (size_t)(tid + it * NTHR) * 4);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = tid / QG + 32 * pass;
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = min(tid % QG + QG * it, NO - 1);   // clamped: surplus lanes reload the last octet
                cv[(2 * pass) * NOIT + it] = *reinterpret_cast<const v4f *>(cm + (size_t)r * (16 * KB) + 8 * o);
                cv[(2 * pass + 1) * NOIT + it] = *reinterpret_cast<const v4f *>(cm + (size_t)r * (16 * KB) + 8 * o + 4);
            }
        }
    };
    // registers -> the LDS pair `buf` (0 / 1)
    auto park = [&](int buf) {
        float *cs = Cs + buf * PAIR, *gt = Gt + buf * PAIR;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int f = tid + it * NTHR;
            *reinterpret_cast<v4f *>(gt + (f >> 4) * LDGT + 4 * (f & 15)) = gv[it];
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = tid / QG + 32 * pass;
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = tid % QG + QG * it;
                if (o < NO) {
                    *reinterpret_cast<v4f *>(cs + r * LDK + 8 * o) = cv[(2 * pass) * NOIT + it];
                    *reinterpret_cast<v4f *>(cs + r * LDK + 8 * o + 4) = cv[(2 * pass + 1) * NOIT + it];
                }
            }
        }
    };
    // A[i = b][slot s, step t] = G^T[n = 16s + t][b = 16wq + c] ; B[slot][k] = C[n = 16s + t][k]
    auto product = [&](int buf) {
        const float *cs = Cs + buf * PAIR, *gt = Gt + buf * PAIR;
        if (kh == 0) grad_product<KB, false, LDK, 0, 8>(acc, gt + 16 * s * LDGT + 16 * wq + c, LDGT, cs + 16 * s * LDK, c);
        else         grad_product<KB, false, LDK, 8, 16>(acc, gt + 16 * s * LDGT + 16 * wq + c, LDGT, cs + 16 * s * LDK, c);
    };
    if (ch_lo < ch_hi) prefetch(ch_lo);
    if (DB) {
        if (ch_lo < ch_hi) {
            park(0);
            if (ch_lo + 1 < ch_hi) prefetch(ch_lo + 1);
        }
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            const int buf = (ch - ch_lo) & 1;
            __syncthreads();                       // chunk ch is parked; the other pair's readers (chunk ch - 1) are done
            const bool more = ch + 1 < ch_hi;
            if (kh == 0 && more) {                 // this wave group parks its share of chunk ch + 1 first ...
                park(buf ^ 1);
                if (ch + 2 < ch_hi) prefetch(ch + 2);
            }
            product(buf);
            if (kh == 1 && more) {                 // ... the other one after its MFMAs: a SIMD's two waves take turns
                park(buf ^ 1);
                if (ch + 2 < ch_hi) prefetch(ch + 2);
            }
        }
        __syncthreads();                           // the pairs are free: pair 0's candidate tile takes the kh = 1 partial sums
    } else {
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            park(0);
            __syncthreads();
            if (ch + 1 < ch_hi) prefetch(ch + 1);
            product(0);
            __syncthreads();
        }
    }
    // the two contraction halves: kh = 1 parks its partial rows in LDS, kh = 0 adds them and stores the slab rows
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *dst = Cs + (16 * wq + 4 * s + i) * LDK;
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq)
                *reinterpret_cast<v4f *>(dst + 64 * kq + 4 * c) = (v4f){acc[4 * kq][i], acc[4 * kq + 1][i], acc[4 * kq + 2][i], acc[4 * kq + 3][i]};
#pragma unroll
            for (int r = 0; r < KR; ++r) dst[64 * KQ + 16 * r + c] = acc[4 * KQ + r][i];
        }
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *src = Cs + (16 * wq + 4 * s + i) * LDK;
            float *dst = a.slab + ((size_t)split * a.Bpad + b0 + 16 * wq + 4 * s + i) * a.ldq;
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                v4f v = (v4f){acc[4 * kq][i], acc[4 * kq + 1][i], acc[4 * kq + 2][i], acc[4 * kq + 3][i]} +
                        *reinterpret_cast<const v4f *>(src + 64 * kq + 4 * c);
                if (a.accumulate) v += *reinterpret_cast<const v4f *>(dst + 64 * kq + 4 * c);
                *reinterpret_cast<v4f *>(dst + 64 * kq + 4 * c) = v;
            }
#pragma unroll
            for (int r = 0; r < KR; ++r) {
                const float v = acc[4 * KQ + r][i] + src[64 * KQ + 16 * r + c];
                dst[64 * KQ + 16 * r + c] = a.accumulate ? dst[64 * KQ + 16 * r + c] + v : v;
            }
        }
    }
}

// ---- dQ = G . C for slot sizes above 256 (KB = 32): 32-candidate chunks, double-buffered ------------------------------------
// dq_kernel<32> keeps a 64-candidate tile (132 KB) + its G^T block in LDS, single-buffered: per chunk every wave parks 16
// float4, the workgroup passes two barriers and waits for the next chunk's loads while the MFMA pipes idle -- 0.52 of the
// fp32-MFMA peak at the DistMult d = 512 shape, and (one workgroup per CU but a grid sized for two) 64 candidate splits =
// 64 MB of dQ slabs that the prefix backward reads back.  Here a chunk is 32 candidates: two (candidate tile, G^T tile) pairs
// fit the LDS (2 x 66 KB + 2 x 8.5 KB), chunk ch + 1 is parked in the second pair while chunk ch is being multiplied -- ONE
// barrier per chunk, no staging phase --, the two waves of a SIMD take turns (group ks = 0 parks before its MFMAs, group
// ks = 1 after), and the candidate range of a workgroup is twice as long (32 splits: half the slabs).  Wave (wq = w & 3,
// ks = w >> 2): batch rows 16 wq .., output columns 256 ks ..; contraction rows of slot s: candidates 8 s + t, t < 8.
// KB = 16 (slot sizes 209 .. 256, where two 64-candidate pairs do not fit the LDS and dq8_kernel runs its single-pair loop):
// the same loop on 128-column halves.
template <int KB, bool DEEP>
__global__ __launch_bounds__(512, 2) void dq8k_kernel(const DqArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LDK = lds_ld(16 * KB), NC = 32;     // NC candidates per chunk
    constexpr int KBW = KB / 2, KQ = KBW / 4, NV = KB / 4;   // NV float4 of a chunk row per staging thread
    static_assert(KB % 8 == 0, "two column halves of whole float4 quads");
    constexpr int PAIR = NC * LDK + NC * LDGT;        // floats of one (candidate tile, G^T tile) pair
    float *Cs = reinterpret_cast<float *>(smem);      // [NC][LDK]
    float *Gt = Cs + NC * LDK;                        // [NC (n)][LDGT]

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int wq = w & 3, ks = w >> 2;
    const int split = blockIdx.x % a.nsplit, bblk = blockIdx.x / a.nsplit;
    const int b0 = bblk * BC;
    const int nJ = a.Bpad / BC;
    const int nchunks = 2 * ((a.N + NT - 1) / NT);    // whole 64-candidate blocks: the tile kernel wrote all of their rows
    const int ch_lo = (int)((int64_t)split * nchunks / a.nsplit);
    const int ch_hi = (int)((int64_t)(split + 1) * nchunks / a.nsplit);

    v4f acc[KBW];                                     // dQ[b = b0 + 16wq + 4s + i][k = 16 KBW ks + grad_col(kbi, c)]
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) acc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};

    const int r16 = tid >> 4, q16 = tid & 15;         // staging: row r16 of the chunk, float4 columns q16 + 16 it
    // register sets of chunks in flight: one (the chunk after the one being multiplied), or DEEP: two -- chunk ch + 3 is
    // requested while chunk ch is multiplied and parked two chunks later, i.e. two chunk times (~4 us) to cover the load latency
    constexpr int NSET = DEEP ? 2 : 1;
    v4f gv[NSET], cv[NSET][NV];
    auto prefetch = [&](int ch, auto set) {
        constexpr int S = decltype(set)::value;
        gv[S] = *reinterpret_cast<const v4f *>(a.G + ((size_t)(ch >> 1) * nJ + bblk) * 4096 + (ch & 1) * 2048 + (size_t)tid * 4);
        const float *cm = a.Cm + ((size_t)ch * NC + r16) * (16 * KB);
#pragma unroll
        for (int it = 0; it < NV; ++it) cv[S][it] = *reinterpret_cast<const v4f *>(cm + 4 * (q16 + 16 * it));
    };
    auto park = [&](int buf, auto set) {
        constexpr int S = decltype(set)::value;
        float *cs = Cs + buf * PAIR, *gt = Gt + buf * PAIR;
        *reinterpret_cast<v4f *>(gt + (tid >> 4) * LDGT + 4 * (tid & 15)) = gv[S];    // float4 number tid: candidate tid >> 4
#pragma unroll
        for (int it = 0; it < NV; ++it) *reinterpret_cast<v4f *>(cs + r16 * LDK + 4 * (q16 + 16 * it)) = cv[S][it];
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, DEEP ? 1 : 0>;
    // one chunk: `buf` = the pair chunk ch was parked in; the next chunk goes from register set `nxt` into the other pair, and
    // the freed set takes chunk ch + 1 + NSET
    auto step = [&](int ch, int buf, auto nxt) {
        __syncthreads();                       // chunk ch is parked; the other pair's readers (chunk ch - 1) are done
        const bool more = ch + 1 < ch_hi;
        if (ks == 0 && more) {                 // this wave group parks its share of chunk ch + 1 first ...
            park(buf ^ 1, nxt);
            if (ch + 1 + NSET < ch_hi) prefetch(ch + 1 + NSET, nxt);
        }
        // A[i = b][slot s, step t] = G^T[n = 8s + t][b = 16wq + c] ; B[slot][k] = C[n = 8s + t][16 KBW ks + k]
        grad_product<KBW, false, LDK, 0, 8>(acc, Gt + buf * PAIR + 8 * s * LDGT + 16 * wq + c, LDGT,
                                            Cs + buf * PAIR + 8 * s * LDK + 16 * KBW * ks, c);
        if (ks == 1 && more) {                 // ... the other one after its MFMAs: a SIMD's two waves take turns
            park(buf ^ 1, nxt);
            if (ch + 1 + NSET < ch_hi) prefetch(ch + 1 + NSET, nxt);
        }
    };
    if (ch_lo < ch_hi) {
        prefetch(ch_lo, S0{});
        park(0, S0{});
        if (DEEP) {
            if (ch_lo + 1 < ch_hi) prefetch(ch_lo + 1, S1{});
            if (ch_lo + 2 < ch_hi) prefetch(ch_lo + 2, S0{});
        } else if (ch_lo + 1 < ch_hi) {
            prefetch(ch_lo + 1, S0{});
        }
    }
    // (chunk ch_lo + j: LDS pair j & 1; its registers were set (j & 1) of DEEP, else the one set)
    for (int ch = ch_lo; ch < ch_hi; ch += 2) {
        step(ch, 0, S1{});
        if (ch + 1 < ch_hi) step(ch + 1, 1, S0{});
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *dst = a.slab + ((size_t)split * a.Bpad + b0 + 16 * wq + 4 * s + i) * a.ldq + 16 * KBW * ks;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            v4f v = (v4f){acc[4 * kq][i], acc[4 * kq + 1][i], acc[4 * kq + 2][i], acc[4 * kq + 3][i]};
            if (a.accumulate) v += *reinterpret_cast<const v4f *>(dst + 64 * kq + 4 * c);
            *reinterpret_cast<v4f *>(dst + 64 * kq + 4 * c) = v;
        }
    }
}

// ---- host-side launchers -----------------------------------------------------------------------------
template <int KB, int MODE>
static hipError_t launch_fused_t(const FusedArgs &a, dim3 grid, size_t shmem, hipStream_t st)
{
    auto k = fused_tile_kernel<KB, MODE>;
    static LdsOptIn lds_opt_in;
    if (hipError_t e = ensure_dynamic_lds(lds_opt_in, reinterpret_cast<const void *>(k), shmem); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(FUSED_THREADS), shmem, st, a);
    return hipGetLastError();
}

template <int KB>
static hipError_t launch_fused_m(int mode, const FusedArgs &a, dim3 grid, size_t shmem, hipStream_t st)
{
    switch (mode) {
        case MODE_SCORE: return launch_fused_t<KB, MODE_SCORE>(a, grid, shmem, st);
        case MODE_STATS: return launch_fused_t<KB, MODE_STATS>(a, grid, shmem, st);
        case MODE_COUNT: return launch_fused_t<KB, MODE_COUNT>(a, grid, shmem, st);
        default:         return hipErrorInvalidValue;      // training: fused_tile32_kernel
    }
}

size_t fused_shmem_bytes(int LDK)
{
    return (size_t)(NT + BC) * LDK * sizeof(float) + (size_t)BC * LDG * sizeof(float);
}

size_t dq_shmem_bytes(int LDK)
{
    return (size_t)NT * LDK * sizeof(float) + (size_t)NT * 68 * sizeof(float);
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
    static LdsOptIn lds_opt_in;
    if (hipError_t e = ensure_dynamic_lds(lds_opt_in, reinterpret_cast<const void *>(k), shmem); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid_x), dim3(DqCfg<KB>::THREADS), shmem, st, a);
    return hipGetLastError();
}

template <int KB>
static hipError_t launch_dq8_t(const DqArgs &a, int grid_x, size_t shmem, hipStream_t st)
{
    // two LDS tile pairs (slot sizes up to 208); OKGE_DQ8_DB=0: the single-pair loop
    static const bool db_on = [] { const char *e = getenv("OKGE_DQ8_DB"); return !e || atoi(e) != 0; }();
    constexpr bool CAN_DB = KB <= 13;
    const bool db = CAN_DB && db_on;
    auto k = db ? dq8_kernel<KB, CAN_DB> : dq8_kernel<KB, false>;
    if (db) shmem *= 2;
    static LdsOptIn lds_opt_in;
    if (hipError_t e = ensure_dynamic_lds(lds_opt_in, reinterpret_cast<const void *>(k), shmem); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid_x), dim3(512), shmem, st, a);
    return hipGetLastError();
}

template <int KB>
static hipError_t launch_dq8k_t(const DqArgs &a, int grid_x, hipStream_t st)
{
    const size_t sh = (size_t)2 * (32 * lds_ld(16 * KB) + 32 * LDGT) * sizeof(float);
    // two chunks of loads in flight per thread at KB = 16 (cfg4 shard: 1059 -> 1046 us per range; at KB = 32 the 36 extra
    // registers cost more than the latency they cover: 53.6 -> 54.2 us at cfg3); OKGE_DQ_DEEP=0 / 1 overrides
    static const bool deep = [] { const char *e = getenv("OKGE_DQ_DEEP"); return e ? atoi(e) != 0 : KB == 16; }();
    auto k = deep ? dq8k_kernel<KB, true> : dq8k_kernel<KB, false>;
    static LdsOptIn lds_opt_in[2];
    if (hipError_t e = ensure_dynamic_lds(lds_opt_in[deep], reinterpret_cast<const void *>(k), sh); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid_x), dim3(512), sh, st, a);
    return hipGetLastError();
}

hipError_t launch_dq(const DqArgs &a, int grid_x, hipStream_t st)
{
    const size_t shmem = dq_shmem_bytes(a.LDK);
    if (a.waves8) {
        switch (a.KB) {
            case 4:  return launch_dq8_t<4>(a, grid_x, shmem, st);
            case 8:  return launch_dq8_t<8>(a, grid_x, shmem, st);
            case 13: return launch_dq8_t<13>(a, grid_x, shmem, st);
            case 16: {
                // 32-candidate chunks in two LDS pairs (dq8k_kernel<16>); OKGE_DQ8K16=0: the single-pair loop of dq8_kernel
                static const bool k16 = [] { const char *e = getenv("OKGE_DQ8K16"); return !e || atoi(e) != 0; }();
                return k16 ? launch_dq8k_t<16>(a, grid_x, st) : launch_dq8_t<16>(a, grid_x, shmem, st);
            }
            default: break;
        }
    }
    if (a.waves8 && a.KB == 32) return launch_dq8k_t<32>(a, grid_x, st);
    switch (a.KB) {
        case 4:  return launch_dq_t<4>(a, grid_x, shmem, st);
        case 8:  return launch_dq_t<8>(a, grid_x, shmem, st);
        case 13: return launch_dq_t<13>(a, grid_x, shmem, st);
        case 16: return launch_dq_t<16>(a, grid_x, shmem, st);
        case 32: return launch_dq_t<32>(a, grid_x, shmem, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace okge
