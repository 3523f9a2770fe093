// fused_tile64_kernel<KB, MODE>: the fused score -> loss -> dCand tile kernel for slot sizes up to 256, one 8-wave
// workgroup per CU: 64 candidates x 64 batch rows per step.  Replaces the reference's encode_obj(candidates) + 4 mm +
// cat + BCEWithLogits / log_softmax+KLDiv forward and the mm / sigmoid half of autograd's backward
// (openkge/model.py:198-229,268-274,455-510; openkge/trainer.py:75-106,234).
//
// One workgroup owns 64 candidate entities: it gathers and drops them out ONCE into LDS (and hands the masked rows
// to dq_kernel through `Cm`), then sweeps the batch's folded query rows in
// 64-row chunks.  Wave roles: blk = w & 3 (16-candidate block of the tile), h = w >> 2 (32-row half of the chunk: row
// groups 2h, 2h + 1).
//   score product : the wave's two 16x16 blocks X[rows of group][candidates of blk], ONE candidate operand
//                   (ds_read_b128, four contraction steps) feeds both blocks; the two accumulator chains alternate
//                   (40-cycle dependent latency of v_mfma_f32_16x16x4_f32 behind a 32-cycle issue).
//   loss epilogue : on the accumulator registers (8 elements per lane); label bits from an LDS bitmask.
//   dC product    : dC[candidates of blk][all columns] += G^T . Q over the wave's 32 batch rows; the G blocks are the A
//                   operands straight from the epilogue's registers (the MFMA result layout of X is the A layout of
//                   the next product); both row groups accumulate into the SAME 13 x 4 accumulator registers.
// Why this cut (measured on the 32 x 32 predecessor, two 4-wave workgroups per CU, profiles/round2_*): fp32 MFMA shares
// the SIMD's issue with VALU, so the loop costs MFMA + VALU cycles; per MFMA this cut stages half as many query rows,
// passes half as many barriers and issues 40 % of the LDS operand reads, and the per-chunk address / label / loop VALU is
// spread over twice the MFMAs per wave.
// The two row halves' dC partial sums are added through LDS in the write-back, which also applies the cached dropout
// flags.  G is stored UNMASKED: padding rows (b >= B) have zero query rows and candidates n >= N zero candidate rows, so
// whatever dLoss/dX says there meets a zero in both gradient products and the dQ rows of padding are never read.  G leaves for dq_kernel from registers as 64x64 transposed blocks Gt[(T * nJ + J)][n_local][b_local].
#include <cstdio>
#include <cstdlib>

#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

constexpr int NT64 = 64, BC64 = 64, T64_THREADS = 512;

template <int KB> struct Tile64Cfg {
    static constexpr int LDK = lds_ld(16 * KB);
    static constexpr int NO = 2 * KB;                         // 8-column octets per row
    static constexpr int KEEP_LD = NO < 32 ? 32 : NO;         // keep-flag bytes per row (LDS and global)
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
#define TL_STAMP_AT(idx)                                                                                          \
    do {                                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        unsigned long long t_;                                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                  \
        if (a.stamps_dbg && threadIdx.x == 0)                                                                      \
            a.stamps_dbg[(size_t)gridDim.x * gridDim.y * 4 + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 80 + (idx)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
    } while (0)
#else
#define TL_STAMP() do { } while (0)
#define TL_STAMP_AT(idx) do { } while (0)
#endif

// REGC (slot sizes up to 208): every lane keeps its slice of the wave's candidate block -- the B operand of the whole
// score sweep, KB float4 -- in registers, read from the gathered tile once.  The tile's LDS then serves as a SECOND query
// chunk buffer: chunk i+1 is parked while chunk i is being worked on, so a chunk has ONE barrier and no staging phase
// (with one buffer every wave parks, requests and waits between two barriers while the MFMA pipes idle: ~1.3 K of a
// chunk's 17.9 K cycles), and the score product reads a third less from LDS.
// SHORTK (with REGC; the launcher picks it for slot sizes 193..200 at KB = 13, i.e. the d = 200 of BASELINE configs[1]): the
// contraction needs only 200 of the 208 padded columns.  In the normal k order MFMA j of a round takes the columns
// 16 r + 4 s + j of lane slot s, so all four MFMAs of the last round touch a padding column; here the LAST round is fed in the
// order 16 r + 4 j + s (two ds_read_b32 per block instead of one ds_read_b128, the candidate operand read that way once in
// the prologue), which puts columns 200..207 into MFMAs 2 and 3 -- and those are not issued: 50 instead of 52 score MFMAs
// per block.  (The gradient product still writes all 208 columns: its OUTPUT is padded, not its contraction.)
template <int KB, int MODE, bool REGC, bool SHORTK = false>
__global__ __launch_bounds__(T64_THREADS, 2) void fused_tile64_kernel(const FusedArgs a)
{
    static_assert(!SHORTK || (REGC && KB >= 3), "the short last round comes with the register-resident candidate operand");
#ifdef OKGE_STAMPS
    unsigned long long wg_t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0)::"memory");
    int tl_n = 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Cfg = Tile64Cfg<KB>;
    constexpr int LDK = Cfg::LDK, NO = Cfg::NO, KEEP_LD = Cfg::KEEP_LD;
    constexpr int KQ = KB / 4, KR = KB % 4;
    constexpr int QG = 8;                                       // staging: column groups per row (512 threads / 64 rows)
    constexpr int NOIT = (NO + QG - 1) / QG;
    constexpr int NQ = 4 * KB, NQIT = (NQ + QG - 1) / QG;       // float4 per row
    static_assert(MODE == MODE_TRAIN_BCE || MODE == MODE_TRAIN_KL, "training kernel");
    const int d = a.d;
    float *Cs = reinterpret_cast<float *>(smem);              // [64][LDK]
    float *Qs = Cs + NT64 * LDK;                              // [64][LDK]
    uint32_t *ybits3 = reinterpret_cast<uint32_t *>(Qs + BC64 * LDK);    // [3][2 halves][64 rows] label bits, three chunks in rotation
    double *red = reinterpret_cast<double *>(ybits3 + 3 * BC64 * 2);     // [8]
    uint8_t *keepb = reinterpret_cast<uint8_t *>(red + 8);               // [64][KEEP_LD] keep flags of the tile
    uint32_t *posc = reinterpret_cast<uint32_t *>(keepb + NT64 * KEEP_LD);   // [POS_CACHE] (row << 6 | col)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c = lane & 15, s = lane >> 4;
    const int blk = w & 3, h = w >> 2;
    const int n0 = blockIdx.x * NT64;
    const int b_begin = blockIdx.y * a.b_per_block;
    const int b_end = min(a.B, b_begin + a.b_per_block);
    const bool vec_ok = (d & 3) == 0;
    const int r8 = tid >> 3, q8 = tid & 7;                     // staging role: row r8, column group q8

    // ---- register-staged query chunk (64 rows x 16*KB): thread holds float4 columns q8 + 8*it of row r8 -----------
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
    const uint32_t dstep = a.drop_c.enabled ? drop_step(a.drop_c) : 0u;   // before the loads whose latency hides the masks
    fetch_chunk(b_begin);

    // positives of this tile: offsets requested before the gather (their loads overlap the candidate rows')
    const int pos_lo = a.tile_ptr[blockIdx.x], pos_hi = a.tile_ptr[blockIdx.x + 1];
    const int pos_cached = min(pos_hi - pos_lo, POS_CACHE);

    // ---- candidate tile: gather, dropout, LDS; masked copy for dq_kernel ------------------------------------------------
    {
        const int n = n0 + r8;
        const bool valid = n < a.N;
        int64_t cid = 0;
        if (valid) cid = checked_row(a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n, a.n_table_rows, q8 ? nullptr : a.id_err);
        const float *row = a.E + cid * d;
        v4f v0[NOIT], v1[NOIT];
        if (vec_ok) {
            // branch-free: every thread issues its 2*NOIT 16-byte loads back to back (out-of-range pieces read the row's
            // first floats and are zeroed below), so the row's round trips to HBM overlap instead of queueing up
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = q8 + QG * it, k = 8 * o;
                const bool in0 = o < NO && valid && k < d, in1 = in0 && k + 4 < d;
                v0[it] = *reinterpret_cast<const v4f *>(row + (in0 ? k : 0));
                v1[it] = *reinterpret_cast<const v4f *>(row + (in1 ? k + 4 : 0));
            }
        } else {
#pragma unroll
            for (int it = 0; it < NOIT; ++it) {
                const int o = q8 + QG * it, k = 8 * o;
                v0[it] = (v4f){0.f, 0.f, 0.f, 0.f};
                v1[it] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (o < NO && valid && k < d) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (k + e < d) v0[it][e] = row[k + e];
                        if (k + 4 + e < d) v1[it][e] = row[k + 4 + e];
                    }
                }
            }
        }
        TL_STAMP_AT(40);   // gather loads issued
        // the keep bits depend on (candidate, octet, seed) only: computed while the rows are in flight
        uint32_t bits[NOIT];
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = q8 + QG * it, k = 8 * o;
            bits[it] = (o < NO && valid && k < d) ? 0xFFu : 0u;
            if (a.drop_c.enabled && bits[it]) bits[it] = drop_keep8<true>(a.drop_c, (uint32_t)(n + a.cand_col0), o, d, dstep);
        }
        __builtin_amdgcn_sched_barrier(0);
        // (after the mask arithmetic: vmcnt retires in order, so waiting for these loads waits for the rows as well)
        for (int i = tid; i < pos_cached; i += T64_THREADS)
            posc[i] = ((uint32_t)a.pos_row[pos_lo + i] << 6) | (uint32_t)(a.pos_col[pos_lo + i] - a.cand_col0 - n0);
        if (tid < 3 * BC64 * 2) ybits3[tid] = 0u;
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = q8 + QG * it, k = 8 * o;
            if (o < NO) {
                const float sc = a.drop_c.enabled ? a.drop_c.scale : 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[it][e] = (bits[it] >> e & 1u) && k + e < d ? v0[it][e] * sc : 0.f;
                    v1[it][e] = (bits[it] >> (4 + e) & 1u) && k + 4 + e < d ? v1[it][e] * sc : 0.f;
                }
                *reinterpret_cast<v4f *>(Cs + r8 * LDK + k) = v0[it];
                *reinterpret_cast<v4f *>(Cs + r8 * LDK + k + 4) = v1[it];
                keepb[r8 * KEEP_LD + o] = (uint8_t)bits[it];
                if (REGC && blockIdx.y == 0 && !a.loss_only) {        // the masked rows for dq_kernel, straight from the registers
                    float *cm = a.Cm + (size_t)(n0 + r8) * (16 * KB);  // (the tile's LDS is reused before a later chunk could)
                    *reinterpret_cast<v4f *>(cm + k) = v0[it];
                    *reinterpret_cast<v4f *>(cm + k + 4) = v1[it];
                }
            }
        }
    }
    TL_STAMP_AT(41);       // tile parked in LDS
    __syncthreads();
    TL_STAMP_AT(42);

    v4f dc[KB];                                      // dC[n = 16blk + 4s + i][k = grad col(kbi, c)], rows of half h
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) dc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
    const bool col_edge = n0 + NT64 > a.N;           // the last tile: candidates past N are masked out of the loss
    // masked candidate rows for dq_kernel: written from LDS one octet column group per chunk (chunks 1 .. NOIT), so that
    // neither the prologue's HBM burst -- every workgroup gathering at once -- nor a later chunk carries all 12 MB;
    // whatever is left when the sweep ends (fewer chunks than column groups) goes out in the write-back
    auto write_cm = [&](int it) {
        if (blockIdx.y != 0 || a.loss_only) return;
        float *cm = a.Cm + (size_t)(n0 + r8) * (16 * KB);
        const int o = q8 + QG * it, k = 8 * o;
        if (o < NO) {
            *reinterpret_cast<v4f *>(cm + k) = *reinterpret_cast<const v4f *>(Cs + r8 * LDK + k);
            *reinterpret_cast<v4f *>(cm + k + 4) = *reinterpret_cast<const v4f *>(Cs + r8 * LDK + k + 4);
        }
    };
    int cm_done = 0;
    // REGC: the lane's candidate operand (row 16blk + c, columns 16r + 4s .. + 3 of every round r) into registers and the
    // first query chunk parked -- then the tile's LDS is free for the odd chunks
    v4f breg[REGC ? KB : 1];
    if (REGC) {
#pragma unroll
        for (int r = 0; r < KB; ++r) breg[r] = *reinterpret_cast<const v4f *>(Cs + (16 * blk + c) * LDK + 16 * r + 4 * s);
        if (SHORTK) {                   // last round in the order 16 r + 4 j + s, j = 0, 1
            breg[KB - 1][0] = Cs[(16 * blk + c) * LDK + 16 * (KB - 1) + s];
            breg[KB - 1][1] = Cs[(16 * blk + c) * LDK + 16 * (KB - 1) + 4 + s];
        }
        cm_done = NOIT;                 // (written in the prologue)
#pragma unroll
        for (int it = 0; it < NQIT; ++it) {
            const int q = q8 + QG * it;
            if (q < NQ) *reinterpret_cast<v4f *>(Qs + r8 * LDK + 4 * q) = qreg[it];
        }
        if (b_begin + BC64 < b_end) fetch_chunk(b_begin + BC64);
    }

    // Label bits of a chunk (bit = candidate column of the tile, word = candidate half x batch row) from the tile's positives.
    // Three buffers in rotation: chunk i's bits are SET while chunk i-1's score product runs (they need no barrier of their
    // own: two lie between the set and the epilogue that reads them) and the buffer chunk i-1 used is cleared then too --
    // in the staging phase, between the two barriers of a chunk, nothing overlaps these dependent LDS round trips.
    auto set_label_bits = [&](int bb, uint32_t *yb) {
        for (int i = tid; i < pos_cached; i += T64_THREADS) {
            const uint32_t v = posc[i];
            const int row = (int)(v >> 6) - bb;
            if (row >= 0 && row < BC64) atomicOr(&yb[BC64 * ((v >> 5) & 1u) + row], 1u << (v & 31u));
        }
        for (int q = pos_lo + POS_CACHE + tid; q < pos_hi; q += T64_THREADS) {      // overflow: rare
            const int row = a.pos_row[q] - bb;
            const int col = a.pos_col[q] - a.cand_col0 - n0;
            if (row >= 0 && row < BC64) atomicOr(&yb[BC64 * (col >> 5) + row], 1u << (col & 31));
        }
    };
    set_label_bits(b_begin, ybits3);        // chunk 0 (the three buffers were cleared before the prologue's barrier)
    int par = 0;
    for (int b0 = b_begin; b0 < b_end; b0 += BC64, par = par == 2 ? 0 : par + 1) {
        uint32_t *ybits = ybits3 + par * (2 * BC64);
        // query chunk buffer of this chunk / of the next (REGC: the two LDS tiles alternate)
        float *Qc = REGC && (((b0 - b_begin) >> 6) & 1) ? Cs : Qs;
        float *Qn = REGC ? (Qc == Qs ? Cs : Qs) : Qs;
        TL_STAMP_AT(48 + ((b0 - b_begin) >> 6));   // chunk entered (the previous chunk's closing barrier passed)
        if (!REGC) {
            // ---- phase A: park the prefetched chunk, prefetch the next chunk -------------------------------------
#pragma unroll
            for (int it = 0; it < NQIT; ++it) {
                const int q = q8 + QG * it;
                if (q < NQ) *reinterpret_cast<v4f *>(Qs + r8 * LDK + 4 * q) = qreg[it];
            }
            TL_STAMP_AT(56 + ((b0 - b_begin) >> 6));   // query chunk parked
            if (b0 + BC64 < b_end) fetch_chunk(b0 + BC64);
            TL_STAMP_AT(72 + ((b0 - b_begin) >> 6));   // next chunk requested
        }
        TL_STAMP();   // [0] staged, before the barrier
        __syncthreads();   // REGC: this chunk's rows are parked (during the previous chunk) and the other buffer is free
        TL_STAMP();   // [1] start of score product
        {   // under the score product: the next chunk's label bits, the buffer after that cleared, one group of masked rows out
            const int pn = par == 2 ? 0 : par + 1, pc = pn == 2 ? 0 : pn + 1;
            if (tid < 2 * BC64) ybits3[pc * (2 * BC64) + tid] = 0u;
            if (b0 + BC64 < b_end) set_label_bits(b0 + BC64, ybits3 + pn * (2 * BC64));
            if (!REGC && b0 > b_begin && cm_done < NOIT) write_cm(cm_done++);
        }

        // ---- score blocks (rows 32h + 16rg + 4s + i, columns 16blk + c), rg = 0, 1 -----------------------------------
        v4f x0 = (v4f){0.f, 0.f, 0.f, 0.f}, x1 = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            const float *qa0 = Qc + (32 * h + c) * LDK + 4 * s;
            const float *qa1 = qa0 + 16 * LDK;
            const float *cb = Cs + (16 * blk + c) * LDK + 4 * s;
            constexpr int NRD = REGC ? 2 : 3;                    // ds_reads per round
            v4f a00 = *reinterpret_cast<const v4f *>(qa0), a01 = *reinterpret_cast<const v4f *>(qa1);
            v4f b0v = REGC ? breg[0] : *reinterpret_cast<const v4f *>(cb);
            v4f a10 = a00, a11 = a01, b1v = b0v;
            if (KB > 1) {
                a10 = *reinterpret_cast<const v4f *>(qa0 + 16);
                a11 = *reinterpret_cast<const v4f *>(qa1 + 16);
                b1v = REGC ? breg[1] : *reinterpret_cast<const v4f *>(cb + 16);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, KB > 1 ? 2 * NRD : NRD, 0);   // rounds 0 and 1 operands
#pragma unroll
            for (int r = 0; r < KB; ++r) {
                v4f a20 = a10, a21 = a11, b2v = b1v;
                const bool short_next = SHORTK && r + 2 == KB - 1;          // the round being requested is the short one
                if (r + 2 < KB) {
                    if (short_next) {
                        const float *q0 = Qc + (32 * h + c) * LDK + 16 * (KB - 1) + s, *q1 = q0 + 16 * LDK;
                        a20[0] = q0[0]; a20[1] = q0[4];
                        a21[0] = q1[0]; a21[1] = q1[4];
                    } else {
                        a20 = *reinterpret_cast<const v4f *>(qa0 + 16 * (r + 2));
                        a21 = *reinterpret_cast<const v4f *>(qa1 + 16 * (r + 2));
                    }
                    b2v = REGC ? breg[r + 2 < KB ? r + 2 : 0] : *reinterpret_cast<const v4f *>(cb + 16 * (r + 2));
                }
                constexpr int NJ_FULL = 4;
                const int nj = (SHORTK && r == KB - 1) ? 2 : NJ_FULL;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nj) {
                        x0 = mfma16(a00[j], b0v[j], x0);
                        x1 = mfma16(a01[j], b0v[j], x1);
                    }
                }
                a00 = a10; a01 = a11; b0v = b1v;
                a10 = a20; a11 = a21; b1v = b2v;
                if (short_next) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // ds_reads of round r+2
                else __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
                if (nj == 2) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // the MFMAs of round r
                else __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        }

        TL_STAMP();   // [2] end of score product
        // B operands (query rows) of the dC product's first step: requested now, consumed after the epilogue
        const float *qb = Qc + (32 * h + 4 * s) * LDK;
        v4f pb[KQ > 0 ? KQ : 1];
        float pr[KR > 0 ? KR : 1];
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) pb[kq] = *reinterpret_cast<const v4f *>(qb + 64 * kq + 4 * c);
#pragma unroll
        for (int r = 0; r < KR; ++r) pr[r] = qb[64 * KQ + 16 * r + c];

        // ---- loss epilogue: G = dLoss/dX / normalizer, kept in registers --------------------------------------
        v4f g4[2];
        {
            constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
            // label words of the wave's 2 x 4 rows, candidate half blk >> 1; the block's bit is 16 * (blk & 1) + c
            const uint32_t *yrow = ybits + BC64 * (blk >> 1) + 32 * h + 4 * s;
            const uint4 yw0 = *reinterpret_cast<const uint4 *>(yrow), yw1 = *reinterpret_cast<const uint4 *>(yrow + 16);
            const uint32_t yw[2][4] = {{yw0.x, yw0.y, yw0.z, yw0.w}, {yw1.x, yw1.y, yw1.z, yw1.w}};
            const int ybit = 16 * (blk & 1) + c;
            const bool edge = col_edge || b0 + BC64 > b_end;      // uniform: only the last tile / a partial last chunk
            const bool nvalid = n0 + 16 * blk + c < a.N;
            // BCE, LOGPROD: the log1p terms of the lane's 8 scores as ONE log of the product of (1 + e^-|x|) (each factor in
            // (1, 2]) -- v_log_f32 is a quarter-rate instruction and fp32 MFMA and VALU exclude each other on a SIMD, so the
            // epilogue's VALU time comes straight out of the MFMA time -- and G = sig * inv_norm - y * inv_norm as one fma.
            // cfg4 shard (KB = 16): 2197 -> 2142 us per range, cfg5 302 -> 295 us.  NOT at KB = 13 with the register-resident
            // operand: that instance sits at 256 registers and the two extra accumulators cost 18 more spilled dwords
            // (S-FB 72.9 -> 73.8 us; one log per FOUR scores and the old G arithmetic spill just as much), so it keeps one
            // log per score.
            constexpr bool LOGPROD = !(REGC && KB > 8);
            float lin = 0.f, prod = 1.f;
            const float gy_pos = -a.y_pos * a.inv_norm, gy_neg = -a.y_neg * a.inv_norm;
#pragma unroll
            for (int rg = 0; rg < 2; ++rg) {
                const v4f x = rg == 0 ? x0 : x1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float xv = x[i];
                    const bool pos = (yw[rg][i] >> ybit) & 1u;
                    float gg, l;
                    if (MODE == MODE_TRAIN_BCE) {
                        // BCEWithLogits: max(x,0) - x*y + log1p(exp(-|x|)); d/dx = sigmoid(x) - y
                        // v_exp_f32 / v_rcp_f32 / v_log_f32 (1 ulp each); 1 + e is in (1, 2]
                        const float y = pos ? a.y_pos : a.y_neg;
                        const float e = __builtin_amdgcn_exp2f(-fabsf(xv) * LOG2E);
                        float ope = 1.f + e;
                        const float rcp = __builtin_amdgcn_rcpf(ope);
                        const float sig = xv >= 0.f ? rcp : e * rcp;
                        if (LOGPROD) {
                            l = fmaxf(xv, 0.f) - xv * y;
                            if (edge && !(nvalid && b0 + 32 * h + 16 * rg + 4 * s + i < b_end)) { l = 0.f; ope = 1.f; }
                            lin += l;
                            prod *= ope;
                            g4[rg][i] = fmaf(sig, a.inv_norm, pos ? gy_pos : gy_neg);
                            continue;
                        }
                        l = fmaxf(xv, 0.f) - xv * y + __builtin_amdgcn_logf(ope) * LN2;
                        gg = sig - y;
                    } else {
                        // KLDiv(sum)(log_softmax(x), y), y in {0,1} unnormalised (trainer.py:99-101):
                        // loss = -sum_pos log_softmax; d/dx = softmax * sum_n y - y
                        const int b = min(b0 + 32 * h + 16 * rg + 4 * s + i, a.B - 1);
                        const float lsm = xv - a.row_lse[b];
                        l = pos ? -lsm : 0.f;
                        gg = __builtin_amdgcn_exp2f(lsm * LOG2E) * a.row_ysum[b] - (pos ? 1.f : 0.f);
                    }
                    // (a shard's last tile also sees the positives of the next shard's first columns: masked like padding)
                    if (edge) l = (nvalid && b0 + 32 * h + 16 * rg + 4 * s + i < b_end) ? l : 0.f;
                    lsum += l;
                    g4[rg][i] = gg * a.inv_norm;
                }
            }
            if (MODE == MODE_TRAIN_BCE && LOGPROD) lsum += lin + __builtin_amdgcn_logf(prod) * LN2;
        }
        if (REGC && b0 + BC64 < b_end) {
            // park the next chunk in the other buffer (its last readers finished before this chunk's barrier) and request the
            // chunk after it: by now the waves of a SIMD are a phase apart, so this runs beside the other wave's MFMAs
#pragma unroll
            for (int it = 0; it < NQIT; ++it) {
                const int q = q8 + QG * it;
                if (q < NQ) *reinterpret_cast<v4f *>(Qn + r8 * LDK + 4 * q) = qreg[it];
            }
            if (b0 + 2 * BC64 < b_end) fetch_chunk(b0 + 2 * BC64);
        }
        if (!a.loss_only) {
            // ---- G blocks -> HBM for dq_kernel: Gt[T][J][n_local][b_local], 4 consecutive batch rows per lane ------
            {
                const size_t blk_idx = (size_t)blockIdx.x * (a.Bpad >> 6) + (b0 >> 6);
                float *gdst = a.G + blk_idx * 4096 + (16 * blk + c) * 64 + 32 * h + 4 * s;
                *reinterpret_cast<v4f *>(gdst) = g4[0];
                *reinterpret_cast<v4f *>(gdst + 16) = g4[1];
            }
            // ---- dC += G^T . Q over this wave's 32 batch rows: A operands straight from g4 ------------------------
            // sub-step u = 4 rg + t, slot s  <->  batch row 32h + 16rg + 4s + t ; A = G[row][n = 16blk + c] = g4[rg][t]
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float av = g4[u >> 2][u & 3];
                v4f nb[KQ > 0 ? KQ : 1];
                float nr[KR > 0 ? KR : 1];
                if (u + 1 < 8) {
                    const float *brow = qb + (16 * ((u + 1) >> 2) + ((u + 1) & 3)) * LDK;
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) nb[kq] = *reinterpret_cast<const v4f *>(brow + 64 * kq + 4 * c);
#pragma unroll
                    for (int r = 0; r < KR; ++r) nr[r] = brow[64 * KQ + 16 * r + c];
                }
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dc[4 * kq + e] = mfma16(av, pb[kq][e], dc[4 * kq + e]);
                }
#pragma unroll
                for (int r = 0; r < KR; ++r) dc[4 * KQ + r] = mfma16(av, pr[r], dc[4 * KQ + r]);
                if (u + 1 < 8) {
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) pb[kq] = nb[kq];
#pragma unroll
                    for (int r = 0; r < KR; ++r) pr[r] = nr[r];
                    __builtin_amdgcn_sched_group_barrier(0x100, KQ + KR, 1);  // next step's ds_reads first
                }
                __builtin_amdgcn_sched_group_barrier(0x008, KB, 1);           // then this step's MFMAs
            }
        }
        TL_STAMP();   // [3] end of dC product
        if (!REGC || b0 + BC64 >= b_end) __syncthreads();   // (REGC: the next chunk's opening barrier is the only one it needs)
    }
    TL_STAMP();       // loop done

    // ---- write-back: the two row halves' partial dC are summed through LDS (h0 -> Qs; h1 += Qs -> Cs), then rows are
    //      masked with the cached dropout flags and added into dE ----------------------------------------------------
    if (cm_done < NOIT) {                  // short sweeps: the rest of the masked rows (uniform condition)
        for (; cm_done < NOIT; ++cm_done) write_cm(cm_done);
        __syncthreads();
    }
    if (!a.loss_only) {
        if (h == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *srow = Qs + (16 * blk + 4 * s + i) * LDK;
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq)
                    *reinterpret_cast<v4f *>(srow + 64 * kq + 4 * c) =
                        (v4f){dc[4 * kq][i], dc[4 * kq + 1][i], dc[4 * kq + 2][i], dc[4 * kq + 3][i]};
#pragma unroll
                for (int r = 0; r < KR; ++r) srow[64 * KQ + 16 * r + c] = dc[4 * KQ + r][i];
            }
        }
        __syncthreads();
        if (h == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float *srow = Qs + (16 * blk + 4 * s + i) * LDK;
                float *orow = Cs + (16 * blk + 4 * s + i) * LDK;
#pragma unroll
                for (int kq = 0; kq < KQ; ++kq) {
                    const v4f v = *reinterpret_cast<const v4f *>(srow + 64 * kq + 4 * c);
                    *reinterpret_cast<v4f *>(orow + 64 * kq + 4 * c) =
                        (v4f){dc[4 * kq][i] + v[0], dc[4 * kq + 1][i] + v[1], dc[4 * kq + 2][i] + v[2], dc[4 * kq + 3][i] + v[3]};
                }
#pragma unroll
                for (int r = 0; r < KR; ++r) orow[64 * KQ + 16 * r + c] = dc[4 * KQ + r][i] + srow[64 * KQ + 16 * r + c];
            }
        }
    }
    TL_STAMP_AT(43);       // partial sums combined
    {
        const double ls = wave_sum((double)lsum);
        if (lane == 0) red[w] = ls;
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) tot += red[i];
        a.loss_partial[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
    TL_STAMP_AT(44);       // loss partial out
    const int n = n0 + r8;
    if (!a.loss_only && n < a.N) {
        const int64_t cid = checked_row(a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n, a.n_table_rows, nullptr);
        float *drow = a.dE + cid * d;
        const bool exclusive = gridDim.y == 1 && a.cand_exclusive;    // one workgroup per entity row: plain stores
        // batch split over blockIdx.y: every workgroup stores ITS partial rows into a slab (plain 16-byte stores);
        // dc_reduce_kernel sums the slabs into dE
        float *srow = gridDim.y > 1 ? a.dC_slab + ((size_t)blockIdx.y * gridDim.x * NT64 + n) * (16 * KB) : nullptr;
#pragma unroll
        for (int it = 0; it < NOIT; ++it) {
            const int o = q8 + QG * it, k = 8 * o;
            if (o >= NO || k >= d) continue;
            v4f v[2];
            v[0] = *reinterpret_cast<const v4f *>(Cs + r8 * LDK + k);
            v[1] = *reinterpret_cast<const v4f *>(Cs + r8 * LDK + k + 4);
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
static size_t shmem64()
{
    using Cfg = Tile64Cfg<KB>;
    return (size_t)(NT64 + BC64) * Cfg::LDK * sizeof(float) + 3 * BC64 * 2 * sizeof(uint32_t) + 8 * sizeof(double) +
           NT64 * Cfg::KEEP_LD + POS_CACHE * sizeof(uint32_t);
}

template <int KB, int MODE>
static hipError_t launch64_t(const FusedArgs &a, dim3 grid, hipStream_t st)
{
    // candidate operand in registers + two query chunk buffers: 4 KB more registers per lane, so slot sizes up to 208
    static const bool regc_on = [] { const char *e = getenv("OKGE_TILE64_REGC"); return !e || atoi(e) != 0; }();
    static const bool shortk_on = [] { const char *e = getenv("OKGE_TILE64_SHORTK"); return !e || atoi(e) != 0; }();
    auto k = KB <= 13 && regc_on ? fused_tile64_kernel<KB, MODE, (KB <= 13)> : fused_tile64_kernel<KB, MODE, false>;
    // slot sizes 16 (KB - 1) + 1 .. + 8: the last contraction round needs two of its four MFMAs
    if (KB == 13 && regc_on && shortk_on && a.d > 16 * (KB - 1) && a.d <= 16 * (KB - 1) + 8)
        k = fused_tile64_kernel<KB, MODE, (KB == 13), (KB == 13)>;
    const size_t shmem = shmem64<KB>();
    static LdsOptIn lds_opt_in;
    if (hipError_t e = ensure_dynamic_lds(lds_opt_in, reinterpret_cast<const void *>(k), shmem); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(T64_THREADS), shmem, st, a);
    return hipGetLastError();
}

template <int KB>
static hipError_t launch64_m(int mode, const FusedArgs &a, dim3 grid, hipStream_t st)
{
    return mode == MODE_TRAIN_KL ? launch64_t<KB, MODE_TRAIN_KL>(a, grid, st) : launch64_t<KB, MODE_TRAIN_BCE>(a, grid, st);
}

// grid_x = number of 64-candidate tiles
hipError_t launch_fused64(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st)
{
    if (mode != MODE_TRAIN_BCE && mode != MODE_TRAIN_KL) return hipErrorInvalidValue;
    const dim3 grid(grid_x, grid_y);
    switch (a.KB) {
        case 4:  return launch64_m<4>(mode, a, grid, st);
        case 8:  return launch64_m<8>(mode, a, grid, st);
        case 13: return launch64_m<13>(mode, a, grid, st);
        case 16: { static const bool k64 = [] { const char *e = getenv("OKGE_TILE64K_D256"); return e && atoi(e) != 0; }();
                   return k64 ? launch_fused64k(mode, a, grid_x, grid_y, st) : launch64_m<16>(mode, a, grid, st); }
        case 32: return launch_fused64k(mode, a, grid_x, grid_y, st);      // okge_train64k.hip
        default: return hipErrorInvalidValue;
    }
}

}  // namespace okge
