// fused_tile64k_kernel<KB, MODE>: the fused score -> loss -> dCand tile kernel for slot sizes ABOVE 256 (KB = 32: d <= 512),
// one 8-wave workgroup per CU on 64 candidates x 32 batch rows per step.  Same contract as fused_tile64_kernel
// (okge_train64.hip): replaces the reference's encode_obj(candidates) + mm + cat + BCEWithLogits / log_softmax+KLDiv forward
// and the mm / sigmoid half of autograd's backward (openkge/model.py:198-229,268-274,455-510; openkge/trainer.py:75-106,234).
//
// Why another cut.  At d = 512 a 64-candidate tile is 128 KB and so is its gradient: the round-1/2 kernel for these sizes
// (fused_tile32_kernel<32>) therefore worked on 32 x 32 steps with BOTH tiles in LDS (2 x 66 KB, single-buffered), 8 waves
// splitting the contraction: 128 MFMAs per wave between barriers, three barriers and an un-overlapped staging phase per step,
// half-filled third round of workgroups at the DistMult d = 512 / N = 10 000 shape -- 0.435 of the fp32-MFMA peak
// (profiles/round3_cfg_S-DM_*).  Here the on-chip budget is spent differently:
//   * the CANDIDATE TILE LIVES IN REGISTERS.  Wave (blk = w & 3, ks = w >> 2) owns candidates 16 blk .. +15 and the
//     contraction / gradient columns 256 ks .. +255: the B operand of its whole score sweep is 16 float4 per lane (64
//     registers), read from global memory once in the prologue (dropout applied there, Philox keyed by candidate position as
//     everywhere); its share of the gradient, dC[16 candidates][256 columns], is another 64 registers -- no redundancy
//     between waves, so the write-back adds nothing up.
//   * LDS holds only QUERY CHUNKS: two buffers of 32 rows x 512 columns (2 x 66 KB).  Chunk i+1 is parked (from registers
//     that were loaded a chunk earlier) after a wave's epilogue while chunk i is still being multiplied: no staging phase.
//   * per chunk and wave: score partials of the two 16 x 16 blocks (32 rows x 16 candidates) over its column half = 128
//     MFMAs (A operand one ds_read_b128 per block and round, B from registers), exchange of the partial blocks with the
//     partner wave (w ^ 4, same SIMD) through LDS -- the one extra barrier --, the loss epilogue on the summed block (both
//     partners compute it: a + b == b + a bit for bit, so they hold the same G), then dC += G^T . Q over the 32 rows for
//     its 256 columns = 128 MFMAs with G as the A operand straight from the epilogue's registers.
//     256 MFMAs per wave and chunk between two barriers (the 32 x 32 cut: 128 between three).
// G leaves for dq_kernel from the ks = 0 waves as 64 x 64 transposed blocks, the masked candidate rows `Cm` from the
// prologue's registers.
//
// WORK DISTRIBUTION.  The shapes these slot sizes come with have FEW candidate tiles (DistMult d = 512 on N = 10 000
// sampled candidates: 157 tiles for 256 CUs) and a fixed cost per (tile, workgroup) of ~1.8 chunks (gather + Philox +
// write-back), so neither "one workgroup per tile" (157 of 256 CUs busy) nor an even batch split over blockIdx.y (471
// workgroups = 1.84 rounds, the candidate gather repeated three times) fills the chip: measured 161 / 177 / 147 / 166 us for
// splits 1 / 2 / 3 / 4 = exactly rounds x (16 us + 9.1 us per chunk).  With a.sk_tiles > 0 the launch is therefore
// STREAM-K shaped: the (tile, chunk) units, tile-major, are cut into gridDim.x EQUAL contiguous runs (one workgroup per
// CU), workgroup p owns units [p U / P, (p + 1) U / P) and walks them as segments (tile, chunk range).  A segment that
// covers its whole tile stores the candidate gradient itself; the others store partial rows into slab 2 p (the run's first
// segment) or 2 p + 1 (its last) and dc_reduce_streamk_kernel adds a tile's partials up -- every workgroup pays at most two
// fixed costs and none waits for another (no flags, no spinning: the reduction is a separate launch).
// With a.sk_tiles == 0 the grid is (tiles, batch splits) as for the other tile kernels.
#include <cstdio>
#include <cstdlib>

#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

constexpr int NTK = 64, BCK = 32, TK_THREADS = 512;

template <int KB> struct Tile64kCfg {
    static constexpr int LDK = lds_ld(16 * KB);
    static constexpr int NO = 2 * KB;                         // 8-column octets per row
    static constexpr int KEEP_LD = NO < 32 ? 32 : NO;         // keep-flag bytes per row
    static constexpr int KBW = KB / 2;                        // 16-column rounds of one wave's column half
    static constexpr int KQW = KBW / 4;                       // 64-column quads of one wave's gradient columns
};

template <int KB, int MODE>
__global__ __launch_bounds__(TK_THREADS, 2) void fused_tile64k_kernel(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Cfg = Tile64kCfg<KB>;
    constexpr int LDK = Cfg::LDK, NO = Cfg::NO, KEEP_LD = Cfg::KEEP_LD, KBW = Cfg::KBW, KQW = Cfg::KQW;
    constexpr int D16 = 16 * KB, KH = 16 * KBW;                 // padded slot size, columns of one half
    constexpr int NQ = 4 * KB, QG = 16, NQIT = NQ / QG;         // float4 per row; staging: 16 column groups per row
    static_assert(KB % 8 == 0, "the column split needs whole quads of 16-column blocks per wave");
    static_assert(MODE == MODE_TRAIN_BCE || MODE == MODE_TRAIN_KL || MODE == MODE_SCORE || MODE == MODE_STATS || MODE == MODE_COUNT,
                  "training kernel, or the score / statistics / counting sweep");
    // MODE_SCORE / MODE_STATS: the same sweep stops after the score blocks and writes X, or per (16-candidate block, row) the
    // (max, sum-exp) pairs of the KL loss' log-softmax (kl_row_lse_kernel merges them).
    // MODE_COUNT (fused evaluation, round 4): the score product runs with its operands SWAPPED -- the result block is X^T, lane
    // (c, s) holds row c's scores of candidates 4 s .. 4 s + 3 -- so that a row's comparison with its answer groups' true scores
    // is lane-local; the chunk's packed {#greater | #equal << 16} counters are summed in LDS (the training modes' keep-flag /
    // positives cache is free here) and leave as one coalesced store per (tile, chunk), as in fused_tile_kernel<count>.
    // Same products in the same k order as MODE_SCORE: the counts equal those of the materialised scores bit for bit.
    constexpr bool TRAIN = MODE == MODE_TRAIN_BCE || MODE == MODE_TRAIN_KL;
    const int d = a.d;
    float *Qb = reinterpret_cast<float *>(smem);                              // [2][32][LDK]  (end: [64][LDK] gradient stage)
    v4f *xs = reinterpret_cast<v4f *>(Qb + 2 * BCK * LDK);                    // [8 waves][2 row groups][64 lanes]
    uint32_t *ybits3 = reinterpret_cast<uint32_t *>(xs + 8 * 2 * 64);         // [3][2 candidate halves][32 rows]
    double *red = reinterpret_cast<double *>(ybits3 + 3 * 2 * BCK);           // [8]
    uint8_t *keepb = reinterpret_cast<uint8_t *>(red + 8);                    // [64][KEEP_LD] keep flags of the tile
    uint32_t *posc = reinterpret_cast<uint32_t *>(keepb + NTK * KEEP_LD);     // [POS_CACHE] (row << 6 | col)
    uint32_t *cnt = reinterpret_cast<uint32_t *>(keepb);                      // MODE_COUNT: [CNT_CAP] packed counters of a chunk's groups
    constexpr int CNT_CAP = (NTK * KEEP_LD) / 4 + POS_CACHE;

    const bool vec_ok = (d & 3) == 0;
    const uint32_t dstep = a.drop_c.enabled ? drop_step(a.drop_c) : 0u;   // before the loads whose latency hides the masks

    // ---- this workgroup's run of (tile, chunk) units ---------------------------------------------------------------------
    const bool sk = a.sk_tiles > 0;
    const int J = (a.B + BCK - 1) / BCK;                        // chunks per tile
    int u = 0, u_end = 0, u_first = 0;                          // (units of one launch fit 31 bits: tiles x chunks of ONE range)
    if (sk) {
        const int64_t U = (int64_t)a.sk_tiles * J;
        u_first = u = (int)(U * blockIdx.x / gridDim.x);
        u_end = (int)(U * (blockIdx.x + 1) / gridDim.x);
    }
    float lsum = 0.f;
#pragma nounroll
    for (int seg = 0;; ++seg) {
    int tile, b_begin, b_end;
    bool first_rows;                     // the segment holding a tile's first rows hands the masked candidate rows to dq_kernel
    float *slab_rows = nullptr;          // partial candidate gradients go here (row n_local); nullptr: straight into dE
    if (sk) {
        if (u >= u_end) break;
        tile = u / J;
        const int j0 = u - tile * J, j1 = min(J, u_end - tile * J);
        b_begin = BCK * j0;
        b_end = min(a.B, BCK * j1);
        first_rows = j0 == 0;
        if (!(j0 == 0 && j1 == J))
            slab_rows = a.dC_slab + ((size_t)2 * blockIdx.x + (u == u_first ? 0 : 1)) * NTK * D16;
        u = tile * J + j1;
    } else {
        if (seg > 0) break;
        tile = blockIdx.x;
        b_begin = blockIdx.y * a.b_per_block;
        b_end = min(a.B, b_begin + a.b_per_block);
        first_rows = blockIdx.y == 0;
        if (gridDim.y > 1) slab_rows = a.dC_slab + ((size_t)blockIdx.y * gridDim.x * NTK + (size_t)tile * NTK) * D16;
    }
    const int n0 = tile * NTK;
    if (seg > 0) __syncthreads();        // the previous segment's write-back has read the LDS this one fills
    // the thread's roles, re-derived per segment from an opaque copy of its id: everything below -- a few hundred address
    // values of the unrolled loops -- is invariant across segments, and hoisted out of this loop it would have to live in
    // registers the chunk loop needs (measured: 364 VGPRs + 470 SGPRs spilled without this)
    int tid_o = threadIdx.x;
    asm volatile("" : "+v"(tid_o));
    const int tid = tid_o, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, s = lane >> 4;
    const int blk = w & 3, ks = w >> 2;
    const int r16 = tid >> 4, q16 = tid & 15;                   // staging role: row r16 of the chunk, column group q16

    // ---- register-staged query chunk (32 rows x 16*KB): thread holds float4 columns q16 + 16*it of row r16 ------------
    v4f qreg[NQIT];
    auto fetch_chunk = [&](int b0) {
        const int b = b0 + r16;
        const float *src = a.Q + (size_t)b * a.ldq;
#pragma unroll
        for (int it = 0; it < NQIT; ++it)
            qreg[it] = (b < b_end) ? *reinterpret_cast<const v4f *>(src + 4 * (q16 + QG * it)) : (v4f){0.f, 0.f, 0.f, 0.f};
    };
    auto park_chunk = [&](float *dst) {
#pragma unroll
        for (int it = 0; it < NQIT; ++it) *reinterpret_cast<v4f *>(dst + r16 * LDK + 4 * (q16 + QG * it)) = qreg[it];
    };
    fetch_chunk(b_begin);
    // MODE_COUNT: the group range of the lane's row (row 16 ks + c of the chunk) one chunk ahead, its first RK_PRE true scores
    // before the score product: the counting loop does not wait on dependent global loads
    constexpr int RK_PRE = 4;
    int64_t rk_glo = 0, rk_glo_n = 0, rk_c0 = 0, rk_c1 = 0, rk_c0_n = 0, rk_c1_n = 0;
    int rk_ng = 0, rk_ng_n = 0;
    float rk_t[RK_PRE];
    auto rk_fetch_rows = [&](int b0) {
        if (MODE != MODE_COUNT) return;
        const int b = b0 + 16 * ks + c;
        rk_glo_n = 0; rk_ng_n = 0;
        if (b < b_end) {
            rk_glo_n = a.rk_row_ptr[b];
            rk_ng_n = (int)(a.rk_row_ptr[b + 1] - rk_glo_n);
        }
        rk_c0_n = a.rk_row_ptr[min(b0, b_end)];
        rk_c1_n = a.rk_row_ptr[min(b0 + BCK, b_end)];
    };
    rk_fetch_rows(b_begin);

    const int pos_lo = TRAIN ? a.tile_ptr[tile] : 0, pos_hi = TRAIN ? a.tile_ptr[tile + 1] : 0;
    const int pos_cached = min(pos_hi - pos_lo, POS_CACHE);

    // ---- candidate operand: lane (c, s) of wave (blk, ks) keeps C[16 blk + c][256 ks + 16 r + 4 s .. + 3], r < KBW ----------
    v4f breg[KBW];
    {
        const int nl = 16 * blk + c, n = n0 + nl;
        const bool valid = n < a.N;
        int64_t cid = 0;
        if (valid) cid = checked_row(a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n, a.n_table_rows,
                                     (s == 0 && ks == 0) ? a.id_err : nullptr);
        const float *row = a.E + cid * d;
        if (vec_ok) {
            // branch-free: all 16 loads back to back (out-of-range pieces read the row's first floats and are zeroed below)
#pragma unroll
            for (int r = 0; r < KBW; ++r) {
                const int k = KH * ks + 16 * r + 4 * s;
                breg[r] = *reinterpret_cast<const v4f *>(row + ((valid && k < d) ? k : 0));
            }
        } else {
#pragma unroll
            for (int r = 0; r < KBW; ++r) {
                const int k = KH * ks + 16 * r + 4 * s;
                breg[r] = (v4f){0.f, 0.f, 0.f, 0.f};
                if (valid) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (k + e < d) breg[r][e] = row[k + e];
                }
            }
        }
        // keep bits: one Philox call covers the octet that two neighbouring column quads share (lanes s and s ^ 1 of a
        // candidate): each of the two computes the octets of every other round and they swap -- 8 calls per lane, not 16.
        // The computing lane also files the octet's flags for the write-back.
        uint32_t nib[KBW];
#pragma unroll
        for (int r = 0; r < KBW; ++r) nib[r] = 0xFu;
        if (a.drop_c.enabled) {
            uint32_t mine[KBW / 2];
#pragma unroll
            for (int j = 0; j < KBW / 2; ++j) {
                const int r = 2 * j + (s & 1);                                   // the rounds this lane computes
                const int o = (KH * ks + 16 * r + 4 * s) >> 3;
                mine[j] = (valid && 8 * o < d) ? drop_keep8<true>(a.drop_c, (uint32_t)(n + a.cand_col0), o, d, dstep) : 0u;
                if (TRAIN) keepb[nl * KEEP_LD + o] = (uint8_t)mine[j];
            }
#pragma unroll
            for (int j = 0; j < KBW / 2; ++j) {
                const uint32_t theirs = (uint32_t)__shfl_xor((int)mine[j], 16);  // lane s ^ 1: round 2 j + (~s & 1), same octet row
                nib[2 * j + (s & 1)] = (mine[j] >> (4 * (s & 1))) & 0xFu;
                nib[2 * j + ((s & 1) ^ 1)] = (theirs >> (4 * (s & 1))) & 0xFu;
            }
        }
        const float sc = a.drop_c.enabled ? a.drop_c.scale : 1.f;
        float *cm = (TRAIN && first_rows && !a.loss_only) ? a.Cm + (size_t)n * D16 + KH * ks + 4 * s : nullptr;
#pragma unroll
        for (int r = 0; r < KBW; ++r) {
            const int k = KH * ks + 16 * r + 4 * s;
#pragma unroll
            for (int e = 0; e < 4; ++e) breg[r][e] = (valid && k + e < d && (nib[r] >> e & 1u)) ? breg[r][e] * sc : 0.f;
            if (cm) *reinterpret_cast<v4f *>(cm + 16 * r) = breg[r];            // masked rows for dq_kernel (padding rows: 0)
        }
    }
    for (int i = tid; i < pos_cached; i += TK_THREADS)
        posc[i] = ((uint32_t)a.pos_row[pos_lo + i] << 6) | (uint32_t)(a.pos_col[pos_lo + i] - a.cand_col0 - n0);
    if (tid < 3 * 2 * BCK) ybits3[tid] = 0u;
    park_chunk(Qb);
    if (b_begin + BCK < b_end) fetch_chunk(b_begin + BCK);
    __syncthreads();

    v4f dc[KBW];                         // dC[n = 16 blk + 4 s + i][k = 256 ks + 64 kq + 4 c + e] in dc[4 kq + e][i]
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) dc[kb] = (v4f){0.f, 0.f, 0.f, 0.f};
    const bool col_edge = n0 + NTK > a.N;

    auto set_label_bits = [&](int bb, uint32_t *yb) {
        for (int i = tid; i < pos_cached; i += TK_THREADS) {
            const uint32_t v = posc[i];
            const int row = (int)(v >> 6) - bb;
            if (row >= 0 && row < BCK) atomicOr(&yb[BCK * ((v >> 5) & 1u) + row], 1u << (v & 31u));
        }
        for (int q = pos_lo + POS_CACHE + tid; q < pos_hi; q += TK_THREADS) {      // overflow: rare
            const int row = a.pos_row[q] - bb;
            const int col = a.pos_col[q] - a.cand_col0 - n0;
            if (row >= 0 && row < BCK) atomicOr(&yb[BCK * (col >> 5) + row], 1u << (col & 31));
        }
    };
    if (TRAIN) set_label_bits(b_begin, ybits3);     // (chunk 0: set after the clear above, read after the chunk's mid barrier)

    int par = 0, buf = 0;
    for (int b0 = b_begin; b0 < b_end; b0 += BCK, par = par == 2 ? 0 : par + 1, buf ^= 1) {
        const uint32_t *ybits = ybits3 + par * (2 * BCK);
        const float *Qc = Qb + buf * (BCK * LDK);
        float *Qn = Qb + (buf ^ 1) * (BCK * LDK);
        if (b0 > b_begin) __syncthreads();   // chunk parked by everyone (during the previous chunk); the other buffer is free
        if (MODE == MODE_COUNT) {
            rk_glo = rk_glo_n; rk_ng = rk_ng_n;
#pragma unroll
            for (int jj = 0; jj < RK_PRE; ++jj)
                rk_t[jj] = jj < rk_ng ? a.rk_true[rk_glo + jj] : __builtin_nanf("");      // NaN never compares true
            rk_c0 = rk_c0_n; rk_c1 = rk_c1_n;
            if (b0 + BCK < b_end) rk_fetch_rows(b0 + BCK);
            if (a.rk_slab && rk_c1 - rk_c0 <= CNT_CAP)     // (read again after the chunk's mid barrier)
                for (int k = tid; k < (int)(rk_c1 - rk_c0); k += TK_THREADS) cnt[k] = 0u;
        }
        if (TRAIN) {   // under the score product: the next chunk's label bits, the buffer after that cleared
            const int pn = par == 2 ? 0 : par + 1, pc = pn == 2 ? 0 : pn + 1;
            if (tid < 2 * BCK) ybits3[pc * (2 * BCK) + tid] = 0u;
            if (b0 + BCK < b_end) set_label_bits(b0 + BCK, ybits3 + pn * (2 * BCK));
        }

        // ---- partial score blocks over this wave's column half: rows 16 rg + 4 s + i, candidates 16 blk + c ---------------
        v4f x0 = (v4f){0.f, 0.f, 0.f, 0.f}, x1 = (v4f){0.f, 0.f, 0.f, 0.f};
        {
            const float *qa0 = Qc + c * LDK + KH * ks + 4 * s;
            const float *qa1 = qa0 + 16 * LDK;
            // operands one round ahead: a round is 8 MFMAs = 256 cycles of this wave alone, twice an LDS round trip
            v4f a00 = *reinterpret_cast<const v4f *>(qa0), a01 = *reinterpret_cast<const v4f *>(qa1);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // round 0 operands
#pragma unroll
            for (int r = 0; r < KBW; ++r) {
                v4f a10 = a00, a11 = a01;
                if (r + 1 < KBW) {
                    a10 = *reinterpret_cast<const v4f *>(qa0 + 16 * (r + 1));
                    a11 = *reinterpret_cast<const v4f *>(qa1 + 16 * (r + 1));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == MODE_COUNT) {           // X^T blocks: lane (c, s) = row c (+ 16), candidates 4 s + i
                        x0 = mfma16(breg[r][j], a00[j], x0);
                        x1 = mfma16(breg[r][j], a01[j], x1);
                    } else {
                        x0 = mfma16(a00[j], breg[r][j], x0);
                        x1 = mfma16(a01[j], breg[r][j], x1);
                    }
                }
                a00 = a10; a01 = a11;
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // ds_reads of round r+1
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // 8 MFMA   (round r)
            }
        }
        // ---- the partner's partial blocks (the other column half) through LDS ------------------------------------------------
        xs[(w * 2 + 0) * 64 + lane] = x0;
        xs[(w * 2 + 1) * 64 + lane] = x1;
        __syncthreads();
        x0 += xs[((w ^ 4) * 2 + 0) * 64 + lane];             // a + b == b + a: both partners hold the same bits
        x1 += xs[((w ^ 4) * 2 + 1) * 64 + lane];
        if (MODE == MODE_COUNT) {
            // partner ks takes row group ks: row b0 + 16 ks + c against candidates n0 + 16 blk + 4 s + i; the row's 64
            // candidates of this tile sit in 16 lanes (s = 0 .. 3 of the four waves blk = 0 .. 3 of this ks)
            const v4f xv = ks == 0 ? x0 : x1;
            float xm[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xm[i] = n0 + 16 * blk + 4 * s + i < a.N ? xv[i] : -INFINITY;
            const int64_t gc_lo = rk_c0, gc_hi = rk_c1, g_lo = rk_glo;
            const int ng = rk_ng;
            const bool in_lds = a.rk_slab && gc_hi - gc_lo <= CNT_CAP;
            uint32_t *slab_row = a.rk_slab ? a.rk_slab + (size_t)tile * a.rk_ngroups : nullptr;
            auto count_pass = [&](bool add) {
                for (int j = 0; __builtin_amdgcn_ballot_w64(j < ng) != 0; ++j) {
                    float t = j < ng && j >= RK_PRE ? a.rk_true[g_lo + j] : __builtin_nanf("");
#pragma unroll
                    for (int jj = 0; jj < RK_PRE; ++jj) t = j == jj ? rk_t[jj] : t;
                    int pk = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) pk += (xm[i] > t ? 1 : 0) + (xm[i] == t ? 65536 : 0);
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
                for (int k = tid; k < (int)(gc_hi - gc_lo); k += TK_THREADS) slab_row[gc_lo + k] = cnt[k];
            } else {
                // more groups in the chunk than the LDS counters hold (or the atomics path of very large batches): the four
                // waves that share a row take turns -- the first stores, the others add
#pragma unroll 1
                for (int turn = 0; turn < 4; ++turn) {
                    if (blk == turn) count_pass(turn > 0);
                    __syncthreads();
                }
            }
            if (b0 + BCK < b_end) {
                park_chunk(Qn);
                if (b0 + 2 * BCK < b_end) fetch_chunk(b0 + 2 * BCK);
            }
            continue;
        }
        if (MODE == MODE_SCORE) {
            // scores out: partner ks takes row group ks (rows 16 ks + 4 s + i, candidate 16 blk + c: 64-byte runs per row)
            const v4f x = ks == 0 ? x0 : x1;
            const int n = n0 + 16 * blk + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = b0 + 16 * ks + 4 * s + i;
                if (b < b_end && n < a.N) a.X[(size_t)b * a.ldx + n] = x[i];
            }
            if (b0 + BCK < b_end) {
                park_chunk(Qn);
                if (b0 + 2 * BCK < b_end) fetch_chunk(b0 + 2 * BCK);
            }
            continue;
        }
        if (MODE == MODE_STATS) {
            // per row: max and sum-exp over this wave's 16 candidates -> stats[(4 tile + blk)][row]; partner ks takes row group ks
            const v4f x = ks == 0 ? x0 : x1;
            const bool nvalid = n0 + 16 * blk + c < a.N;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float m = nvalid ? x[i] : -INFINITY;
                m = row16_max(m);
                float se = nvalid ? __expf(x[i] - m) : 0.f;
                se = row16_sum(se);
                const int b = b0 + 16 * ks + 4 * s + i;
                if (c == 0 && b < b_end)
                    reinterpret_cast<float2 *>(a.stats)[(size_t)(4 * tile + blk) * a.Bpad + b] = make_float2(m, se);
            }
            if (b0 + BCK < b_end) {
                park_chunk(Qn);
                if (b0 + 2 * BCK < b_end) fetch_chunk(b0 + 2 * BCK);
            }
            continue;
        }

        // B operands (query rows) of the dC product's first step: requested now, consumed after the epilogue
        const float *qb = Qc + (4 * s) * LDK + KH * ks + 4 * c;
        v4f pb[KQW];
#pragma unroll
        for (int kq = 0; kq < KQW; ++kq) pb[kq] = *reinterpret_cast<const v4f *>(qb + 64 * kq);

        // ---- loss epilogue: G = dLoss/dX / normalizer, kept in registers -------------------------------------------------
        v4f g4[2];
        {
            constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
            const uint32_t *yrow = ybits + BCK * (blk >> 1) + 4 * s;
            const uint4 yw0 = *reinterpret_cast<const uint4 *>(yrow), yw1 = *reinterpret_cast<const uint4 *>(yrow + 16);
            const uint32_t yw[2][4] = {{yw0.x, yw0.y, yw0.z, yw0.w}, {yw1.x, yw1.y, yw1.z, yw1.w}};
            const int ybit = 16 * (blk & 1) + c;
            const bool edge = col_edge || b0 + BCK > b_end;       // uniform: only the last tile / a partial last chunk
            const bool nvalid = n0 + 16 * blk + c < a.N;
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
                        const float y = pos ? a.y_pos : a.y_neg;
                        const float e = __builtin_amdgcn_exp2f(-fabsf(xv) * LOG2E);
                        const float ope = 1.f + e;
                        const float rcp = __builtin_amdgcn_rcpf(ope);
                        const float sig = xv >= 0.f ? rcp : e * rcp;
                        l = fmaxf(xv, 0.f) - xv * y + __builtin_amdgcn_logf(ope) * LN2;
                        gg = sig - y;
                    } else {
                        // KLDiv(sum)(log_softmax(x), y), y in {0,1} unnormalised (trainer.py:99-101)
                        const int b = min(b0 + 16 * rg + 4 * s + i, a.B - 1);
                        const float lsm = xv - a.row_lse[b];
                        l = pos ? -lsm : 0.f;
                        gg = __builtin_amdgcn_exp2f(lsm * LOG2E) * a.row_ysum[b] - (pos ? 1.f : 0.f);
                    }
                    if (edge) l = (nvalid && b0 + 16 * rg + 4 * s + i < b_end) ? l : 0.f;
                    lsum += l;                                   // (counted by the ks == 0 partner only, below)
                    g4[rg][i] = gg * a.inv_norm;
                }
            }
        }
        if (b0 + BCK < b_end) {
            // park the next chunk in the other buffer (its last readers finished before this chunk's opening barrier) and
            // request the chunk after it
            park_chunk(Qn);
            if (b0 + 2 * BCK < b_end) fetch_chunk(b0 + 2 * BCK);
        }
        if (!a.loss_only) {
            // ---- G blocks -> HBM for dq_kernel: Gt[T][J][n_local][b_local], 4 consecutive batch rows per lane ------
            if (ks == 0) {
                const size_t blk_idx = (size_t)tile * (a.Bpad >> 6) + (b0 >> 6);
                float *gdst = a.G + blk_idx * 4096 + (16 * blk + c) * 64 + 32 * ((b0 >> 5) & 1) + 4 * s;
                *reinterpret_cast<v4f *>(gdst) = g4[0];
                *reinterpret_cast<v4f *>(gdst + 16) = g4[1];
            }
            // ---- dC += G^T . Q over the chunk's 32 rows, this wave's 256 columns: A operands straight from g4 -------
            // sub-step u = 4 rg + t, slot s  <->  batch row 16 rg + 4 s + t ; A = G[row][n = 16 blk + c] = g4[rg][t]
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float av = g4[u >> 2][u & 3];
                const float *brow = qb + (16 * ((u + 1) >> 2) + ((u + 1) & 3)) * LDK;
#pragma unroll
                for (int kq = 0; kq < KQW; ++kq) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dc[4 * kq + e] = mfma16(av, pb[kq][e], dc[4 * kq + e]);
                    // the quad's operand of the NEXT sub-step goes into the registers these four MFMAs have just read
                    if (u + 1 < 8) pb[kq] = *reinterpret_cast<const v4f *>(brow + 64 * kq);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
                    if (u + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
                }
            }
        }
    }
    if (!TRAIN) continue;
    __syncthreads();

    // ---- write-back: every wave stages ITS (16 candidates x 256 columns) of dC into LDS (the two chunk buffers are one
    //      [64][LDK] block), then rows are masked with the cached dropout flags and stored ------------------------------------
    if (!a.loss_only) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float *srow = Qb + (16 * blk + 4 * s + i) * LDK + KH * ks + 4 * c;
#pragma unroll
            for (int kq = 0; kq < KQW; ++kq)
                *reinterpret_cast<v4f *>(srow + 64 * kq) = (v4f){dc[4 * kq][i], dc[4 * kq + 1][i], dc[4 * kq + 2][i], dc[4 * kq + 3][i]};
        }
    }
    __syncthreads();
    const int r8 = tid >> 3, q8 = tid & 7;                       // write-back role: row r8 of the tile, octets q8 + 8 it
    const int n = n0 + r8;
    if (!a.loss_only && n < a.N) {
        const int64_t cid = checked_row(a.cand_ids ? (int64_t)a.cand_ids[n] : (int64_t)a.cand_first + n, a.n_table_rows, nullptr);
        float *drow = a.dE + cid * d;
        const bool exclusive = !slab_rows && a.cand_exclusive;    // one workgroup per entity row: plain stores
        float *srow = slab_rows ? slab_rows + (size_t)r8 * D16 : nullptr;
#pragma unroll
        for (int it = 0; it < NO / 8; ++it) {
            const int o = q8 + 8 * it, k = 8 * o;
            if (k >= d) continue;
            v4f v[2];
            v[0] = *reinterpret_cast<const v4f *>(Qb + r8 * LDK + k);
            v[1] = *reinterpret_cast<const v4f *>(Qb + r8 * LDK + k + 4);
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
    }   // segments

    if (!TRAIN) return;
    // ---- this workgroup's loss partial (all its segments; the ks == 0 partner of every pair counted) -------------------------
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    {
        const double ls = wave_sum((double)((w >> 2) == 0 ? lsum : 0.f));
        if (lane == 0) red[w] = ls;
    }
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) tot += red[i];
        a.loss_partial[sk ? (size_t)blockIdx.x : (size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
}

// ---- launcher ---------------------------------------------------------------------------------------------------
template <int KB>
static size_t shmem64k()
{
    using Cfg = Tile64kCfg<KB>;
    return (size_t)2 * BCK * Cfg::LDK * sizeof(float) + 8 * 2 * 64 * sizeof(v4f) + 3 * 2 * BCK * sizeof(uint32_t) +
           8 * sizeof(double) + NTK * Cfg::KEEP_LD + POS_CACHE * sizeof(uint32_t);
}

template <int KB, int MODE>
static hipError_t launch64k_t(const FusedArgs &a, dim3 grid, hipStream_t st)
{
    auto k = fused_tile64k_kernel<KB, MODE>;
    const size_t shmem = shmem64k<KB>();
    static LdsOptIn lds_opt_in;
    if (hipError_t e = ensure_dynamic_lds(lds_opt_in, reinterpret_cast<const void *>(k), shmem); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, grid, dim3(TK_THREADS), shmem, st, a);
    return hipGetLastError();
}

// grid_x = number of 64-candidate tiles (or workgroups of a stream-K launch), slot sizes above 256 (KB = 32); MODE_SCORE: the
// score sweep of okge_score_prefixes / okge_score_queries / `all_outputs` on the same layout
hipError_t launch_fused64k(int mode, const FusedArgs &a, int grid_x, int grid_y, hipStream_t st)
{
    const dim3 grid(grid_x, grid_y);
    if (a.KB == 16 && (mode == MODE_TRAIN_BCE || mode == MODE_TRAIN_KL))      // experiment (OKGE_TILE64K_D256=1): d <= 256 on this layout
        return mode == MODE_TRAIN_KL ? launch64k_t<16, MODE_TRAIN_KL>(a, grid, st) : launch64k_t<16, MODE_TRAIN_BCE>(a, grid, st);
    if ((mode != MODE_TRAIN_BCE && mode != MODE_TRAIN_KL && mode != MODE_SCORE && mode != MODE_STATS && mode != MODE_COUNT) || a.KB != 32)
        return hipErrorInvalidValue;
    if (mode == MODE_SCORE) return launch64k_t<32, MODE_SCORE>(a, grid, st);
    if (mode == MODE_COUNT) return launch64k_t<32, MODE_COUNT>(a, grid, st);
    if (mode == MODE_STATS) return launch64k_t<32, MODE_STATS>(a, grid, st);
    return mode == MODE_TRAIN_KL ? launch64k_t<32, MODE_TRAIN_KL>(a, grid, st) : launch64k_t<32, MODE_TRAIN_BCE>(a, grid, st);
}

}  // namespace okge
