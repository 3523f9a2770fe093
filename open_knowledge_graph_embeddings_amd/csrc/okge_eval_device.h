// Device code of the fused evaluation's SIDE work -- the point scores before a tile sweep and the ranks + meters after it
// (okge_evaluate_fused*), 256-thread workgroups: one per batch row of the points, one per four answer groups of the ranks.
// (Tried and dropped in round 2: running this work in spare workgroups of the sweep kernel, a wave or 16 lanes per unit.
//  It is a chain of dependent loads; beside the sweep's memory traffic a unit took 40+ us and the sweep launch grew from
//  45 to 65 us.  As its own launch between two sweeps it costs 11 us.)
#pragma once
#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

struct RowSrc {
    int64_t ent, rel;   // ent: row in the LOCAL entity table (global id - ent_lo); valid only if owned
    uint32_t pos;
    bool sp, owned;
};

// Entity rows are sharded by id: this rank holds global ids [ent_lo, ent_hi) as local rows 0 .. ent_hi-ent_lo-1.
// (leader: the one thread of the row's workers that reports a bad id)
__device__ __forceinline__ RowSrc row_source(const PrefixDev &p, int b, bool leader)
{
    RowSrc r;
    int64_t gid;
    if (b < p.n_po) {
        r.rel = p.po_rel[b]; gid = p.po_obj[b]; r.pos = (uint32_t)b; r.sp = false;
    } else {
        const int i = b - p.n_po;
        gid = p.sp_subj[i]; r.rel = p.sp_rel[i]; r.pos = (uint32_t)i; r.sp = true;
    }
    r.owned = gid >= p.ent_lo && gid < p.ent_hi;
    r.ent = gid - p.ent_lo;
    int *err = leader ? p.id_err : nullptr;
    if (!r.owned && (p.whole_table || gid < 0) && err) atomicAdd(err, 1);   // not "another rank's row": a bad id
    r.rel = checked_row(r.rel, p.n_rel, err);
    return r;
}

// ComplEx query fold, written with explicit roundings (no fma contraction) so that every kernel that folds the same
// masked rows produces the same bits:  sp [s1 r1 - s2 r2 , s2 r1 + s1 r2]   po [o1 r1 + o2 r2 , o2 r1 - o1 r2]
__device__ __forceinline__ void fold_complex(bool sp, float e1, float e2, float r1, float r2, float &q1, float &q2)
{
    const float a = __fmul_rn(e1, r1), b = __fmul_rn(e2, r2), c = __fmul_rn(e2, r1), dd = __fmul_rn(e1, r2);
    q1 = sp ? __fsub_rn(a, b) : __fadd_rn(a, b);
    q2 = sp ? __fadd_rn(c, dd) : __fsub_rn(c, dd);
}

__device__ __forceinline__ RowSrc row_source(const PrefixDev &p, int b) { return row_source(p, b, threadIdx.x == 0); }

// score(b, n) exactly as fused_tile_kernel<KB, MODE_SCORE/MODE_COUNT> computes it: v_mfma_f32_16x16x4_f32 is a
// k-ordered fp32 fma chain (MI355X guide), and the tile kernel feeds it k = 16r + 4s + j in the order r, j, s -- so a
// scalar fmaf chain in that order gives the same bits.  Columns >= d hold zeros on both sides: fma(0, 0, acc) == acc.
// The row's loads are issued eight 16-column blocks at a time (32 float4 in flight) ahead of their fma chain: one
// dependent round trip per block of eight instead of one per block.
__device__ __forceinline__ float point_score(const float *__restrict__ q /* LDS, zero padded to 16*KB */,
                                             const float *__restrict__ row, int d, int KB, bool vec_ok)
{
    // slot sizes above 256 (KB = 32, fused_tile64k_kernel): the tile's contraction is split over two waves -- columns 0 .. 255 and
    // 256 .. 511, each a k-ordered chain of its own -- and the two partial scores are added once (a + b == b + a bit for bit)
    float acc = 0.f, acc_lo = 0.f;
    constexpr int RB = 8;
    for (int r0 = 0; r0 < KB; r0 += RB) {
        if (KB == 32 && r0 == 16) { acc_lo = acc; acc = 0.f; }
        float4 cv[RB][4];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int r = r0 + rr;
            if (r < KB && vec_ok && 16 * r + 16 <= d) {
#pragma unroll
                for (int m = 0; m < 4; ++m) cv[rr][m] = *reinterpret_cast<const float4 *>(row + 16 * r + 4 * m);
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const int k = 16 * r + 4 * m;
                    cv[rr][m] = make_float4(r < KB && k < d ? row[k] : 0.f, r < KB && k + 1 < d ? row[k + 1] : 0.f,
                                            r < KB && k + 2 < d ? row[k + 2] : 0.f, r < KB && k + 3 < d ? row[k + 3] : 0.f);
                }
            }
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int r = r0 + rr;
            if (r >= KB) break;
            const float c16[16] = {cv[rr][0].x, cv[rr][0].y, cv[rr][0].z, cv[rr][0].w, cv[rr][1].x, cv[rr][1].y, cv[rr][1].z, cv[rr][1].w,
                                   cv[rr][2].x, cv[rr][2].y, cv[rr][2].z, cv[rr][2].w, cv[rr][3].x, cv[rr][3].y, cv[rr][3].z, cv[rr][3].w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) acc = __fmaf_rn(q[16 * r + 4 * sl + j], c16[4 * sl + j], acc);
        }
    }
    return KB == 32 ? __fadd_rn(acc_lo, acc) : acc;
}

// One workgroup per batch row: the folded query row (-> Q for the tile sweep, and LDS), then the POINT scores the rank
// rule needs besides the sweep's counts: each answer group's true score = max over its ids (dataset.py:436), and the
// score under every filter entry (they are replaced by -1e8, dataset.py:441: corrected in eval_ranks_block).
//
// The sweep sees the batch SORTED BY GROUP COUNT (descending, stable): its counting loop runs as long as the busiest of a
// wave's 16 rows, so rows with many answers are put together (measured: 2.4x fewer loop iterations at a mean of 1.8
// groups per row).  Every workgroup finds its own row's position by counting the rows in front of it -- O(B) loads,
// no sort kernel, no extra launch: pos = #{ng' > ng} + #{b' < b, ng' == ng}; the sorted CSR offset is the same sum over
// ng'.  Q row, true scores and the sorted row_ptr go to the sorted positions; gshift[b] maps a group index back and
// group_row[g] names the row of a group (eval_ranks_block would otherwise search row_ptr: ten dependent loads).
//
// The kernel is a chain of dependent loads (ids -> rows, row_ptr -> grp_ptr -> ids -> rows), so it is written in stages
// that issue everything whose address is known before waiting: (A) the row's CSR bounds and prefix ids, (B) the fold
// operands, the group bounds / filter column of the thread's first item and the row-order counts, (C) the first answer
// id, (D) the candidate rows.  Eval mode (no dropout: the entry point refuses it); 256 threads fold up to 512 columns.
__device__ __forceinline__ void eval_points_block(const EvalPointsArgs &a, int b)
{
    __shared__ float qs[512];
    __shared__ int red_pos[4];
    __shared__ long long red_start[4];
    const PrefixDev &p = a.p;
    const int B = p.n_po + p.n_sp, d = a.d, KB = a.KB, ldq = a.ldq, tid = threadIdx.x, nthr = blockDim.x;
    const bool in = b < B;
    // ---- stage A
    RowSrc rs;
    rs.owned = false; rs.sp = false; rs.ent = rs.rel = 0; rs.pos = 0;
    int64_t g_lo = 0, g_hi = 0, f_lo = 0, f_hi = 0;
    if (in) {
        if (!a.Q_in) rs = row_source(p, b);
        g_lo = a.row_ptr[b]; g_hi = a.row_ptr[b + 1];
        f_lo = a.filt_ptr[b]; f_hi = a.filt_ptr[b + 1];
    }
    // ---- stage B
    const int h = d >> 1;
    const bool cplx = a.scorer != SC_DISTMULT;
    const int n_fold = cplx ? h : d;                        // folded columns: one per thread, two above 256 (DistMult d <= 512)
    float ea[2] = {0.f, 0.f}, eb[2] = {0.f, 0.f}, ra[2] = {0.f, 0.f}, rb[2] = {0.f, 0.f};
    if (rs.owned) {
        const float *e = a.E + rs.ent * d, *r = a.R + rs.rel * d;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * nthr;
            if (k < n_fold) {
                ea[it] = e[k]; ra[it] = r[k];
                if (cplx) { eb[it] = e[h + k]; rb[it] = r[h + k]; }
            }
        }
    }
    const int64_t g0 = g_lo + tid, f0 = f_lo + (nthr - 1 - tid);   // filter entries take the threads from the top: a row's
    const bool has_g = g0 < g_hi, has_f = f0 < f_hi;                // groups and filter entries run side by side
    int64_t j_lo = 0, j_hi = 0;
    if (has_g) { j_lo = a.grp_ptr[g0]; j_hi = a.grp_ptr[g0 + 1]; }
    const int fcol0 = has_f ? a.filt_col[f0] : 0;
    // position of this row in the order sorted by group count, and the first sorted group index of the row
    int pos = b;
    int64_t start = 0;
    const int64_t ng = g_hi - g_lo;
    int cnt = 0;
    long long sum = 0;
    if (in)
        for (int o = tid; o < B; o += nthr) {
            const int64_t ngo = a.row_ptr[o + 1] - a.row_ptr[o];
            const bool before = ngo > ng || (ngo == ng && o < b);
            cnt += before;
            sum += before ? ngo : 0;
        }
    // ---- stage C
    const int id0 = has_g && j_lo < j_hi ? a.ids[j_lo] : 0;
    cnt = wave_sum(cnt);
    sum = (long long)wave_sum((double)sum);                  // exact: group counts are far below 2^53
    if ((tid & 63) == 0) { red_pos[tid >> 6] = cnt; red_start[tid >> 6] = sum; }
    // the folded query row (model.py:205-216, :269-272; same roundings as encode_query_row with all masks = 1)
    for (int k = tid; k < 512; k += nthr) qs[k] = 0.f;
    __syncthreads();
    if (rs.owned) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int k = tid + it * nthr;
            if (k < n_fold) {
                if (cplx) fold_complex(rs.sp, ea[it], eb[it], ra[it], rb[it], qs[k], qs[h + k]);
                else qs[k] = __fmul_rn(ea[it], ra[it]);
            }
        }
    }
    if (a.Q_in && in)                    // sharded: the row was folded from the exchanged entity rows (okge_fold_queries)
        for (int k = tid; k < 16 * KB; k += nthr) qs[k] = a.Q_in[(size_t)b * ldq + k];
    if (in) {
        pos = red_pos[0] + red_pos[1] + red_pos[2] + red_pos[3];
        start = red_start[0] + red_start[1] + red_start[2] + red_start[3];
        if (tid == 0) {
            a.row_ptr_sorted[pos] = start;
            if (pos == B - 1) a.row_ptr_sorted[B] = start + ng;
            a.gshift[b] = start - g_lo;
        }
    }
    __syncthreads();
    for (int k = tid; k < ldq; k += nthr) a.Q[(size_t)pos * ldq + k] = k < 16 * KB ? qs[k] : 0.f;
    if (!in) return;
    // ---- stage D
    const bool vec_ok = (d & 3) == 0;
    // col: a position in the (global) candidate list (checked); candidates of another shard have no row here (nullptr:
    // their point scores are that shard's business); then an entity row (checked)
    auto cand_row = [&](int col) -> const float * {
        col = (int)checked_row(col, a.n_cand_global, p.id_err);
        const int loc = col - a.col_lo;
        if (loc < 0 || loc >= a.n_cand) return nullptr;
        const int64_t cid = checked_row(a.cand_ids ? (int64_t)a.cand_ids[loc] : (int64_t)a.cand_first + loc, a.table_rows, p.id_err);
        return a.E + cid * d;
    };
    for (int64_t g = g0; g < g_hi; g += nthr) {
        float t = -INFINITY;                                 // (sharded: the maximum over THIS shard's ids; -inf if it has none)
        const int64_t lo = g == g0 ? j_lo : a.grp_ptr[g], hi = g == g0 ? j_hi : a.grp_ptr[g + 1];
        for (int64_t j = lo; j < hi; ++j) {
            const float *row = cand_row(g == g0 && j == lo ? id0 : a.ids[j]);
            if (row) t = fmaxf(t, point_score(qs, row, d, KB, vec_ok));
        }
        a.true_out[g + (start - g_lo)] = t;                  // sorted group index
        a.group_row[g] = b;
    }
    for (int64_t f = f0; f < f_hi; f += nthr) {
        const float *row = cand_row(f == f0 ? fcol0 : a.filt_col[f]);
        a.filt_x[f] = row ? point_score(qs, row, d, KB, vec_ok) : __builtin_nanf("");      // NaN: not this shard's column
    }
}

// One wave per answer group: rank = #greater + #equal / 2 from the sweep's counts, after replacing the scores under the
// row's filter entries by -1e8 (dataset.py:441-446); then the meters of compute_metrics (dataset.py:447-452) are added
// to acc[7] (double atomics: one set per workgroup -- a set per wave, 934 x 7 atomics on seven words, took 35 us).
__device__ __forceinline__ void eval_ranks_block(const EvalRanksArgs &a, int blk)
{
    __shared__ double red[4][7];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t g = (int64_t)blk * 4 + w;
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    if (g < a.n_groups) {
        const int row = a.group_row[g];
        const int64_t gs = g + a.gshift[row];            // the group's index in the sweep's (sorted) numbering
        const int64_t f_lo = a.filt_ptr[row], f_hi = a.filt_ptr[row + 1];
        const float t = a.true_scores[gs];
        int gt = 0, eq = 0;
        for (int64_t f = f_lo + lane; f < f_hi; f += 64) {
            const float x = a.filt_x[f];
            if (x == x) {                                // (NaN: a filter column of another shard, corrected there)
                gt += (-1e8f > t) - (x > t);
                eq += (-1e8f == t) - (x == t);
            }
        }
        if (a.slab)                                      // the sweep's per-tile packed counts of this group
            for (int tl = lane; tl < a.tiles; tl += 64) {
                const uint32_t pk = a.slab[(size_t)tl * a.n_groups + gs];
                gt += (int)(pk & 0xFFFFu);
                eq += (int)(pk >> 16);
            }
        gt = wave_sum(gt);
        eq = wave_sum(eq);
        if (lane == 0 && a.counts_out) {                 // sharded: this shard's counts; the caller adds the shards up
            a.counts_out[2 * g] = (int64_t)(a.slab ? 0 : a.counts[2 * gs]) + gt;
            a.counts_out[2 * g + 1] = (int64_t)(a.slab ? 0 : a.counts[2 * gs + 1]) + eq;
        } else if (lane == 0) {
            const int64_t r = (int64_t)(a.slab ? 0 : a.counts[2 * gs]) + gt + ((int64_t)(a.slab ? 0 : a.counts[2 * gs + 1]) + eq) / 2;
            a.ranks[g] = r;
            v[0] = 1.0;
            v[1] = (double)(1.0f / (float)(r + 1));       // fp32 reciprocal like the reference's (1/(rank+1).float())
            v[2] = (double)r;
            v[3] = r < 1; v[4] = r < 3; v[5] = r < 10; v[6] = r < 50;
        }
    }
    if (lane == 0)
        for (int k = 0; k < 7; ++k) red[w][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 7) {
        const double tsum = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (tsum != 0.0) atomicAdd(a.acc + threadIdx.x, tsum);
    }
}

}  // namespace okge
