// Small HBM-bound kernels around the fused tile kernels (gfx950):
//   encode_queries_kernel   gather + dropout + fold (s,r)/(r,o) into one query row   model.py:455-510, :205-216, :269-272
//   prefix_backward_kernel  sum dQ slabs, chain rule to s/r/o rows, scatter-add       autograd of the above + embedding_dense_backward
//   loss_reduce_kernel      deterministic sum of per-workgroup loss partials           trainer.py:106 (reduction='sum')
//   kl_* kernels            log-sum-exp per row from tile partials, positives per row  trainer.py:99-100
//   adagrad_kernel          dense Adagrad sweep (+ zero_grad)                          utils/optim.py:139-160 / torch.optim.Adagrad
//   ranks_kernel            filtered ranks, exact integer counts                       dataset.py:423-446
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "okge_device.h"
#include "okge_kernels.h"
#include "okge_eval_device.h"

namespace okge {

// One folded query row (and / or its masked entity row): gather + dropout + fold.  All threads of the workgroup take part.
__device__ __forceinline__ void encode_query_row(const float *__restrict__ E, const float *__restrict__ R, int d, int scorer,
                                                 const PrefixDev &p, int b, float *__restrict__ q, float *__restrict__ er,
                                                 int ldq)
{
    const int B = p.n_po + p.n_sp;
    RowSrc rs;
    rs.owned = false;
    if (b < B) rs = row_source(p, b);
    if (!rs.owned) {       // padding row, or a prefix whose entity lives on another rank: contributes zero
        for (int k = threadIdx.x; k < ldq; k += blockDim.x) {
            if (q) q[k] = 0.f;
            if (er) er[k] = 0.f;
        }
        return;
    }
    const DropDev &de = rs.sp ? p.drop_sp_ent : p.drop_po_ent;
    const DropDev &dr = rs.sp ? p.drop_sp_rel : p.drop_po_rel;
    const float *e = E + rs.ent * d, *r = R + rs.rel * d;
    if (scorer == SC_DISTMULT) {
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            const float ev = e[k] * drop_mult1(de, rs.pos, k, d);
            if (q) q[k] = __fmul_rn(ev, __fmul_rn(r[k], drop_mult1(dr, rs.pos, k, d)));
            if (er) er[k] = ev;
        }
    } else {
        const int h = d >> 1;
        if ((h & 3) == 0 && (de.enabled || dr.enabled)) {
            // The four threads of an aligned column quad need the same four keep octets -- columns k and h + k of the
            // entity and of the relation mask (h % 4 == 0: a quad never straddles an octet) -- so each computes ONE of
            // them (a Philox call) and they trade by shuffles, instead of four Philox calls per thread.
            const int role = threadIdx.x & 3, lane0 = (threadIdx.x & 63) & ~3;
            for (int k = threadIdx.x; k < h; k += blockDim.x) {
                const float ve1 = e[k], ve2 = e[h + k], vr1 = r[k], vr2 = r[h + k];     // requested before the mask arithmetic
                const DropDev &dd = (role & 2) ? dr : de;
                const int kc = (role & 1) ? h + k : k;
                const uint32_t mine = dd.enabled ? drop_keep8(dd, rs.pos, kc >> 3, d) : 0xFFu;
                const uint32_t b_e1 = __shfl(mine, lane0), b_e2 = __shfl(mine, lane0 + 1);
                const uint32_t b_r1 = __shfl(mine, lane0 + 2), b_r2 = __shfl(mine, lane0 + 3);
                auto mult = [](const DropDev &x, uint32_t bits, int col) { return !x.enabled ? 1.f : (bits >> (col & 7) & 1u) ? x.scale : 0.f; };
                const float e1 = ve1 * mult(de, b_e1, k), e2 = ve2 * mult(de, b_e2, h + k);
                const float r1 = vr1 * mult(dr, b_r1, k), r2 = vr2 * mult(dr, b_r2, h + k);
                if (q) fold_complex(rs.sp, e1, e2, r1, r2, q[k], q[h + k]);
                if (er) { er[k] = e1; er[h + k] = e2; }
            }
        } else {
            for (int k = threadIdx.x; k < h; k += blockDim.x) {
                const float e1 = e[k] * drop_mult1(de, rs.pos, k, d), e2 = e[h + k] * drop_mult1(de, rs.pos, h + k, d);
                const float r1 = r[k] * drop_mult1(dr, rs.pos, k, d), r2 = r[h + k] * drop_mult1(dr, rs.pos, h + k, d);
                if (q) fold_complex(rs.sp, e1, e2, r1, r2, q[k], q[h + k]);
                if (er) { er[k] = e1; er[h + k] = e2; }
            }
        }
    }
    for (int k = d + threadIdx.x; k < ldq; k += blockDim.x) {
        if (q) q[k] = 0.f;
        if (er) er[k] = 0.f;
    }
}

__global__ __launch_bounds__(128) void encode_queries_kernel(const float *__restrict__ E, const float *__restrict__ R,
                                                             int d, int scorer, const PrefixDev p,
                                                             float *__restrict__ Q, int ldq, int Bpad,
                                                             float *__restrict__ ent_rows,
                                                             const int32_t *__restrict__ pos_col, int nnz,
                                                             int32_t *__restrict__ tile_ptr, int tiles, int tile_w,
                                                             int cand_col0, int tp_wgs, const ClearSpec clr)
{
    if ((int)blockIdx.x >= Bpad + tp_wgs) {
        // more extra workgroups (OKGE_TRAIN_CLEAR_GRADS): the gradient regions the step accumulates into without storing
        // them first -- all of dR, the rows of dE outside the candidate range -- are cleared here, one float4 per thread, in
        // the launch that precedes every kernel that adds to them (instead of two fill launches by the caller)
        int64_t i = 4 * (((int64_t)blockIdx.x - Bpad - tp_wgs) * 128 + threadIdx.x);
#pragma unroll
        for (int r = 0; r < CLEAR_REGIONS; ++r) {
            float *base = clr.p[r];
            const int64_t n = clr.n[r], n4 = (n + 3) / 4 * 4;
            if (i < n4) {
                if (i + 4 <= n && (reinterpret_cast<uintptr_t>(base + i) & 15) == 0) *reinterpret_cast<float4 *>(base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
                else for (int64_t j = i; j < n && j < i + 4; ++j) base[j] = 0.f;
                return;
            }
            i -= n4;
        }
        return;
    }
    if ((int)blockIdx.x >= Bpad) {
        // extra workgroups: offsets of each candidate tile's positives in the column-sorted coordinate list
        const int t = ((int)blockIdx.x - Bpad) * 128 + threadIdx.x;
        if (t <= tiles) tile_ptr[t] = lower_bound_i32(pos_col, nnz, cand_col0 + t * tile_w);
        return;
    }
    const int b = blockIdx.x;
    if (clr.prefix_flags && threadIdx.x == 0 && b < p.n_po + p.n_sp) {              // okge_train_step: this row's entity gets prefix gradients
        const RowSrc rs = row_source(p, b, false);
        if (rs.owned) clr.prefix_flags[rs.ent] = 1;
    }
    encode_query_row(E, R, d, scorer, p, b, Q ? Q + (size_t)b * ldq : nullptr,      // Q == nullptr: only the masked
                     ent_rows ? ent_rows + (size_t)b * ldq : nullptr, ldq);         // entity rows are wanted
}

// Q[b] = fold(masked entity row b, dropout(R[rel_b])) from ALREADY MASKED entity rows (sharded path: the rows arrive
// through the all-reduce; the replicated relation table and the counter-based masks let every rank fold them itself,
// so only B x d floats cross the links instead of 2 x B x d).  Same arithmetic as encode_queries_kernel: bit-identical.
__global__ __launch_bounds__(128) void fold_queries_kernel(const float *__restrict__ R, int d, int scorer, const PrefixDev p,
                                                           const float *__restrict__ ent_rows, float *__restrict__ Q, int ldq)
{
    const int b = blockIdx.x, B = p.n_po + p.n_sp;
    float *q = Q + (size_t)b * ldq;
    if (b >= B) {
        for (int k = threadIdx.x; k < ldq; k += blockDim.x) q[k] = 0.f;
        return;
    }
    const RowSrc rs = row_source(p, b);
    const DropDev &dr = rs.sp ? p.drop_sp_rel : p.drop_po_rel;
    const float *e = ent_rows + (size_t)b * ldq, *r = R + rs.rel * d;
    if (scorer == SC_DISTMULT) {
        for (int k = threadIdx.x; k < d; k += blockDim.x) q[k] = __fmul_rn(e[k], __fmul_rn(r[k], drop_mult1(dr, rs.pos, k, d)));
    } else {
        const int h = d >> 1;
        for (int k = threadIdx.x; k < h; k += blockDim.x) {
            const float e1 = e[k], e2 = e[h + k];
            const float r1 = r[k] * drop_mult1(dr, rs.pos, k, d), r2 = r[h + k] * drop_mult1(dr, rs.pos, h + k, d);
            fold_complex(rs.sp, e1, e2, r1, r2, q[k], q[h + k]);
        }
    }
    for (int k = d + threadIdx.x; k < ldq; k += blockDim.x) q[k] = 0.f;
}

__device__ __forceinline__ void loss_reduce_block(const double *__restrict__ partials, int n, double *__restrict__ out)
{
    __shared__ double red[4];
    const int nw = blockDim.x >> 6;
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += partials[i];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < nw; ++i) t += red[i];
        out[0] = t;
    }
}

// dQ[b][k] = sum over candidate ranges of the dq_kernel slabs (sharded path: reduced before the all-reduce)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float *__restrict__ slab, int nsplit, int64_t n4,
                                                          float *__restrict__ out, const double *__restrict__ loss_partials,
                                                          int n_partials, double *__restrict__ loss_out)
{
    if (blockIdx.x == gridDim.x - 1) {                      // one extra workgroup: the deterministic loss reduction
        if (loss_partials) loss_reduce_block(loss_partials, n_partials, loss_out);
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sidx = 0; sidx < nsplit; ++sidx) {
        const float4 v = reinterpret_cast<const float4 *>(slab)[(size_t)sidx * n4 + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    reinterpret_cast<float4 *>(out)[i] = acc;
}

// dE[candidate n] (+)= sum over the batch splits of the fused tile kernel's partial candidate-gradient slabs
// (one thread per 4 columns; rows of repeated candidate ids accumulate with atomics)
__global__ __launch_bounds__(256) void dc_reduce_kernel(const float *__restrict__ slab, int nsplit, int rows_pad, int D16,
                                                        int N, int d, const int32_t *__restrict__ cand_ids, int cand_first,
                                                        int exclusive, int grads_zero, float *__restrict__ dE,
                                                        int64_t table_rows, int *__restrict__ id_err)
{
    const int q4 = D16 >> 2;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)N * q4) return;
    const int n = (int)(i / q4), k = 4 * (int)(i % q4);
    if (k >= d) return;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sidx = 0; sidx < nsplit; ++sidx) {
        const float4 v = *reinterpret_cast<const float4 *>(slab + ((size_t)sidx * rows_pad + n) * D16 + k);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int64_t cid = checked_row(cand_ids ? (int64_t)cand_ids[n] : (int64_t)cand_first + n, table_rows, id_err);
    float *dst = dE + cid * d + k;
    const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (k + e >= d) break;
        if (!exclusive) atomicAdd(dst + e, v[e]);
        else dst[e] = grads_zero ? v[e] : dst[e] + v[e];
    }
}

// Stream-K launches of fused_tile64k_kernel (okge_train64k.hip): the (tile, chunk) units, tile-major, are cut into `P` equal
// runs; workgroup p owns units [U p / P, U (p + 1) / P), U = tiles * J.  A tile covered by ONE segment was written by that
// workgroup; otherwise every segment left its partial rows in slab 2 p (p's run starts inside the tile) or 2 p + 1 (it
// ends there).  One workgroup per tile: list the slabs (in p order: the sum is reproducible), add them, store / accumulate.
constexpr int SK_MAX_WGS = 511, SK_SLOTS = 1024;     // a tile's slab list holds at most two entries per workgroup of the launch
static_assert(SK_SLOTS >= 2 * (SK_MAX_WGS + 1), "dc_reduce_streamk_kernel: slot list too small for the largest stream-K launch");
__global__ __launch_bounds__(256) void dc_reduce_streamk_kernel(const float *__restrict__ slab, int tiles, int J, int P, int D16,
                                                                int N, int d, const int32_t *__restrict__ cand_ids,
                                                                int cand_first, int exclusive, int grads_zero,
                                                                float *__restrict__ dE, int64_t table_rows,
                                                                int *__restrict__ id_err)
{
    // grid: 8 workgroups per tile (8 candidate rows each: the launch is a 3 x 20 MB stream, it needs the whole chip)
    __shared__ int slots[SK_SLOTS];
    __shared__ int n_slots;
    const int t = blockIdx.x >> 3, r0 = 8 * (blockIdx.x & 7);
    const int64_t U = (int64_t)tiles * J, lo = (int64_t)t * J, hi = lo + J;
    if (threadIdx.x == 0) {
        int64_t p = lo * P / U;
        while (p > 0 && U * p / P > lo) --p;
        while (U * (p + 1) / P <= lo) ++p;
        int cnt = 0;
        bool whole = false;
        for (; p < P && U * p / P < hi; ++p) {
            const int64_t ub = U * p / P, ue = U * (p + 1) / P;
            const int64_t j0 = (ub > lo ? ub : lo) - lo, j1 = (ue < hi ? ue : hi) - lo;
            if (j1 <= j0) continue;
            if (j0 == 0 && j1 == J) { whole = true; break; }
            if (cnt < SK_SLOTS) slots[cnt++] = (int)(2 * p + (ub >= lo ? 0 : 1));
            else if (id_err) atomicAdd(id_err, 1);            // (unreachable: the launcher refuses P > SK_MAX_WGS; never a silent drop)
        }
        n_slots = whole ? 0 : cnt;
    }
    __syncthreads();
    const int ns = n_slots;
    if (ns == 0) return;
    const int q4 = D16 >> 2;
    for (int i = threadIdx.x; i < 8 * q4; i += 256) {
        const int r = r0 + i / q4, k = 4 * (i % q4), n = 64 * t + r;
        if (n >= N || k >= d) continue;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = 0; j < ns; ++j) {
            const float4 v = *reinterpret_cast<const float4 *>(slab + ((size_t)slots[j] * 64 + r) * D16 + k);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const int64_t cid = checked_row(cand_ids ? (int64_t)cand_ids[n] : (int64_t)cand_first + n, table_rows, k ? nullptr : id_err);
        float *dst = dE + cid * d + k;
        const float v[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (k + e >= d) break;
            if (!exclusive) atomicAdd(dst + e, v[e]);
            else dst[e] = grads_zero ? v[e] : dst[e] + v[e];
        }
    }
}

__global__ __launch_bounds__(128) void prefix_backward_kernel(const float *__restrict__ E, const float *__restrict__ R,
                                                              int d, int scorer, const PrefixDev p,
                                                              const float *__restrict__ slab, int nsplit, int Bpad,
                                                              int ldq, const float *__restrict__ ent_rows,
                                                              float *__restrict__ dE, float *__restrict__ dR, int distinct)
{
    // ent_rows (sharded path): the already masked prefix entity rows of ALL prefixes, so every rank forms the full
    // relation gradient; the entity gradient is scattered by the owner only.
    // distinct (OKGE_TRAIN_DISTINCT_PREFIX_ROWS): every prefix id occurs once -- its gradient row is stored, not accumulated
    auto put = [&](float *dst, float v) {
        if (distinct) *dst = v;
        else atomicAdd(dst, v);
    };
    const int b = blockIdx.x;
    const RowSrc rs = row_source(p, b);
    if (!ent_rows && !rs.owned) return;
    const DropDev &de = rs.sp ? p.drop_sp_ent : p.drop_po_ent;
    const DropDev &dr = rs.sp ? p.drop_sp_rel : p.drop_po_rel;
    const float *e = ent_rows ? ent_rows + (size_t)b * ldq : E + rs.ent * d, *r = R + rs.rel * d;
    const bool e_masked = ent_rows != nullptr;
    float *ge = dE + (rs.owned ? rs.ent : 0) * d, *gr = dR + rs.rel * d;
    const size_t split_stride = (size_t)Bpad * ldq;
    const float *sl = slab + (size_t)b * ldq;
    if (scorer == SC_DISTMULT) {
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            float dq = 0.f;
            for (int sidx = 0; sidx < nsplit; ++sidx) dq += sl[sidx * split_stride + k];
            const float me = drop_mult1(de, rs.pos, k, d), mr = drop_mult1(dr, rs.pos, k, d);
            const float ev = e_masked ? e[k] : e[k] * me, rv = r[k] * mr;
            if (rs.owned) put(ge + k, dq * rv * me);
            put(gr + k, dq * ev * mr);
        }
        return;
    }
    const int h = d >> 1;
    for (int k = threadIdx.x; k < h; k += blockDim.x) {
        float q1 = 0.f, q2 = 0.f;
        for (int sidx = 0; sidx < nsplit; ++sidx) {
            q1 += sl[sidx * split_stride + k];
            q2 += sl[sidx * split_stride + h + k];
        }
        const float me1 = drop_mult1(de, rs.pos, k, d), me2 = drop_mult1(de, rs.pos, h + k, d);
        const float mr1 = drop_mult1(dr, rs.pos, k, d), mr2 = drop_mult1(dr, rs.pos, h + k, d);
        const float e1 = e_masked ? e[k] : e[k] * me1, e2 = e_masked ? e[h + k] : e[h + k] * me2;
        const float r1 = r[k] * mr1, r2 = r[h + k] * mr2;
        float de1, de2, dr1, dr2;
        if (rs.sp) {
            de1 = q1 * r1 + q2 * r2;  de2 = -q1 * r2 + q2 * r1;
            dr1 = q1 * e1 + q2 * e2;  dr2 = -q1 * e2 + q2 * e1;
        } else {
            de1 = q1 * r1 - q2 * r2;  de2 = q1 * r2 + q2 * r1;
            dr1 = q1 * e1 + q2 * e2;  dr2 = q1 * e2 - q2 * e1;
        }
        if (rs.owned) {
            put(ge + k, de1 * me1);
            put(ge + h + k, de2 * me2);
        }
        put(gr + k, dr1 * mr1);
        put(gr + h + k, dr2 * mr2);
    }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const double *__restrict__ partials, int n,
                                                          double *__restrict__ out)
{
    loss_reduce_block(partials, n, out);
}

__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c)
{
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 f4neg(float4 a) { return make_float4(-a.x, -a.y, -a.z, -a.w); }
__device__ __forceinline__ void atomic_add4(float *p, float4 v)
{
    atomicAdd(p, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

// One batch row per 128-thread workgroup: lane = (column group g = tid >> 2, split quarter sq = tid & 3).  The four
// lanes of a column group sum disjoint quarters of the nsplit dQ slabs (all their loads are issued at once) and
// combine with two shuffles; sq == 0 then applies the chain rule and scatters.  Needs d % 8 == 0 (ComplEx) or
// d % 4 == 0 (DistMult).  The last workgroup (blockIdx.x == gridDim.x - 1) instead sums the loss partials.
// NB: slab loads kept in flight per lane (8 for the single-device step's 32 split-K slabs; 1 for the sharded step, whose dQ
// arrives already reduced: 89 instead of 149 registers = 5 instead of 3 waves per SIMD, which is what bounds a launch of
// 4096 one-row workgroups -- 2.7 rounds of a ~7 us dependent-load chain at 3 waves)
// ---- okge_train_step: Adagrad inside the step's last launches (AdagradFuse, okge_kernels.h) -------------------------------
// (bits_differ, adagrad4: okge_device.h)

// entity rows without a prefix flag: their gradient row is final (the tile kernel stored it, no prefix of the batch names them)
__device__ __forceinline__ void fused_entity_sweep(const AdagradFuse &af, int wg, int n_wgs)
{
    const uint32_t row4 = (uint32_t)af.d >> 2;
    const int64_t n4 = af.n_ent * row4, stride = (int64_t)n_wgs * blockDim.x;
    float4 *p4 = reinterpret_cast<float4 *>(af.E), *g4 = reinterpret_cast<float4 *>(af.dE), *s4 = reinterpret_cast<float4 *>(af.sumE);
    for (int64_t i = (int64_t)wg * blockDim.x + threadIdx.x; i < n4; i += stride) {
        // (the flag is requested WITH the row data, not before it: one round trip; a flagged row's loads are wasted, there are <= B)
        const int32_t flagged = af.flags[(uint32_t)i / row4];
        float4 pv = p4[i], sv = s4[i];
        const float4 gv = g4[i], p_old = pv, s_old = sv;
        if (flagged != 0) continue;
        adagrad4(pv, gv, sv, af.lr, af.wd, af.eps);
        if (bits_differ(pv, p_old)) p4[i] = pv;
        if (bits_differ(sv, s_old)) s4[i] = sv;
        if (af.zero_dE && ((__float_as_uint(gv.x) | __float_as_uint(gv.y) | __float_as_uint(gv.z) | __float_as_uint(gv.w)) != 0u))
            g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// the rest: workgroup b < B claims its batch row's prefix entity row (the flag goes back to 0: several batch rows may name one
// entity, one of them updates it), the workgroups behind sweep the relation table (gradient cleared, as okge_adagrad_step2 does)
__global__ __launch_bounds__(128) void adagrad_finish_kernel(const AdagradFuse af, const PrefixDev p, int B)
{
    __shared__ int64_t claimed;
    const uint32_t row4 = (uint32_t)af.d >> 2;
    if ((int)blockIdx.x < B) {
        if (threadIdx.x == 0) {
            const RowSrc rs = row_source(p, blockIdx.x, false);
            claimed = (rs.owned && atomicExch(&af.flags[rs.ent], 0) != 0) ? rs.ent : -1;
        }
        __syncthreads();
        const int64_t row = claimed;
        if (row < 0) return;
        float4 *p4 = reinterpret_cast<float4 *>(af.E) + row * row4, *g4 = reinterpret_cast<float4 *>(af.dE) + row * row4;
        float4 *s4 = reinterpret_cast<float4 *>(af.sumE) + row * row4;
        for (uint32_t i = threadIdx.x; i < row4; i += blockDim.x) {
            float4 pv = p4[i], sv = s4[i];
            const float4 gv = g4[i];
            adagrad4(pv, gv, sv, af.lr, af.wd, af.eps);
            p4[i] = pv;
            s4[i] = sv;
            if (af.zero_dE) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const int64_t n4 = (int64_t)af.n_rel * row4, stride = (int64_t)(gridDim.x - B) * blockDim.x;
    float4 *p4 = reinterpret_cast<float4 *>(af.R), *g4 = reinterpret_cast<float4 *>(af.dR), *s4 = reinterpret_cast<float4 *>(af.sumR);
    for (int64_t i = (int64_t)(blockIdx.x - B) * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = p4[i], sv = s4[i];
        const float4 gv = g4[i], p_old = pv, s_old = sv;
        adagrad4(pv, gv, sv, af.lr, af.wd, af.eps);
        if (bits_differ(pv, p_old)) p4[i] = pv;
        if (bits_differ(sv, s_old)) s4[i] = sv;
        if ((__float_as_uint(gv.x) | __float_as_uint(gv.y) | __float_as_uint(gv.z) | __float_as_uint(gv.w)) != 0u) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int NB>
__global__ __launch_bounds__(128) void prefix_backward_vec_kernel(const float *__restrict__ E, const float *__restrict__ R,
                                                                  int d, int scorer, const PrefixDev p,
                                                                  const float *__restrict__ slab, int nsplit, int Bpad,
                                                                  int ldq, const float *__restrict__ ent_rows,
                                                                  float *__restrict__ dE, float *__restrict__ dR,
                                                                  const double *__restrict__ loss_partials,
                                                                  int n_partials, double *__restrict__ loss_out,
                                                                  float *__restrict__ dr_rows, float *__restrict__ de_rows,
                                                                  int distinct, const AdagradFuse af, int B_rows)
{
    // blockIdx.y: 128-column chunk of the row (rows longer than 128 floats per half: the chunks were a loop of dependent
    // round trips inside one workgroup -- 21 us at DistMult d = 512 -- and are workgroups of their own now)
    if ((int)blockIdx.x >= B_rows && blockIdx.y != 0) return;
    if ((int)blockIdx.x > B_rows) {                      // okge_train_step: workgroups behind the loss reduction sweep the entity table
        fused_entity_sweep(af, (int)blockIdx.x - B_rows - 1, (int)gridDim.x - B_rows - 1);
        return;
    }
    // dr_rows / de_rows ([B][ldq] each, or nullptr): the relation- / entity-gradient row of every batch row is STORED there
    // instead of being added into dR / dE with float atomics -- row_segment_sum_kernel then adds up the rows of each
    // relation / entity (okge_prefix_backward_segmented)
    if ((int)blockIdx.x == B_rows) {
        if (loss_partials) loss_reduce_block(loss_partials, n_partials, loss_out);
        return;
    }
    const int b = blockIdx.x, grp = threadIdx.x >> 2, sq = threadIdx.x & 3;
    const RowSrc rs = row_source(p, b);
    if (!ent_rows && !rs.owned) return;
    const DropDev &de = rs.sp ? p.drop_sp_ent : p.drop_po_ent;
    const DropDev &dr = rs.sp ? p.drop_sp_rel : p.drop_po_rel;
    const float *e = ent_rows ? ent_rows + (size_t)b * ldq : E + rs.ent * d, *r = R + rs.rel * d;
    const bool e_masked = ent_rows != nullptr;
    float *ge = de_rows ? de_rows + (size_t)b * ldq : dE + (rs.owned ? rs.ent : 0) * d;
    float *gr = dr_rows ? dr_rows + (size_t)b * ldq : dR + rs.rel * d;
    const size_t split_stride = (size_t)Bpad * ldq;
    const float *sl = slab + (size_t)b * ldq;
    const int s_lo = (nsplit * sq) >> 2, s_hi = (nsplit * (sq + 1)) >> 2;
    // distinct (OKGE_TRAIN_DISTINCT_PREFIX_ROWS): every prefix id occurs once in the batch: the table's gradient row itself is
    // a row of its own
    auto put_r = [&](float *pr, float4 v) {      // relation gradient: a row of its own (plain store) or an atomic add into dR
        if (dr_rows || distinct) *reinterpret_cast<float4 *>(pr) = v;
        else atomic_add4(pr, v);
    };
    auto put_e = [&](float *pe, float4 v) {
        if (de_rows || distinct) *reinterpret_cast<float4 *>(pe) = v;
        else atomic_add4(pe, v);
    };
    auto dq_sum = [&](int k, bool active) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (active) {
            int sidx = s_lo;
            if (NB > 1)
                for (; sidx + NB <= s_hi; sidx += NB) {
                    float4 v[NB];
#pragma unroll
                    for (int u = 0; u < NB; ++u) v[u] = *reinterpret_cast<const float4 *>(sl + (sidx + u) * split_stride + k);
#pragma unroll
                    for (int u = 0; u < NB; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
            for (; sidx < s_hi; ++sidx) {
                const float4 v = *reinterpret_cast<const float4 *>(sl + sidx * split_stride + k);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        // the four lanes of a column group are a DPP quad: xor 1, xor 2 as quad permutes (a ds_bpermute shuffle each
        // would cost ~60 cycles on this latency-bound path)
        acc.x += dpp_mov<0xB1>(acc.x); acc.y += dpp_mov<0xB1>(acc.y); acc.z += dpp_mov<0xB1>(acc.z); acc.w += dpp_mov<0xB1>(acc.w);
        acc.x += dpp_mov<0x4E>(acc.x); acc.y += dpp_mov<0x4E>(acc.y); acc.z += dpp_mov<0x4E>(acc.z); acc.w += dpp_mov<0x4E>(acc.w);
        return acc;
    };
    // The keep nibbles of a column group: lane sq computes ONE of the (up to four) Philox calls the group needs and the
    // chain-rule lane collects them by shuffles; the table rows are requested before the slab sums so that their round
    // trip overlaps the slabs' (ids -> {rows, slabs, masks} -> atomics instead of ids -> slabs -> masks -> rows -> atomics).
    const int lane0 = (threadIdx.x & 63) & ~3;
    auto nibble = [&](const DropDev &dd, int k, bool active) -> uint32_t {
        if (!dd.enabled) return 15u;
        return active ? (drop_keep8(dd, rs.pos, k >> 3, d) >> (k & 4)) & 15u : 0u;
    };
    auto mult4 = [&](const DropDev &dd, uint32_t nib) {
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (dd.enabled) apply_keep4(m, nib, dd.scale);
        return m;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scorer == SC_DISTMULT) {
        for (int k0 = 128 * blockIdx.y; k0 < d; k0 += 128 * gridDim.y) {
            const int k = k0 + 4 * grp;
            const bool act = k < d, lead = act && sq == 0;
            const float4 ev0 = lead ? *reinterpret_cast<const float4 *>(e + k) : zero4;
            const float4 rv0 = lead ? *reinterpret_cast<const float4 *>(r + k) : zero4;
            const uint32_t nib = nibble((sq & 1) ? dr : de, k, act && sq < 2);
            const uint32_t nib_e = __shfl(nib, lane0), nib_r = __shfl(nib, lane0 + 1);
            const float4 dq = dq_sum(act ? k : 0, act);
            if (lead) {
                const float4 me = mult4(de, nib_e), mr = mult4(dr, nib_r);
                const float4 ev = e_masked ? ev0 : f4mul(ev0, me);
                const float4 rv = f4mul(rv0, mr);
                if (rs.owned) put_e(ge + k, f4mul(f4mul(dq, rv), me));
                put_r(gr + k, f4mul(f4mul(dq, ev), mr));
            }
        }
        return;
    }
    const int h = d >> 1;
    for (int k0 = 128 * blockIdx.y; k0 < h; k0 += 128 * gridDim.y) {
        const int k = k0 + 4 * grp;
        const bool act = k < h, lead = act && sq == 0;
        float4 e1 = zero4, e2 = zero4, r1 = zero4, r2 = zero4;
        if (lead) {
            e1 = *reinterpret_cast<const float4 *>(e + k);  e2 = *reinterpret_cast<const float4 *>(e + h + k);
            r1 = *reinterpret_cast<const float4 *>(r + k);  r2 = *reinterpret_cast<const float4 *>(r + h + k);
        }
        const uint32_t nib = nibble((sq & 2) ? dr : de, (sq & 1) ? h + k : k, act);
        const uint32_t n_e1 = __shfl(nib, lane0), n_e2 = __shfl(nib, lane0 + 1);
        const uint32_t n_r1 = __shfl(nib, lane0 + 2), n_r2 = __shfl(nib, lane0 + 3);
        const float4 q1 = dq_sum(act ? k : 0, act), q2 = dq_sum(act ? h + k : 0, act);
        if (!lead) continue;
        const float4 me1 = mult4(de, n_e1), me2 = mult4(de, n_e2), mr1 = mult4(dr, n_r1), mr2 = mult4(dr, n_r2);
        if (!e_masked) { e1 = f4mul(e1, me1); e2 = f4mul(e2, me2); }
        r1 = f4mul(r1, mr1);
        r2 = f4mul(r2, mr2);
        float4 de1, de2, dr1, dr2;
        if (rs.sp) {
            de1 = f4fma(q1, r1, f4mul(q2, r2));           de2 = f4fma(q2, r1, f4neg(f4mul(q1, r2)));
            dr1 = f4fma(q1, e1, f4mul(q2, e2));           dr2 = f4fma(q2, e1, f4neg(f4mul(q1, e2)));
        } else {
            de1 = f4fma(q1, r1, f4neg(f4mul(q2, r2)));    de2 = f4fma(q1, r2, f4mul(q2, r1));
            dr1 = f4fma(q1, e1, f4mul(q2, e2));           dr2 = f4fma(q1, e2, f4neg(f4mul(q2, e1)));
        }
        if (rs.owned) {
            put_e(ge + k, f4mul(de1, me1));
            put_e(ge + h + k, f4mul(de2, me2));
        }
        put_r(gr + k, f4mul(dr1, mr1));
        put_r(gr + h + k, f4mul(dr2, mr2));
    }
}

// dR[relation] / dE[entity] += the gradient rows of the batch rows that name this relation / entity, in the plan's (stable)
// order: one workgroup per segment, one float4 column group per thread -- plain loads and ONE read-modify-write per table
// row, no float atomics (1.6 M of them at the 8-rank FB15k-237 shape: 4096 batch rows x 200 columns x 2 tables, 17 batch
// rows per relation) and a fixed summation order.  order[] = batch rows sorted by id, seg_ptr[] = the bounds of the runs of
// equal ids (host-built, like the exchange plan: the ids are on the host before the batch is uploaded).  Workgroups
// 0 .. n_rel_seg-1 take the relation segments, the rest the entity segments (skipped when another rank owns the entity).
struct RowSegments { const int32_t *order, *seg_ptr; int32_t n_seg; };
__global__ __launch_bounds__(128) void row_segment_sum_kernel(const float *__restrict__ dr_rows, const float *__restrict__ de_rows,
                                                              int ldq, int d, const PrefixDev p, const RowSegments rel,
                                                              const RowSegments ent, float *__restrict__ dR, float *__restrict__ dE)
{
    const bool is_rel = (int)blockIdx.x < rel.n_seg;
    const RowSegments &sg = is_rel ? rel : ent;
    const int sidx = is_rel ? blockIdx.x : blockIdx.x - rel.n_seg;
    const int B = p.n_po + p.n_sp;
    // (the plan is host-built and trusted for speed, never for safety: bounds outside [0, B] would read `order` out of range)
    const int lo = min(max(sg.seg_ptr[sidx], 0), B), hi = min(max(sg.seg_ptr[sidx + 1], 0), B);
    if (hi <= lo) return;
    const RowSrc rs = row_source(p, min(max(sg.order[lo], 0), B - 1), false);
    if (!is_rel && !rs.owned) return;
    const float *rows = is_rel ? dr_rows : de_rows;
    float *dst = is_rel ? dR + rs.rel * d : dE + rs.ent * d;
    for (int k = 4 * threadIdx.x; k < d; k += 4 * 128) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = lo; i < hi; ++i) {
            const int b = min(max(sg.order[i], 0), B - 1);
            const float4 v = *reinterpret_cast<const float4 *>(rows + (size_t)b * ldq + k);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (k + 4 <= d) {
            float4 o = *reinterpret_cast<const float4 *>(dst + k);
            o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
            *reinterpret_cast<float4 *>(dst + k) = o;
        } else {
            const float v[4] = {acc.x, acc.y, acc.z, acc.w};
            for (int e = 0; k + e < d; ++e) dst[k + e] += v[e];
        }
    }
}

__global__ __launch_bounds__(256) void kl_count_pos_kernel(const int32_t *__restrict__ pos_row, int nnz,
                                                           float *__restrict__ row_ysum)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nnz && pos_row[i] >= 0) atomicAdd(row_ysum + pos_row[i], 1.0f);      // row < 0: padding (fixed-size graphs)
}

// row log-sum-exp from the per-(tile, row) (max, sum-exp) pairs of the stats pass: one wave per batch row, lanes stride
// over the tiles (a thread per row walked 2 x tiles strided loads in sequence: 98 us at 228 tiles)
__global__ __launch_bounds__(256) void kl_row_lse_kernel(const float *__restrict__ stats, int tiles, int B, int Bpad,
                                                         float2 *__restrict__ run, int first, int last,
                                                         float *__restrict__ row_lse, int row_blocks,
                                                         const int32_t *__restrict__ pos_row, int nnz,
                                                         float *__restrict__ row_ysum)
{
    if ((int)blockIdx.x >= row_blocks) {
        // extra workgroups: label mass per row (trainer.py:99-101: y is not normalised) -- row_ysum was cleared by the step's
        // first launch; this replaces a memset + a launch of its own
        const int i = ((int)blockIdx.x - row_blocks) * 256 + threadIdx.x;
        if (i < nnz && pos_row[i] >= 0) atomicAdd(row_ysum + pos_row[i], 1.0f);      // row < 0: padding (fixed-size graphs)
        return;
    }
    // candidate ranges: `run` carries the row's (max, sum of exp(x - max)) over the ranges seen so far
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float2 *st = reinterpret_cast<const float2 *>(stats);
    float M = -INFINITY, S = 0.f;                    // running (max, sum of exp(x - max)) of this lane's tiles
    if (!first && lane == 0) { const float2 v = run[b]; M = v.x; S = v.y; }
    for (int t = lane; t < tiles; t += 64) {
        const float2 v = st[(size_t)t * Bpad + b];
        const float m2 = fmaxf(M, v.x);
        if (m2 > -INFINITY) S = S * expf(M - m2) + v.y * expf(v.x - m2);
        M = m2;
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float Mo = __shfl_xor(M, o), So = __shfl_xor(S, o);
        const float m2 = fmaxf(M, Mo);
        if (m2 > -INFINITY) S = S * expf(M - m2) + So * expf(Mo - m2);
        M = m2;
    }
    if (lane == 0) {
        if (last) row_lse[b] = M + logf(S);
        else run[b] = make_float2(M, S);
    }
}

__global__ __launch_bounds__(256) void adagrad_kernel(float *__restrict__ p, float *__restrict__ g,
                                                      float *__restrict__ sum, int64_t n, float lr, float wd,
                                                      float eps, int zero_grad)
{
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g), *s4 = reinterpret_cast<float4 *>(sum);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pv = p4[i], gv = g4[i], sv = s4[i];
        const float4 p_old = pv, s_old = sv;
        float *pp = &pv.x, *gg = &gv.x, *ss = &sv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = fmaf(wd, pp[j], gg[j]);
            ss[j] = fmaf(gj, gj, ss[j]);
            pp[j] = pp[j] - lr * (gj / (sqrtf(ss[j]) + eps));
        }
        // store only what changed.  A row no gradient reached still moves by the weight-decay term (the reference applies
        // wd = 1e-10 to ALL rows, utils/optim.py:139-160) -- but once its accumulator has seen a real gradient, that term is
        // below half an ulp of both the accumulator and the parameter and the arithmetic above returns the old bits: such
        // rows (most token rows of a token-pooled step, the entities outside a sampled candidate list) then cost three
        // read streams instead of three reads + two writes.  Memory ends up bit-identical to the unconditional stores.
        if (bits_differ(pv, p_old)) p4[i] = pv;
        if (bits_differ(sv, s_old)) s4[i] = sv;
        // (clear only what is not clear already: rows no gradient reached -- most token rows of a token-pooled step, the
        //  entities outside a sampled candidate list -- cost one store stream less: a sixth of this HBM-bound sweep)
        if (zero_grad && ((__float_as_uint(gv.x) | __float_as_uint(gv.y) | __float_as_uint(gv.z) | __float_as_uint(gv.w)) != 0u))
            g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gj = fmaf(wd, p[i], g[i]);
        sum[i] = fmaf(gj, gj, sum[i]);
        p[i] = p[i] - lr * (gj / (sqrtf(sum[i]) + eps));
        if (zero_grad) g[i] = 0.f;
    }
}

__global__ __launch_bounds__(128) void encode_rows_kernel(const float *__restrict__ table, int64_t table_rows, int d,
                                                          const int32_t *__restrict__ ids, int first_id, const DropDev drop,
                                                          float *__restrict__ out, int64_t ld_out, int *__restrict__ id_err)
{
    const int i = blockIdx.x;
    const int64_t row = checked_row(ids ? (int64_t)ids[i] : (int64_t)first_id + i, table_rows, threadIdx.x == 0 ? id_err : nullptr);
    const float *src = table + row * d;
    float *dst = out + (size_t)i * ld_out;
    for (int k = threadIdx.x; k < d; k += blockDim.x) dst[k] = src[k] * drop_mult1(drop, (uint32_t)i, k, d);
}

__global__ __launch_bounds__(256) void scale_kernel(float *__restrict__ x, int64_t n, const float *__restrict__ alpha)
{
    const float a = *alpha;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] *= a;
}

// The drop-in path's gradients come out of the fused step already multiplied by `applied` = the 1 / normalizer the reference
// Trainer is known to divide by (trainer.py:221); autograd later hands over the factor it really used (*alpha).  Both
// tensors in one launch, and nothing but the scalar is read when the two agree (the normal case): x *= alpha / applied.
__global__ __launch_bounds__(256) void rescale2_kernel(float *__restrict__ x0, int64_t n0, float *__restrict__ x1, int64_t n1,
                                                       const float *__restrict__ alpha, float applied)
{
    const float r = *alpha / applied;
    if (r == 1.0f) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0; i += stride) x0[i] *= r;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n1; i += stride) x1[i] *= r;
}

struct AdagradSeg { float *p, *g, *s; int64_t n; };

__device__ __forceinline__ void adagrad_sweep(const AdagradSeg sg, float lr, float wd, float eps, int zero_grad,
                                              int64_t first, int64_t stride)
{
    const int64_t n4 = sg.n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(sg.p), *g4 = reinterpret_cast<float4 *>(sg.g), *s4 = reinterpret_cast<float4 *>(sg.s);
    for (int64_t i = first; i < n4; i += stride) {
        float4 pv = p4[i], gv = g4[i], sv = s4[i];
        const float4 p_old = pv, s_old = sv;
        float *pp = &pv.x, *gg = &gv.x, *ss = &sv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = fmaf(wd, pp[j], gg[j]);
            ss[j] = fmaf(gj, gj, ss[j]);
            pp[j] = pp[j] - lr * (gj / (sqrtf(ss[j]) + eps));
        }
        // store only what changed.  A row no gradient reached still moves by the weight-decay term (the reference applies
        // wd = 1e-10 to ALL rows, utils/optim.py:139-160) -- but once its accumulator has seen a real gradient, that term is
        // below half an ulp of both the accumulator and the parameter and the arithmetic above returns the old bits: such
        // rows (most token rows of a token-pooled step, the entities outside a sampled candidate list) then cost three
        // read streams instead of three reads + two writes.  Memory ends up bit-identical to the unconditional stores.
        if (bits_differ(pv, p_old)) p4[i] = pv;
        if (bits_differ(sv, s_old)) s4[i] = sv;
        // (clear only what is not clear already: rows no gradient reached -- most token rows of a token-pooled step, the
        //  entities outside a sampled candidate list -- cost one store stream less: a sixth of this HBM-bound sweep)
        if (zero_grad && ((__float_as_uint(gv.x) | __float_as_uint(gv.y) | __float_as_uint(gv.z) | __float_as_uint(gv.w)) != 0u))
            g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (first < (sg.n & 3)) {
        const int64_t i = (n4 << 2) + first;
        const float gj = fmaf(wd, sg.p[i], sg.g[i]);
        sg.s[i] = fmaf(gj, gj, sg.s[i]);
        sg.p[i] = sg.p[i] - lr * (gj / (sqrtf(sg.s[i]) + eps));
        if (zero_grad) sg.g[i] = 0.f;
    }
}

__global__ __launch_bounds__(256) void adagrad2_kernel(const AdagradSeg a, const AdagradSeg b, float lr, float wd,
                                                       float eps, int zero_grad)
{
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    adagrad_sweep(a, lr, wd, eps, zero_grad == 1, first, stride);      // zero_grad: 0 none, 1 both, 2 second tensor only
    adagrad_sweep(b, lr, wd, eps, zero_grad != 0, first, stride);
}

// Up to four tensors in one launch (token tables + batch-norm parameters of a token-pooled step: one launch instead of two),
// each optionally with a touched-row byte map: a row whose byte differs from the segment's stamp holds an all-zero gradient
// by contract (the pooling backward stamps every row it writes, okge_pool.hip pass 4), so its gradient is neither read nor
// cleared -- such rows still run through the arithmetic (the reference's wd * p reaches ALL rows) on p and sum alone.
// The stamp is never erased: the caller moves to another stamp after every update (a stale byte that meets its stamp again
// 255 updates later only costs the read of a gradient row that is zero).
struct AdagradSegsDev { AdagradSegM s[ADAGRAD_MAX_SEGS]; int n, unroll; };

// (U iterations of the grid-stride loop side by side: the map byte decides whether the gradient is loaded at all, so one
//  iteration is TWO dependent round trips -- run one at a time the sweep was latency-bound, slower than the map-less one)
template <int U, bool MAP>
__device__ __forceinline__ void adagrad_sweep_map(const AdagradSegM sg, float lr, float wd, float eps, int64_t first, int64_t stride)
{
    const int64_t n4 = sg.n >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(sg.p), *g4 = reinterpret_cast<float4 *>(sg.g), *s4 = reinterpret_cast<float4 *>(sg.s);
    const uint32_t row4 = MAP ? (uint32_t)(sg.row_len >> 2) : 1u;
    const int shift = (row4 & (row4 - 1)) == 0 ? 31 - __clz(row4) : -1;
    const uint8_t stamp = (uint8_t)sg.stamp;
    for (int64_t i0 = first; i0 < n4; i0 += U * stride) {
        bool ok[U], live[U];
        float4 pv[U], sv[U], gv[U];
        uint8_t mb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                            // the U map bytes first, no branch between the loads
            const int64_t i = min(i0 + u * stride, n4 - 1);
            ok[u] = i0 + u * stride < n4;
            mb[u] = stamp;
            if (MAP) mb[u] = sg.touched[shift >= 0 ? (uint32_t)i >> shift : (uint32_t)i / row4];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            live[u] = ok[u] && mb[u] == stamp;
            if (MAP && sg.rows == 1) { ok[u] = ok[u] && !live[u]; live[u] = false; }     // unstamped rows only (never a gradient read)
            if (MAP && sg.rows == 2) ok[u] = live[u];                                     // stamped rows only
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            pv[u] = sv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok[u]) { pv[u] = p4[i]; sv[u] = s4[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live[u]) gv[u] = g4[i0 + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            const int64_t i = i0 + u * stride;
            const float4 p_old = pv[u], s_old = sv[u];
            float *pp = &pv[u].x, *gg = &gv[u].x, *ss = &sv[u].x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gj = fmaf(wd, pp[j], gg[j]);
                ss[j] = fmaf(gj, gj, ss[j]);
                pp[j] = pp[j] - lr * (gj / (sqrtf(ss[j]) + eps));
            }
            if (bits_differ(pv[u], p_old)) p4[i] = pv[u];            // (store only what changed: see adagrad_sweep)
            if (bits_differ(sv[u], s_old)) s4[i] = sv[u];
            if (live[u] && sg.zero_grad && ((__float_as_uint(gv[u].x) | __float_as_uint(gv[u].y) | __float_as_uint(gv[u].z) | __float_as_uint(gv[u].w)) != 0u))
                g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (first < (sg.n & 3)) {                                    // (tail: only tensors without a map have one)
        const int64_t i = (n4 << 2) + first;
        const float gj = fmaf(wd, sg.p[i], sg.g[i]);
        sg.s[i] = fmaf(gj, gj, sg.s[i]);
        sg.p[i] = sg.p[i] - lr * (gj / (sqrtf(sg.s[i]) + eps));
        if (sg.zero_grad) sg.g[i] = 0.f;
    }
}

// rows == 1 on its own launch: the weight-decay-only update of the rows WITHOUT the stamp, built to run BESIDE the step's matrix
// kernels on another stream -- one-wave workgroups and few registers, so that a wave fits the register file the fused tile
// kernel leaves over (230 x 2 of 512 per SIMD at d = 256); two rows in flight per lane
__global__ __launch_bounds__(64) void adagrad_unstamped_kernel(const AdagradSegsDev segs, float lr, float wd, float eps)
{
    const int64_t first = (int64_t)blockIdx.x * 64 + threadIdx.x, stride = (int64_t)gridDim.x * 64;
    for (int k = 0; k < segs.n; ++k) {
        const AdagradSegM sg = segs.s[k];
        if (!sg.touched || sg.rows != 1) continue;
        const int64_t n4 = sg.n >> 2;
        float4 *p4 = reinterpret_cast<float4 *>(sg.p), *s4 = reinterpret_cast<float4 *>(sg.s);
        const uint32_t row4 = (uint32_t)(sg.row_len >> 2);
        const int shift = (row4 & (row4 - 1)) == 0 ? 31 - __clz(row4) : -1;
        const uint8_t stamp = (uint8_t)sg.stamp;
        for (int64_t i0 = first; i0 < n4; i0 += 2 * stride) {
            const int64_t i1 = i0 + stride;
            const bool ok1 = i1 < n4;
            const uint8_t m0 = sg.touched[shift >= 0 ? (uint32_t)i0 >> shift : (uint32_t)i0 / row4];
            const uint8_t m1 = ok1 ? sg.touched[shift >= 0 ? (uint32_t)i1 >> shift : (uint32_t)i1 / row4] : stamp;
            float4 pa = p4[i0], sa = s4[i0], pb = pa, sb = sa;
            if (ok1) { pb = p4[i1]; sb = s4[i1]; }
            const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 != stamp) {
                const float4 po = pa, so = sa;
                adagrad4(pa, zero, sa, lr, wd, eps);
                if (bits_differ(pa, po)) p4[i0] = pa;
                if (bits_differ(sa, so)) s4[i0] = sa;
            }
            if (ok1 && m1 != stamp) {
                const float4 po = pb, so = sb;
                adagrad4(pb, zero, sb, lr, wd, eps);
                if (bits_differ(pb, po)) p4[i1] = pb;
                if (bits_differ(sb, so)) s4[i1] = sb;
            }
        }
    }
}

__global__ __launch_bounds__(256) void adagrad_multi_kernel(const AdagradSegsDev segs, float lr, float wd, float eps)
{
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
#pragma unroll
    for (int k = 0; k < ADAGRAD_MAX_SEGS; ++k)
        if (k < segs.n) {
            if (segs.s[k].touched) {
                if (segs.unroll == 4) adagrad_sweep_map<4, true>(segs.s[k], lr, wd, eps, first, stride);
                else if (segs.unroll == 2) adagrad_sweep_map<2, true>(segs.s[k], lr, wd, eps, first, stride);
                else adagrad_sweep_map<1, true>(segs.s[k], lr, wd, eps, first, stride);
            } else adagrad_sweep_map<1, false>(segs.s[k], lr, wd, eps, first, stride);
        }
}

// ---- okge_adagrad_lazy: the weight-decay-only updates of rows no batch names, deferred ------------------------------------
// The reference's Adagrad reaches every row of a table every step (weight_decay 1e-10 makes every gradient row non-zero,
// utils/optim.py:139-160): at BASELINE configs[4] 85 % of the token rows are read, moved by their own decay term and written
// back per step -- 1 GB of HBM traffic, 175 us of a 0.78 ms step.  Such a row's update depends on (p, sum) of that row alone,
// so it can be applied LATER, all pending steps at once in registers, as long as it has happened before anything reads the row:
//   row_steps[r]   number of optimizer steps row r has seen; counters[0] = T, the steps taken
//   STEP           rows the backward stamped: their T - row_steps[r] pending decay steps, then this step with the gradient;
//                  rows with r % window == T % window: their T + 1 - row_steps[r] pending decay steps; T += 1 (last workgroup)
//   catch-up       (okge_pool.hip, before the pooling forward) brings the rows the batch's tokens name to T
//   FLUSH          every row to T (before anything else reads the tables: evaluation, checkpoints, the host)
// Per step the sweep touches the stamped rows and 1 / window of the others: the traffic falls by ~window, the arithmetic
// (correctly rounded sqrt and division per element and pending step) stays and becomes the bound.  Same operations in the
// same order per element as the eager sweep: the tables are bit-identical after a FLUSH (tests/test_token_pooled.py).
struct LazySegsDev { LazySeg s[ADAGRAD_MAX_SEGS]; int n; int64_t batch0[ADAGRAD_MAX_SEGS + 1]; };

constexpr int LAZY_BATCH = 16;         // rows per wave and turn: many short turns hide the rows' load latency

__device__ __forceinline__ void adagrad_lazy_body(const LazySegsDev &segs, int32_t *counters, int window, int mode, float lr, float wd,
                                                  float eps, int lazy_batch)
{
    const int T = counters[0], target = mode == LAZY_STEP ? T + 1 : T;
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * 4;
    const int phase = T % window;
    // tensors with a row_steps array: batches of LAZY_BATCH rows, one per wave at a time; lane = row while the batch's bytes
    // and counters are read, lane = column quad while a row is updated
    for (int64_t b = gw; b < segs.batch0[segs.n]; b += nw) {
        int k = 0;
        while (b >= segs.batch0[k + 1]) ++k;
        const LazySeg &sg = segs.s[k];
        const int64_t r = (b - segs.batch0[k]) * lazy_batch + (lane & (lazy_batch - 1));
        const bool in = lane < lazy_batch && r < sg.rows;
        const int up = in ? sg.steps[r] : target;
        const bool stamped = mode == LAZY_STEP && in && sg.touched && sg.touched[r] == (uint8_t)sg.stamp;
        const bool due = in && (mode == LAZY_FLUSH || (int)((uint32_t)r % (uint32_t)window) == phase);      // (rows < 2^31: okge_api.hip)
        const bool need = stamped || (due && up < target);
        lazy_rows(__ballot(need), r, max(0, (stamped ? T : target) - up), stamped, sg.p, sg.s, sg.g, sg.row_len, lane, lr, wd, eps);
        if (need) {
            sg.steps[r] = target;
            if (stamped) sg.touched[r] = 0;              // the map is clean again: a stamp never has to move on
        }
    }
    // plain dense tensors (batch-norm parameters): every element every step
    if (mode == LAZY_STEP) {
        const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
        for (int k = 0; k < segs.n; ++k) {
            const LazySeg &sg = segs.s[k];
            if (sg.steps) continue;
            const int64_t n = sg.rows * sg.row_len;
            for (int64_t i = first; i < n; i += stride) {
                const float gj = fmaf(wd, sg.p[i], sg.g[i]);
                sg.s[i] = fmaf(gj, gj, sg.s[i]);
                sg.p[i] = sg.p[i] - lr * (gj / (sqrtf(sg.s[i]) + eps));
                sg.g[i] = 0.f;
            }
        }
        // every workgroup has read T by the time the last one arrives here
        __shared__ int last;
        __syncthreads();
        if (threadIdx.x == 0) last = atomicAdd(&counters[1], 1) == (int)gridDim.x - 1;
        __syncthreads();
        if (last && threadIdx.x == 0) { counters[1] = 0; counters[0] = T + 1; }
    }
}

// 93 registers = 5 waves per SIMD.  Budgets of 80 / 64 registers (6 / 8 waves) spill inside the replay loops: 85 / 130 us against
// 70 us at configs[4] (profiles/round4_ablation.md section 6)
__global__ __launch_bounds__(256, 5) void adagrad_lazy_kernel(const LazySegsDev segs, int32_t *counters, int window, int mode, float lr,
                                                           float wd, float eps, int lazy_batch)
{
    adagrad_lazy_body(segs, counters, window, mode, lr, wd, eps, lazy_batch);
}

constexpr int RANK_GROUPS = 8;

// Count, for NG answer groups of one row at once, how many (filter-corrected) scores are greater than / equal to
// each group's true score.  NG is the compile-time number of live groups (1, 2, 4 or 8): the sweep is VALU-bound
// on the compares, and most rows have one or two groups.
template <int NG>
__device__ __forceinline__ void rank_sweep(const float *__restrict__ row, int64_t ld_is_vec, int N,
                                           const int32_t *__restrict__ filt_col, int64_t f_lo, int64_t f_hi, int col0,
                                           const float *tv, int tid, int (&gt)[RANK_GROUPS], int (&eq)[RANK_GROUPS])
{
    float t[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) t[j] = tv[j];
    int g[NG], e[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) { g[j] = 0; e[j] = 0; }
    const int n4 = ld_is_vec ? (N >> 2) : 0;
    const float4 *row4 = reinterpret_cast<const float4 *>(row);
    // 16-byte loads, four in flight per thread: a row is 58 KB at the FB15k-237 size and the sweep is a chain of memory
    // round trips, not compares
    for (int i = tid; i < n4; i += 1024) {
        float4 xv[4];
        bool has[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            has[u] = i + 256 * u < n4;
            xv[u] = has[u] ? row4[i + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!has[u]) continue;
            const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int j = 0; j < NG; ++j) { g[j] += xs[k] > t[j]; e[j] += xs[k] == t[j]; }
            }
        }
    }
    for (int n = 4 * n4 + tid; n < N; n += 256) {
        const float x = row[n];
#pragma unroll
        for (int j = 0; j < NG; ++j) { g[j] += x > t[j]; e[j] += x == t[j]; }
    }
    // filtered positions count as -1e8 instead of their score (dataset.py:441)
    for (int64_t f = f_lo + tid; f < f_hi; f += 256) {
        const int col = filt_col[f] - col0;                     // a shard only sees its own candidate columns
        if (col < 0 || col >= N) continue;
        const float x = row[col];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            g[j] += (-1e8f > t[j]) - (x > t[j]);
            e[j] += (-1e8f == t[j]) - (x == t[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < RANK_GROUPS; ++j) { gt[j] = j < NG ? g[j < NG ? j : 0] : 0; eq[j] = j < NG ? e[j < NG ? j : 0] : 0; }
}

// The same counts with the thread's part of the row already in registers (rows up to RANK_REG_ROW floats): the row
// loads are issued at kernel entry and overlap the dependent-load chain that finds the true scores
// (row_ptr -> grp_ptr -> ids -> score), and several group chunks reuse them.
constexpr int RANK_REG_F4 = 16, RANK_REG_ROW = 256 * 4 * RANK_REG_F4;

template <int NG>
__device__ __forceinline__ void rank_sweep_regs(const float4 (&rv)[RANK_REG_F4], int n4, float tail_x, bool has_tail, float filt_x,
                                                bool has_filt, const float *__restrict__ row, int N,
                                                const int32_t *__restrict__ filt_col, int64_t f_lo, int64_t f_hi, int col0,
                                                const float *tv, int tid, int (&gt)[RANK_GROUPS], int (&eq)[RANK_GROUPS])
{
    float t[NG];
    int g[NG], e[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) { t[j] = tv[j]; g[j] = 0; e[j] = 0; }
#pragma unroll
    for (int u = 0; u < RANK_REG_F4; ++u) {
        if (tid + 256 * u >= n4) continue;
        const float xs[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < NG; ++j) { g[j] += xs[k] > t[j]; e[j] += xs[k] == t[j]; }
        }
    }
    if (has_tail) {
#pragma unroll
        for (int j = 0; j < NG; ++j) { g[j] += tail_x > t[j]; e[j] += tail_x == t[j]; }
    }
    // filtered positions count as -1e8 instead of their score (dataset.py:441); the first 256 were fetched at entry
    if (has_filt) {
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            g[j] += (-1e8f > t[j]) - (filt_x > t[j]);
            e[j] += (-1e8f == t[j]) - (filt_x == t[j]);
        }
    }
    for (int64_t f = f_lo + 256 + tid; f < f_hi; f += 256) {
        const int col = filt_col[f] - col0;
        if (col < 0 || col >= N) continue;
        const float x = row[col];
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            g[j] += (-1e8f > t[j]) - (x > t[j]);
            e[j] += (-1e8f == t[j]) - (x == t[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < RANK_GROUPS; ++j) { gt[j] = j < NG ? g[j < NG ? j : 0] : 0; eq[j] = j < NG ? e[j < NG ? j : 0] : 0; }
}

__global__ __launch_bounds__(256) void ranks_kernel(const float *__restrict__ scores, int64_t ld, int N,
                                                    const int64_t *__restrict__ filt_ptr,
                                                    const int32_t *__restrict__ filt_col,
                                                    const int64_t *__restrict__ row_ptr,
                                                    const int64_t *__restrict__ grp_ptr, const int32_t *__restrict__ ids,
                                                    int64_t *__restrict__ ranks, int col0,
                                                    const float *__restrict__ true_in, float *__restrict__ true_out,
                                                    int64_t *__restrict__ counts_out)
{
    // Whole candidate list (col0 = 0, true_in = true_out = counts_out = null): ranks.
    // Candidate-sharded evaluation runs it twice around two tiny all-reduces:
    //   true_out   : the shard's maximum over each group's ids that fall in [col0, col0 + N)   -> all-reduce(max)
    //   true_in    : global true scores in, counts_out[g] = {#greater, #equal} of this shard   -> all-reduce(sum)
    // blockIdx.y strides over the row's chunks of 8 answer groups: rows with many answers (they set the kernel's
    // duration) are spread over gridDim.y workgroups, the others leave at once
    __shared__ float tv[RANK_GROUPS];
    __shared__ int cnt[4][2 * RANK_GROUPS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int64_t g_lo = row_ptr[b] + (int64_t)RANK_GROUPS * blockIdx.y, g_hi = row_ptr[b + 1];
    if (g_lo >= g_hi) return;
    const float *row = scores + (size_t)b * ld;
    const int64_t f_lo = filt_ptr[b], f_hi = filt_ptr[b + 1];
    const int64_t vec = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(scores) & 15) == 0);
    const bool in_regs = vec && N <= RANK_REG_ROW && !true_out;
    const int n4 = N >> 2;
    float4 rv[RANK_REG_F4];
    float tail_x = 0.f, filt_x = 0.f;
    bool has_tail = false, has_filt = false;
    if (in_regs) {
        // everything that does not depend on the true scores is requested now: the row, its (< 4) tail elements and
        // the scores under the first 256 filter entries
#pragma unroll
        for (int u = 0; u < RANK_REG_F4; ++u)
            rv[u] = tid + 256 * u < n4 ? reinterpret_cast<const float4 *>(row)[tid + 256 * u] : make_float4(0.f, 0.f, 0.f, 0.f);
        has_tail = 4 * n4 + tid < N;
        if (has_tail) tail_x = row[4 * n4 + tid];
        if (f_lo + tid < f_hi) {
            const int col = filt_col[f_lo + tid] - col0;
            has_filt = col >= 0 && col < N;
            if (has_filt) filt_x = row[col];
        }
    }
    for (int64_t g0 = g_lo; g0 < g_hi; g0 += (int64_t)RANK_GROUPS * gridDim.y) {
        const int ng = (int)min((int64_t)RANK_GROUPS, g_hi - g0);
        if (tid < RANK_GROUPS) {
            float t = __builtin_nanf("");                       // unused slots never compare true
            if (tid < ng) {
                if (true_in) {
                    t = true_in[g0 + tid];
                } else {
                    t = -INFINITY;
                    for (int64_t j = grp_ptr[g0 + tid]; j < grp_ptr[g0 + tid + 1]; ++j) {
                        const int col = ids[j] - col0;
                        if (col >= 0 && col < N) t = fmaxf(t, row[col]);
                    }
                    if (true_out) true_out[g0 + tid] = t;
                }
            }
            tv[tid] = t;
        }
        if (true_out) continue;                                 // uniform: phase 1 of the sharded evaluation
        __syncthreads();
        int gt[RANK_GROUPS], eq[RANK_GROUPS];
        if (in_regs) {
            if (ng == 1)      rank_sweep_regs<1>(rv, n4, tail_x, has_tail, filt_x, has_filt, row, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
            else if (ng == 2) rank_sweep_regs<2>(rv, n4, tail_x, has_tail, filt_x, has_filt, row, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
            else if (ng <= 4) rank_sweep_regs<4>(rv, n4, tail_x, has_tail, filt_x, has_filt, row, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
            else              rank_sweep_regs<8>(rv, n4, tail_x, has_tail, filt_x, has_filt, row, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
        } else if (ng == 1)   rank_sweep<1>(row, vec, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
        else if (ng == 2)     rank_sweep<2>(row, vec, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
        else if (ng <= 4)     rank_sweep<4>(row, vec, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
        else                  rank_sweep<8>(row, vec, N, filt_col, f_lo, f_hi, col0, tv, tid, gt, eq);
#pragma unroll
        for (int j = 0; j < RANK_GROUPS; ++j) {
            const int a = wave_sum(gt[j]), e = wave_sum(eq[j]);
            if ((tid & 63) == 0) { cnt[tid >> 6][2 * j] = a; cnt[tid >> 6][2 * j + 1] = e; }
        }
        __syncthreads();
        if (tid < ng) {
            const int64_t a = (int64_t)cnt[0][2 * tid] + cnt[1][2 * tid] + cnt[2][2 * tid] + cnt[3][2 * tid];
            const int64_t e = (int64_t)cnt[0][2 * tid + 1] + cnt[1][2 * tid + 1] + cnt[2][2 * tid + 1] + cnt[3][2 * tid + 1];
            if (counts_out) {
                counts_out[2 * (g0 + tid)] = a;
                counts_out[2 * (g0 + tid) + 1] = e;
            } else {
                ranks[g0 + tid] = a + e / 2;
            }
        }
        __syncthreads();
    }
}

// acc[0..6] += {#groups, sum 1/(rank+1), sum rank, #rank<1, #rank<3, #rank<10, #rank<50}: the meters of compute_metrics
// (dataset.py:447-452) kept on the device, so that evaluation batches need no host round trip
__global__ __launch_bounds__(256) void rank_metrics_kernel(const int64_t *__restrict__ ranks, int64_t n, double *__restrict__ acc)
{
    __shared__ double red[4][7];
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const int64_t r = ranks[i];
        v[0] += 1.0;
        v[1] += (double)(1.0f / (float)(r + 1));       // fp32 reciprocal like the reference's (1/(rank+1).float())
        v[2] += (double)r;
        v[3] += r < 1; v[4] += r < 3; v[5] += r < 10; v[6] += r < 50;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double t = wave_sum(v[k]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
    }
    __syncthreads();
    // atomic: launches of independent stream chains may share one `acc` (PipelinedEvaluator hands every chain its own
    // and sums them after the join, which also keeps the double sums order-deterministic)
    if (threadIdx.x < 7)
        atomicAdd(acc + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// Per-triple score of already encoded rows, Hadamard form (model.py:231-238, :276): one wave per triple.
__global__ __launch_bounds__(256) void score_triples_kernel(const float *__restrict__ S, int64_t lds_,
                                                            const float *__restrict__ Rr, int64_t ldr,
                                                            const float *__restrict__ O, int64_t ldo, int n, int d,
                                                            int scorer, float *__restrict__ out)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float *s = S + (size_t)i * lds_, *r = Rr + (size_t)i * ldr, *o = O + (size_t)i * ldo;
    float acc = 0.f;
    if (scorer == SC_DISTMULT) {
        for (int k = lane; k < d; k += 64) acc += s[k] * o[k] * r[k];
    } else {
        const int h = d >> 1;
        for (int k = lane; k < h; k += 64) {
            const float s1 = s[k], s2 = s[h + k], r1 = r[k], r2 = r[h + k], o1 = o[k], o2 = o[h + k];
            acc += s1 * o1 * r1 + s2 * o2 * r1 + s1 * o2 * r2 - s2 * o1 * r2;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) out[i] = acc;
}

// ---- gradient clipping (trainer.py:236-240: torch.nn.utils.clip_grad_norm_) and the sharded KL loss' row log-sum-exp --
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float *__restrict__ g0, int64_t n0, const float *__restrict__ g1,
                                                             int64_t n1, double *__restrict__ partial)
{
    __shared__ double red[4];
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = first; i < n0; i += stride) acc += (double)g0[i] * (double)g0[i];
    for (int64_t i = first; i < n1; i += stride) acc += (double)g1[i] * (double)g1[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void clip_coef_kernel(const double *__restrict__ partial, int n, float max_norm,
                                                        float *__restrict__ coef, double *__restrict__ norm_out)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += partial[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(red[0] + red[1] + red[2] + red[3]);        // torch keeps the norm in fp32
        const float c = max_norm / (total + 1e-6f);
        coef[0] = c < 1.f ? c : 1.f;
        if (norm_out) norm_out[0] = (double)total;
    }
}

__global__ __launch_bounds__(256) void merge_lse_kernel(const float *__restrict__ parts, int world, int B, float *__restrict__ out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -INFINITY;
    for (int r = 0; r < world; ++r) m = fmaxf(m, parts[(size_t)r * B + b]);
    float sacc = 0.f;
    for (int r = 0; r < world; ++r) sacc += expf(parts[(size_t)r * B + b] - m);
    out[b] = m > -INFINITY ? m + logf(sacc) : -INFINITY;
}

// ---- fused evaluation (okge_evaluate_fused): the side work (okge_eval_device.h) ------------------------------------
// The two small kernels of the fused evaluation in ONE launch: workgroups [0, n_points) take the rows of `pts`, the rest
// the groups of `rk`.  In a run of batches the points of batch i+1 ride with the ranks of batch i (independent work of
// two batches; either part may be empty).
__global__ __launch_bounds__(256) void eval_side_kernel(const EvalPointsArgs pts, const EvalRanksArgs rk, int n_points)
{
    if ((int)blockIdx.x < n_points) eval_points_block(pts, blockIdx.x);
    else eval_ranks_block(rk, (int)blockIdx.x - n_points);
}

// ---- launchers -----------------------------------------------------------------------------------------
hipError_t launch_score_triples(const float *S, int64_t lds_, const float *Rr, int64_t ldr, const float *O, int64_t ldo,
                                int n, int d, int scorer, float *out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(score_triples_kernel, dim3((n + 3) / 4), dim3(256), 0, st, S, lds_, Rr, ldr, O, ldo, n, d, scorer,
                       out);
    return hipGetLastError();
}

hipError_t launch_encode_queries(const float *E, const float *R, int d, int scorer, const PrefixDev &p, float *Q,
                                 int ldq, int Bpad, float *ent_rows, const int32_t *pos_col, int nnz, int32_t *tile_ptr,
                                 int tiles, int tile_w, int cand_col0, hipStream_t st, const ClearSpec *clear)
{
    const int extra = tile_ptr ? (tiles + 1 + 127) / 128 : 0;
    ClearSpec clr = {};
    int64_t clear_wgs = 0;
    if (clear) {
        clr = *clear;
        int64_t f4 = 0;
        for (int r = 0; r < CLEAR_REGIONS; ++r) f4 += clr.p[r] ? (clr.n[r] + 3) / 4 : 0;
        for (int r = 0; r < CLEAR_REGIONS; ++r) if (!clr.p[r]) clr.n[r] = 0;
        clear_wgs = (f4 + 127) / 128;
    }
    if (Bpad + extra + clear_wgs <= 0) return hipSuccess;
    hipLaunchKernelGGL(encode_queries_kernel, dim3((unsigned)(Bpad + extra + clear_wgs)), dim3(128), 0, st, E, R, d, scorer, p, Q,
                       ldq, Bpad, ent_rows, pos_col, nnz, tile_ptr, tiles, tile_w, cand_col0, extra, clr);
    return hipGetLastError();
}

hipError_t launch_fold_queries(const float *R, int d, int scorer, const PrefixDev &p, const float *ent_rows, float *Q, int ldq,
                               int Bpad, hipStream_t st)
{
    if (Bpad <= 0) return hipSuccess;
    hipLaunchKernelGGL(fold_queries_kernel, dim3(Bpad), dim3(128), 0, st, R, d, scorer, p, ent_rows, Q, ldq);
    return hipGetLastError();
}

hipError_t launch_slab_reduce(const float *slab, int nsplit, int64_t n, float *out, const double *loss_partials,
                              int n_partials, double *loss_out, hipStream_t st)
{
    const int64_t n4 = n / 4;
    if (n4 <= 0) return hipSuccess;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n4 + 255) / 256) + 1), dim3(256), 0, st, slab, nsplit, n4, out,
                       loss_partials, n_partials, loss_out);
    return hipGetLastError();
}

hipError_t launch_dc_reduce(const float *slab, int nsplit, int rows_pad, int D16, int N, int d, const int32_t *cand_ids,
                            int cand_first, int exclusive, int grads_zero, float *dE, int64_t table_rows, int *id_err,
                            hipStream_t st)
{
    const int64_t total = (int64_t)N * (D16 / 4);
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(dc_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slab, nsplit, rows_pad, D16, N,
                       d, cand_ids, cand_first, exclusive, grads_zero, dE, table_rows, id_err);
    return hipGetLastError();
}

hipError_t launch_dc_reduce_streamk(const float *slab, int tiles, int chunks_per_tile, int workgroups, int D16, int N, int d,
                                    const int32_t *cand_ids, int cand_first, int exclusive, int grads_zero, float *dE,
                                    int64_t table_rows, int *id_err, hipStream_t st)
{
    if (tiles <= 0) return hipSuccess;
    if (workgroups > SK_MAX_WGS) return hipErrorInvalidValue;      // (slot list of a tile: at most two entries per workgroup)
    hipLaunchKernelGGL(dc_reduce_streamk_kernel, dim3(8 * tiles), dim3(256), 0, st, slab, tiles, chunks_per_tile, workgroups, D16,
                       N, d, cand_ids, cand_first, exclusive, grads_zero, dE, table_rows, id_err);
    return hipGetLastError();
}

hipError_t launch_prefix_backward(const float *E, const float *R, int d, int scorer, const PrefixDev &p,
                                  const float *slab, int nsplit, int Bpad, int ldq, const float *ent_rows, float *dE,
                                  float *dR, const double *loss_partials, int n_partials, double *loss_out,
                                  hipStream_t st, const int32_t *rel_order, const int32_t *rel_seg_ptr, int n_rel_seg,
                                  const int32_t *ent_order, const int32_t *ent_seg_ptr, int n_ent_seg, float *grad_rows,
                                  int distinct, const AdagradFuse *fuse)
{
    const int B = p.n_po + p.n_sp;
    if (B <= 0) return hipSuccess;
    const bool vec = scorer == SC_DISTMULT ? (d % 4 == 0) : (d % 8 == 0);
    if (fuse && !vec) return hipErrorInvalidValue;
    if (vec) {
        // grad_rows: [2][Bpad][ldq] scratch -- relation rows, then entity rows
        const bool seg_r = grad_rows && rel_order && rel_seg_ptr && n_rel_seg > 0;
        const bool seg_e = grad_rows && ent_order && ent_seg_ptr && n_ent_seg > 0;
        float *dr_rows = seg_r ? grad_rows : nullptr, *de_rows = seg_e ? grad_rows + (size_t)Bpad * ldq : nullptr;
        AdagradFuse af = {};
        int sweep_wgs = 0;
        if (fuse) {
            af = *fuse;
            sweep_wgs = (int)std::min<int64_t>(16384, (af.n_ent * (d / 4) + 127) / 128);     // one float4 per thread up to 2 M of them
        }
        const int cols = scorer == SC_DISTMULT ? d : d / 2;
        const int chunks = fuse ? 1 : std::min(8, (cols + 127) / 128);       // (the fused sweep's workgroups count on a 1-D grid)
        if (nsplit >= 8)
            hipLaunchKernelGGL(prefix_backward_vec_kernel<8>, dim3(B + 1 + sweep_wgs, chunks), dim3(128), 0, st, E, R, d, scorer, p, slab,
                               nsplit, Bpad, ldq, ent_rows, dE, dR, loss_partials, n_partials, loss_out, dr_rows, de_rows, distinct, af, B);
        else
            hipLaunchKernelGGL(prefix_backward_vec_kernel<1>, dim3(B + 1 + sweep_wgs, chunks), dim3(128), 0, st, E, R, d, scorer, p, slab,
                               nsplit, Bpad, ldq, ent_rows, dE, dR, loss_partials, n_partials, loss_out, dr_rows, de_rows, distinct, af, B);
        if (seg_r || seg_e) {
            const RowSegments rel{rel_order, rel_seg_ptr, seg_r ? n_rel_seg : 0}, ent{ent_order, ent_seg_ptr, seg_e ? n_ent_seg : 0};
            hipLaunchKernelGGL(row_segment_sum_kernel, dim3(rel.n_seg + ent.n_seg), dim3(128), 0, st, dr_rows, de_rows, ldq, d, p,
                               rel, ent, dR, dE);
        }
    } else {
        hipLaunchKernelGGL(prefix_backward_kernel, dim3(B), dim3(128), 0, st, E, R, d, scorer, p, slab, nsplit, Bpad, ldq,
                           ent_rows, dE, dR, distinct);
        if (loss_partials)
            hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, loss_partials, n_partials, loss_out);
    }
    return hipGetLastError();
}

hipError_t launch_adagrad_finish(const AdagradFuse &af, const PrefixDev &p, hipStream_t st)
{
    const int B = p.n_po + p.n_sp;
    const int r_wgs = (int)std::min<int64_t>(256, ((int64_t)af.n_rel * (af.d / 4) + 127) / 128);
    if (B + r_wgs <= 0) return hipSuccess;
    hipLaunchKernelGGL(adagrad_finish_kernel, dim3(B + r_wgs), dim3(128), 0, st, af, p, B);
    return hipGetLastError();
}

hipError_t launch_loss_reduce(const double *partials, int n, double *loss_out, hipStream_t st)
{
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, partials, n, loss_out);
    return hipGetLastError();
}

hipError_t launch_kl_count_pos(const int32_t *pos_row, int nnz, int Bpad, float *row_ysum, hipStream_t st)
{
    (void)Bpad;                                        // (row_ysum was cleared by the caller's encode launch: see train_core)
    if (nnz > 0) hipLaunchKernelGGL(kl_count_pos_kernel, dim3((nnz + 255) / 256), dim3(256), 0, st, pos_row, nnz, row_ysum);
    return hipGetLastError();
}

hipError_t launch_kl_row_lse(const float *stats, int tiles, int B, int Bpad, float *run, int first, int last, float *row_lse,
                             hipStream_t st, const int32_t *pos_row, int nnz, float *row_ysum)
{
    if (tiles <= 0 || B <= 0) return hipSuccess;
    const int row_blocks = (B + 3) / 4, count_blocks = (row_ysum && nnz > 0) ? (nnz + 255) / 256 : 0;
    hipLaunchKernelGGL(kl_row_lse_kernel, dim3(row_blocks + count_blocks), dim3(256), 0, st, stats, tiles, B, Bpad,
                       reinterpret_cast<float2 *>(run), first, last, row_lse, row_blocks, pos_row, nnz, row_ysum);
    return hipGetLastError();
}

hipError_t launch_adagrad(float *p, float *g, float *sum, int64_t n, float lr, float wd, float eps, int zero_grad,
                          hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int64_t n4 = (n + 3) / 4;
    const int blocks = (int)min((int64_t)16384, (n4 + 255) / 256);
    hipLaunchKernelGGL(adagrad_kernel, dim3(blocks), dim3(256), 0, st, p, g, sum, n, lr, wd, eps, zero_grad);
    return hipGetLastError();
}

hipError_t launch_encode_rows(const float *table, int64_t table_rows, int d, const int32_t *ids, int first_id, int n,
                              const DropDev &drop, float *out, int64_t ld_out, int *id_err, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(encode_rows_kernel, dim3(n), dim3(128), 0, st, table, table_rows, d, ids, first_id, drop, out, ld_out, id_err);
    return hipGetLastError();
}

hipError_t launch_scale(float *x, int64_t n, const float *alpha_dev, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int blocks = (int)std::min((int64_t)4096, (n + 255) / 256);
    hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, st, x, n, alpha_dev);
    return hipGetLastError();
}

hipError_t launch_rescale2(float *x0, int64_t n0, float *x1, int64_t n1, const float *alpha_dev, float applied, hipStream_t st)
{
    const int64_t n = std::max(n0, n1);
    if (n <= 0) return hipSuccess;
    const int blocks = (int)std::min((int64_t)2048, (n + 255) / 256);
    hipLaunchKernelGGL(rescale2_kernel, dim3(blocks), dim3(256), 0, st, x0, n0, x1, n1, alpha_dev, applied);
    return hipGetLastError();
}

hipError_t launch_adagrad2(float *p0, float *g0, float *s0, int64_t n0, float *p1, float *g1, float *s1, int64_t n1,
                           float lr, float wd, float eps, int zero_grad, hipStream_t st)
{
    const int64_t n4 = (std::max(n0, n1) + 3) / 4;
    if (n4 <= 0) return hipSuccess;
    const int blocks = (int)std::min((int64_t)16384, (n4 + 255) / 256);
    const AdagradSeg a{p0, g0, s0, n0}, b{p1, g1, s1, n1};
    hipLaunchKernelGGL(adagrad2_kernel, dim3(blocks), dim3(256), 0, st, a, b, lr, wd, eps, zero_grad);
    return hipGetLastError();
}

hipError_t launch_adagrad_multi(const AdagradSegM *segs, int n_segs, float lr, float wd, float eps, hipStream_t st)
{
    if (n_segs <= 0) return hipSuccess;
    if (n_segs > ADAGRAD_MAX_SEGS) return hipErrorInvalidValue;
    AdagradSegsDev a;
    std::memset(&a, 0, sizeof(a));
    a.n = n_segs;
    static const int unroll = getenv("OKGE_ADAGRAD_U") ? atoi(getenv("OKGE_ADAGRAD_U")) : 4;
    a.unroll = unroll;
    int64_t n4 = 0;
    for (int k = 0; k < n_segs; ++k) {
        a.s[k] = segs[k];
        n4 = std::max(n4, (segs[k].n + 3) / 4);
    }
    if (n4 <= 0) return hipSuccess;
    bool all_unstamped = true;
    for (int k = 0; k < n_segs; ++k) all_unstamped = all_unstamped && segs[k].rows == 1 && segs[k].touched;
    if (all_unstamped) {                                 // the lean kernel that runs beside the matrix kernels
        const int blocks = (int)std::min((int64_t)32768, (n4 + 127) / 128);
        hipLaunchKernelGGL(adagrad_unstamped_kernel, dim3(blocks), dim3(64), 0, st, a, lr, wd, eps);
        return hipGetLastError();
    }
    const int blocks = (int)std::min((int64_t)16384, (n4 + 255) / 256);
    hipLaunchKernelGGL(adagrad_multi_kernel, dim3(blocks), dim3(256), 0, st, a, lr, wd, eps);
    return hipGetLastError();
}

hipError_t launch_adagrad_lazy(const LazySeg *segs, int n_segs, int32_t *counters, int window, int mode, float lr, float wd,
                               float eps, hipStream_t st)
{
    if (n_segs <= 0) return hipSuccess;
    if (n_segs > ADAGRAD_MAX_SEGS || window < 1 || !counters) return hipErrorInvalidValue;
    LazySegsDev a;
    std::memset(&a, 0, sizeof(a));
    a.n = n_segs;
    static const int lazy_batch_env = getenv("OKGE_LAZY_BATCH") ? atoi(getenv("OKGE_LAZY_BATCH")) : LAZY_BATCH;
    const int lazy_batch = (lazy_batch_env == 8 || lazy_batch_env == 32 || lazy_batch_env == 64) ? lazy_batch_env : LAZY_BATCH;
    for (int k = 0; k < n_segs; ++k) {
        a.s[k] = segs[k];
        a.batch0[k + 1] = a.batch0[k] + (segs[k].steps ? (segs[k].rows + lazy_batch - 1) / lazy_batch : 0);
    }
    // (every wave resident: 5 per SIMD; a wave takes batch after batch)
    const int64_t wgs = std::max<int64_t>(1, std::min<int64_t>((a.batch0[n_segs] + 3) / 4, 256 * 5));
    hipLaunchKernelGGL(adagrad_lazy_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a, counters, window, mode, lr, wd, eps, lazy_batch);
    return hipGetLastError();
}

hipError_t launch_rank_metrics(const int64_t *ranks, int64_t n, double *acc, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(1), dim3(256), 0, st, ranks, n, acc);
    return hipGetLastError();
}

hipError_t launch_ranks(const float *scores, int64_t ld, int B, int N, const int64_t *filt_ptr,
                        const int32_t *filt_col, const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                        int64_t *ranks, int col0, const float *true_in, float *true_out, int64_t *counts_out,
                        hipStream_t st)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(ranks_kernel, dim3(B, 4), dim3(256), 0, st, scores, ld, N, filt_ptr, filt_col, row_ptr, grp_ptr,
                       ids, ranks, col0, true_in, true_out, counts_out);
    return hipGetLastError();
}

}  // namespace okge

namespace okge {

hipError_t launch_eval_side(const EvalPointsArgs *pts, const EvalRanksArgs *rk, hipStream_t st)
{
    const int n_points = pts ? pts->Bpad : 0;
    const int n_ranks = rk && rk->n_groups > 0 ? (int)((rk->n_groups + 3) / 4) : 0;
    if (n_points + n_ranks <= 0) return hipSuccess;
    static const EvalPointsArgs no_pts = {};
    static const EvalRanksArgs no_rk = {};
    hipLaunchKernelGGL(eval_side_kernel, dim3(n_points + n_ranks), dim3(256), 0, st, pts ? *pts : no_pts, rk ? *rk : no_rk, n_points);
    return hipGetLastError();
}

}  // namespace okge

namespace okge {

hipError_t launch_clip_coef(const float *g0, int64_t n0, const float *g1, int64_t n1, float max_norm, double *partial,
                            int n_partial, float *coef_dev, double *norm_dev, hipStream_t st)
{
    hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(n_partial), dim3(256), 0, st, g0, n0, g1, n1, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, st, partial, n_partial, max_norm, coef_dev, norm_dev);
    return hipGetLastError();
}

hipError_t launch_merge_lse(const float *parts, int world, int B, float *out, hipStream_t st)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(merge_lse_kernel, dim3((B + 255) / 256), dim3(256), 0, st, parts, world, B, out);
    return hipGetLastError();
}

}  // namespace okge
