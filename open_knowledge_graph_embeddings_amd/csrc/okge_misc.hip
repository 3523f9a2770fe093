// Small HBM-bound kernels around the fused tile kernels (gfx950):
//   encode_queries_kernel   gather + dropout + fold (s,r)/(r,o) into one query row   model.py:455-510, :205-216, :269-272
//   prefix_backward_kernel  sum dQ slabs, chain rule to s/r/o rows, scatter-add       autograd of the above + embedding_dense_backward
//   loss_reduce_kernel      deterministic sum of per-workgroup loss partials           trainer.py:106 (reduction='sum')
//   kl_* kernels            log-sum-exp per row from tile partials, positives per row  trainer.py:99-100
//   adagrad_kernel          dense Adagrad sweep (+ zero_grad)                          utils/optim.py:139-160 / torch.optim.Adagrad
//   ranks_kernel            filtered ranks, exact integer counts                       dataset.py:423-446
#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

struct RowSrc {
    int64_t ent, rel;
    uint32_t pos;
    bool sp;
};

__device__ __forceinline__ RowSrc row_source(const PrefixDev &p, int b)
{
    RowSrc r;
    if (b < p.n_po) {
        r.rel = p.po_rel[b]; r.ent = p.po_obj[b]; r.pos = (uint32_t)b; r.sp = false;
    } else {
        const int i = b - p.n_po;
        r.ent = p.sp_subj[i]; r.rel = p.sp_rel[i]; r.pos = (uint32_t)i; r.sp = true;
    }
    return r;
}

__global__ __launch_bounds__(128) void encode_queries_kernel(const float *__restrict__ E, const float *__restrict__ R,
                                                             int d, int scorer, const PrefixDev p,
                                                             float *__restrict__ Q, int ldq)
{
    const int b = blockIdx.x, B = p.n_po + p.n_sp;
    float *q = Q + (size_t)b * ldq;
    if (b >= B) {
        for (int k = threadIdx.x; k < ldq; k += blockDim.x) q[k] = 0.f;
        return;
    }
    const RowSrc rs = row_source(p, b);
    const DropDev &de = rs.sp ? p.drop_sp_ent : p.drop_po_ent;
    const DropDev &dr = rs.sp ? p.drop_sp_rel : p.drop_po_rel;
    const float *e = E + rs.ent * d, *r = R + rs.rel * d;
    if (scorer == SC_DISTMULT) {
        for (int k = threadIdx.x; k < d; k += blockDim.x)
            q[k] = (e[k] * drop_mult1(de, rs.pos, k, d)) * (r[k] * drop_mult1(dr, rs.pos, k, d));
    } else {
        const int h = d >> 1;
        for (int k = threadIdx.x; k < h; k += blockDim.x) {
            const float e1 = e[k] * drop_mult1(de, rs.pos, k, d), e2 = e[h + k] * drop_mult1(de, rs.pos, h + k, d);
            const float r1 = r[k] * drop_mult1(dr, rs.pos, k, d), r2 = r[h + k] * drop_mult1(dr, rs.pos, h + k, d);
            if (rs.sp) {           // [s1 r1 - s2 r2 , s2 r1 + s1 r2]
                q[k] = e1 * r1 - e2 * r2;
                q[h + k] = e2 * r1 + e1 * r2;
            } else {               // [o1 r1 + o2 r2 , o2 r1 - o1 r2]
                q[k] = e1 * r1 + e2 * r2;
                q[h + k] = e2 * r1 - e1 * r2;
            }
        }
    }
    for (int k = d + threadIdx.x; k < ldq; k += blockDim.x) q[k] = 0.f;
}

__global__ __launch_bounds__(128) void prefix_backward_kernel(const float *__restrict__ E, const float *__restrict__ R,
                                                              int d, int scorer, const PrefixDev p,
                                                              const float *__restrict__ slab, int nsplit, int Bpad,
                                                              int ldq, float *__restrict__ dE, float *__restrict__ dR)
{
    const int b = blockIdx.x;
    const RowSrc rs = row_source(p, b);
    const DropDev &de = rs.sp ? p.drop_sp_ent : p.drop_po_ent;
    const DropDev &dr = rs.sp ? p.drop_sp_rel : p.drop_po_rel;
    const float *e = E + rs.ent * d, *r = R + rs.rel * d;
    float *ge = dE + rs.ent * d, *gr = dR + rs.rel * d;
    const size_t split_stride = (size_t)Bpad * ldq;
    const float *sl = slab + (size_t)b * ldq;
    if (scorer == SC_DISTMULT) {
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            float dq = 0.f;
            for (int sidx = 0; sidx < nsplit; ++sidx) dq += sl[sidx * split_stride + k];
            const float me = drop_mult1(de, rs.pos, k, d), mr = drop_mult1(dr, rs.pos, k, d);
            const float ev = e[k] * me, rv = r[k] * mr;
            atomicAdd(ge + k, dq * rv * me);
            atomicAdd(gr + k, dq * ev * mr);
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
        const float e1 = e[k] * me1, e2 = e[h + k] * me2, r1 = r[k] * mr1, r2 = r[h + k] * mr2;
        float de1, de2, dr1, dr2;
        if (rs.sp) {
            de1 = q1 * r1 + q2 * r2;  de2 = -q1 * r2 + q2 * r1;
            dr1 = q1 * e1 + q2 * e2;  dr2 = -q1 * e2 + q2 * e1;
        } else {
            de1 = q1 * r1 - q2 * r2;  de2 = q1 * r2 + q2 * r1;
            dr1 = q1 * e1 + q2 * e2;  dr2 = q1 * e2 - q2 * e1;
        }
        atomicAdd(ge + k, de1 * me1);
        atomicAdd(ge + h + k, de2 * me2);
        atomicAdd(gr + k, dr1 * mr1);
        atomicAdd(gr + h + k, dr2 * mr2);
    }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const double *__restrict__ partials, int n,
                                                          double *__restrict__ out)
{
    __shared__ double red[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void kl_count_pos_kernel(const int32_t *__restrict__ pos_row, int nnz,
                                                           float *__restrict__ row_ysum)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nnz) atomicAdd(row_ysum + pos_row[i], 1.0f);
}

__global__ __launch_bounds__(256) void kl_row_lse_kernel(const float *__restrict__ stats, int tiles, int B, int Bpad,
                                                         float *__restrict__ row_lse)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float2 *st = reinterpret_cast<const float2 *>(stats);
    float M = -INFINITY;
    for (int t = 0; t < tiles; ++t) M = fmaxf(M, st[(size_t)t * Bpad + b].x);
    float S = 0.f;
    for (int t = 0; t < tiles; ++t) {
        const float2 v = st[(size_t)t * Bpad + b];
        S += v.y * expf(v.x - M);
    }
    row_lse[b] = M + logf(S);
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
        float *pp = &pv.x, *gg = &gv.x, *ss = &sv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gj = fmaf(wd, pp[j], gg[j]);
            ss[j] = fmaf(gj, gj, ss[j]);
            pp[j] = pp[j] - lr * (gj / (sqrtf(ss[j]) + eps));
        }
        p4[i] = pv;
        s4[i] = sv;
        if (zero_grad) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        const float gj = fmaf(wd, p[i], g[i]);
        sum[i] = fmaf(gj, gj, sum[i]);
        p[i] = p[i] - lr * (gj / (sqrtf(sum[i]) + eps));
        if (zero_grad) g[i] = 0.f;
    }
}

constexpr int RANK_GROUPS = 8;

__global__ __launch_bounds__(256) void ranks_kernel(const float *__restrict__ scores, int64_t ld, int N,
                                                    const int64_t *__restrict__ filt_ptr,
                                                    const int32_t *__restrict__ filt_col,
                                                    const int64_t *__restrict__ row_ptr,
                                                    const int64_t *__restrict__ grp_ptr, const int32_t *__restrict__ ids,
                                                    int64_t *__restrict__ ranks)
{
    __shared__ float tv[RANK_GROUPS];
    __shared__ int cnt[4][2 * RANK_GROUPS];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *row = scores + (size_t)b * ld;
    const int64_t g_lo = row_ptr[b], g_hi = row_ptr[b + 1];
    const int64_t f_lo = filt_ptr[b], f_hi = filt_ptr[b + 1];
    for (int64_t g0 = g_lo; g0 < g_hi; g0 += RANK_GROUPS) {
        const int ng = (int)min((int64_t)RANK_GROUPS, g_hi - g0);
        if (tid < RANK_GROUPS) {
            float t = __builtin_nanf("");                       // unused slots never compare true
            if (tid < ng) {
                t = -INFINITY;
                for (int64_t j = grp_ptr[g0 + tid]; j < grp_ptr[g0 + tid + 1]; ++j) t = fmaxf(t, row[ids[j]]);
            }
            tv[tid] = t;
        }
        __syncthreads();
        float t[RANK_GROUPS];
        int gt[RANK_GROUPS], eq[RANK_GROUPS];
#pragma unroll
        for (int j = 0; j < RANK_GROUPS; ++j) { t[j] = tv[j]; gt[j] = 0; eq[j] = 0; }
        for (int n = tid; n < N; n += 256) {
            const float x = row[n];
#pragma unroll
            for (int j = 0; j < RANK_GROUPS; ++j) { gt[j] += x > t[j]; eq[j] += x == t[j]; }
        }
        // filtered positions count as -1e8 instead of their score (dataset.py:441)
        for (int64_t f = f_lo + tid; f < f_hi; f += 256) {
            const float x = row[filt_col[f]];
#pragma unroll
            for (int j = 0; j < RANK_GROUPS; ++j) {
                gt[j] += (-1e8f > t[j]) - (x > t[j]);
                eq[j] += (-1e8f == t[j]) - (x == t[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < RANK_GROUPS; ++j) {
            const int a = wave_sum(gt[j]), e = wave_sum(eq[j]);
            if ((tid & 63) == 0) { cnt[tid >> 6][2 * j] = a; cnt[tid >> 6][2 * j + 1] = e; }
        }
        __syncthreads();
        if (tid < ng) {
            const int64_t a = (int64_t)cnt[0][2 * tid] + cnt[1][2 * tid] + cnt[2][2 * tid] + cnt[3][2 * tid];
            const int64_t e = (int64_t)cnt[0][2 * tid + 1] + cnt[1][2 * tid + 1] + cnt[2][2 * tid + 1] + cnt[3][2 * tid + 1];
            ranks[g0 + tid] = a + e / 2;
        }
        __syncthreads();
    }
}

// ---- launchers -----------------------------------------------------------------------------------------
hipError_t launch_encode_queries(const float *E, const float *R, int d, int scorer, const PrefixDev &p, float *Q,
                                 int ldq, int Bpad, hipStream_t st)
{
    if (Bpad <= 0) return hipSuccess;
    hipLaunchKernelGGL(encode_queries_kernel, dim3(Bpad), dim3(128), 0, st, E, R, d, scorer, p, Q, ldq);
    return hipGetLastError();
}

hipError_t launch_prefix_backward(const float *E, const float *R, int d, int scorer, const PrefixDev &p,
                                  const float *slab, int nsplit, int Bpad, int ldq, float *dE, float *dR,
                                  hipStream_t st)
{
    const int B = p.n_po + p.n_sp;
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(prefix_backward_kernel, dim3(B), dim3(128), 0, st, E, R, d, scorer, p, slab, nsplit, Bpad, ldq,
                       dE, dR);
    return hipGetLastError();
}

hipError_t launch_loss_reduce(const double *partials, int n, double *loss_out, hipStream_t st)
{
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, st, partials, n, loss_out);
    return hipGetLastError();
}

hipError_t launch_kl_row_stats(const float *stats, int tiles, int B, int Bpad, const int32_t *pos_row, int nnz,
                               float *row_lse, float *row_ysum, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(row_ysum, 0, sizeof(float) * Bpad, st);
    if (e != hipSuccess) return e;
    if (nnz > 0) hipLaunchKernelGGL(kl_count_pos_kernel, dim3((nnz + 255) / 256), dim3(256), 0, st, pos_row, nnz, row_ysum);
    hipLaunchKernelGGL(kl_row_lse_kernel, dim3((B + 255) / 256), dim3(256), 0, st, stats, tiles, B, Bpad, row_lse);
    return hipGetLastError();
}

hipError_t launch_adagrad(float *p, float *g, float *sum, int64_t n, float lr, float wd, float eps, int zero_grad,
                          hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const int64_t n4 = (n + 3) / 4;
    const int blocks = (int)min((int64_t)2048, (n4 + 255) / 256);
    hipLaunchKernelGGL(adagrad_kernel, dim3(blocks), dim3(256), 0, st, p, g, sum, n, lr, wd, eps, zero_grad);
    return hipGetLastError();
}

hipError_t launch_ranks(const float *scores, int64_t ld, int B, int N, const int64_t *filt_ptr,
                        const int32_t *filt_col, const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                        int64_t *ranks, hipStream_t st)
{
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(ranks_kernel, dim3(B), dim3(256), 0, st, scores, ld, N, filt_ptr, filt_col, row_ptr, grp_ptr,
                       ids, ranks);
    return hipGetLastError();
}

}  // namespace okge
