// Device-side helpers shared by the gfx950 kernels of the open-KGE hot path.
// gfx950 only: 64-wide wavefronts, v_mfma_f32_16x16x4_f32 (exact fp32 matrix FMA), 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace okge {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// ---- dropout description as the kernels see it (host fills it from okge_dropout) -----------------
struct DropDev {
    const uint8_t *keep;   // explicit keep mask [rows][d] or nullptr
    float          scale;  // 1/(1-p)   (1 when disabled)
    uint32_t       thr;    // keep <=> 16-bit philox number >= thr (= floor(p * 65536))
    uint32_t       k0, k1; // philox key = seed
    uint32_t       stream, step;
    int32_t        enabled;
    const uint32_t *step_dev;   // device-resident step counter (graph replay) or nullptr
};

__device__ __forceinline__ uint32_t drop_step(const DropDev &dr) { return dr.step_dev ? *dr.step_dev : dr.step; }

// ---- Philox4x32-10 (Salmon et al. 2011), Random123 known-answer vectors in tests/ -------------------
// WIDE: one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi / v_mul_lo pair -- the integer
// multiplies are quarter-rate and set the cost of a round; same numbers.  (The tile kernels use it; hipcc 7.2 fails to
// select instructions for some of the small kernels of okge_misc.hip with it, so it is opt-in.)
template <bool WIDE = false>
__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if (WIDE) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
            hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        } else {
            hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
            hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        }
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// Keep flags (bit e = column 8*o8 + e) of eight consecutive columns of row `row_pos` of a gathered block with
// `d` columns.  One Philox call yields 8 uniform 16-bit numbers: word e>>1, low half for even e, high half
// for odd e; keep <=> u16 >= thr (thr = floor(p * 65536)).  Columns >= d report "drop".
// (`step` = drop_step(dr), fetched once by the caller: with a device-resident counter it is a load, and a load between a
//  kernel's gather and its mask arithmetic makes the arithmetic wait for the gather -- vmcnt retires in order)
template <bool WIDE = false>
__device__ __forceinline__ uint32_t drop_keep8(const DropDev &dr, uint32_t row_pos, int o8, int d, uint32_t step)
{
    uint32_t bits = 0;
    if (dr.keep) {
        const uint8_t *kp = dr.keep + (size_t)row_pos * d + 8 * o8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (8 * o8 + e < d && kp[e]) bits |= 1u << e;
        return bits;
    }
    const uint4 u = philox4x32_10<WIDE>(row_pos, (uint32_t)o8, dr.stream, step, dr.k0, dr.k1);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t u16 = (w[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
        if (u16 >= dr.thr && 8 * o8 + e < d) bits |= 1u << e;
    }
    return bits;
}

__device__ __forceinline__ uint32_t drop_keep8(const DropDev &dr, uint32_t row_pos, int o8, int d)
{
    return drop_keep8(dr, row_pos, o8, d, drop_step(dr));
}

__device__ __forceinline__ float drop_mult1(const DropDev &dr, uint32_t row_pos, int k, int d)
{
    if (!dr.enabled) return 1.f;
    if (dr.keep) return dr.keep[(size_t)row_pos * d + k] ? dr.scale : 0.f;
    const uint4 u = philox4x32_10(row_pos, (uint32_t)(k >> 3), dr.stream, drop_step(dr), dr.k0, dr.k1);
    const int e = k & 7;
    const uint32_t w = (e >> 1) == 0 ? u.x : (e >> 1) == 1 ? u.y : (e >> 1) == 2 ? u.z : u.w;
    return ((w >> (16 * (e & 1))) & 0xFFFFu) >= dr.thr ? dr.scale : 0.f;
}

// v *= (keep ? scale : 0) for the 4 columns selected by bits (nibble already shifted down)
__device__ __forceinline__ void apply_keep4(float4 &v, uint32_t nibble, float scale)
{
    v.x *= (nibble & 1u) ? scale : 0.f;
    v.y *= (nibble & 2u) ? scale : 0.f;
    v.z *= (nibble & 4u) ? scale : 0.f;
    v.w *= (nibble & 8u) ? scale : 0.f;
}

// D(16x16) += A(16x4) * B(4x16), exact fp32.  Lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
// result register i of lane l is D[4*(l>>4)+i][l&15].
__device__ __forceinline__ v4f mfma16(float a, float b, v4f c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Ids live in device memory and are trusted for speed, but never for safety: a row index outside its table is replaced
// by row 0 (the padding row every table has) and counted in a device word the host reads lazily (okge_id_errors).
__device__ __forceinline__ int64_t checked_row(int64_t id, int64_t n_rows, int *err)
{
    if ((uint64_t)id >= (uint64_t)n_rows) {
        if (err) atomicAdd(err, 1);
        return 0;
    }
    return id;
}

// Reductions over the 16 lanes of a DPP row (lanes 16s .. 16s+15) with DPP moves instead of ds_bpermute shuffles: quad
// swaps (xor 1, xor 2), then row_half_mirror and row_mirror -- after the quad steps all four lanes of a quad agree, so
// the mirrors pair every quad with the one it still misses.  All 16 lanes end up with the result.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v)
{
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
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

// ---- Adagrad arithmetic shared by every kernel that applies it (utils/optim.py:139-160 / torch.optim.Adagrad) -----------------
__device__ __forceinline__ bool bits_differ(const float4 &a, const float4 &b)
{
    return ((__float_as_uint(a.x) ^ __float_as_uint(b.x)) | (__float_as_uint(a.y) ^ __float_as_uint(b.y)) |
            (__float_as_uint(a.z) ^ __float_as_uint(b.z)) | (__float_as_uint(a.w) ^ __float_as_uint(b.w))) != 0u;
}

__device__ __forceinline__ void adagrad4(float4 &pv, const float4 &gv, float4 &sv, float lr, float wd, float eps)
{
    float *pp = &pv.x, *ss = &sv.x;
    const float *gg = &gv.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {                        // the arithmetic of adagrad_sweep, element for element
        const float gj = fmaf(wd, pp[j], gg[j]);
        ss[j] = fmaf(gj, gj, ss[j]);
        pp[j] = pp[j] - lr * (gj / (sqrtf(ss[j]) + eps));
    }
}

// ---- decay-only steps on operands known to be ordinary numbers --------------------------------------------------------------
// The compiler's correctly rounded `sqrtf` and `/` spend a third of their instructions on operands that cannot occur here
// (denormal / huge / zero / infinite: `v_div_scale` x2 + `v_div_fixup`, the 2^32 scaling and class test around `v_sqrt_f32`), and
// none of their fix-up arithmetic is packed.  The replay of deferred steps is bound by exactly this arithmetic, so it
// runs the SAME sequences -- `v_sqrt_f32` + the two neighbour tests; `v_rcp_f32` + the Newton step on the reciprocal + two
// residual corrections of the quotient -- on two elements at a time (`v_pk_fma_f32`), without the branches for operands outside
//     sqrt:  2^-96 <= x < inf                    (below: scaled by 2^32 first;  v_cmp_gt 0x0f800000 in the generic code)
//     a / b: a, b != 0, b and 1/b and a/b normal, exponent(a) - exponent(b) < 96, |a| >= 2^-103      (V_DIV_SCALE_F32 leaves
//            both operands as they are and VCC = 0; V_DIV_FMAS_F32 is then a plain fma and V_DIV_FIXUP_F32 returns its operand)
// Inside that range the result is the generic one bit for bit (the same instructions on the same values);
// tests/test_token_pooled.py::test_fast_decay_arithmetic_is_the_generic_one compares the two over twelve orders of magnitude.
typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f2v sqrt_rn_ordinary2(f2v x)
{
    f2v y, dn, up;
    y.x = __builtin_amdgcn_sqrtf(x.x);
    y.y = __builtin_amdgcn_sqrtf(x.y);
    dn.x = __int_as_float(__float_as_int(y.x) - 1);
    dn.y = __int_as_float(__float_as_int(y.y) - 1);
    up.x = __int_as_float(__float_as_int(y.x) + 1);
    up.y = __int_as_float(__float_as_int(y.y) + 1);
    const f2v e_dn = fma2(-dn, y, x), e_up = fma2(-up, y, x);
    f2v r;
    r.x = 0.f >= e_dn.x ? dn.x : y.x;
    r.y = 0.f >= e_dn.y ? dn.y : y.y;
    r.x = 0.f < e_up.x ? up.x : r.x;
    r.y = 0.f < e_up.y ? up.y : r.y;
    return r;
}

__device__ __forceinline__ f2v div_rn_ordinary2(f2v a, f2v b)
{
    f2v r0;
    r0.x = __builtin_amdgcn_rcpf(b.x);
    r0.y = __builtin_amdgcn_rcpf(b.y);
    const f2v one = {1.f, 1.f};
    const f2v r = fma2(fma2(-b, r0, one), r0, r0);
    f2v q = a * r;
    q = fma2(fma2(-b, q, a), r, q);
    q = fma2(fma2(-b, q, a), r, q);
    return q;
}

// may `n` decay-only steps from (p, s) take the sequences above?  Bounds with room for n <= 64 steps of drift (|p| moves by at
// most lr * wd / eps <= 1/16 of itself per step, s grows by g^2 <= 2^20 per step); the parameters' own ranges are checked once
__device__ __forceinline__ bool decay_params_ordinary(float lr, float wd, float eps)
{
    return wd >= 0x1p-40f && wd <= 0x1p-10f && eps >= 0x1p-40f && eps <= 1.f && lr > 0.f && lr <= 16.f && lr * wd <= 0x1p-4f * eps;
}
__device__ __forceinline__ bool decay_operands_ordinary(const float4 &pv, const float4 &sv)
{
    const float *pp = &pv.x, *ss = &sv.x;
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 4; ++j)          // (unsigned compares of the bit patterns: one subtract + one compare per bound pair)
        ok = ok && (__float_as_uint(fabsf(pp[j])) - 0x26800000u) <= (0x49800000u - 0x26800000u)      // 2^-50 <= |p| <= 2^20
                && (__float_as_uint(ss[j]) - 0x0f800000u) <= (0x53800000u - 0x0f800000u);              // 2^-96 <= s <= 2^40
    return ok;
}

__device__ __forceinline__ void decay_step4_ordinary(float4 &pv, float4 &sv, float lr, float wd, float eps)
{
    const f2v wd2 = {wd, wd}, eps2 = {eps, eps}, nlr2 = {-lr, -lr}, zero2 = {0.f, 0.f};
    f2v p01 = {pv.x, pv.y}, p23 = {pv.z, pv.w}, s01 = {sv.x, sv.y}, s23 = {sv.z, sv.w};
    const f2v g01 = fma2(wd2, p01, zero2), g23 = fma2(wd2, p23, zero2);
    s01 = fma2(g01, g01, s01);
    s23 = fma2(g23, g23, s23);
    p01 = fma2(nlr2, div_rn_ordinary2(g01, sqrt_rn_ordinary2(s01) + eps2), p01);
    p23 = fma2(nlr2, div_rn_ordinary2(g23, sqrt_rn_ordinary2(s23) + eps2), p23);
    pv = make_float4(p01.x, p01.y, p23.x, p23.y);
    sv = make_float4(s01.x, s01.y, s23.x, s23.y);
}

// `n` consecutive updates of a row no gradient reached (g = 0: the weight-decay term alone, okge_adagrad_lazy): the same
// arithmetic as n sweeps, one after the other.  If the first step returns the bits it was given on every active lane of the wave
// (warm accumulators: wd * p is below half an ulp of both), all later ones do too and are skipped.  `ordinary` (wave-uniform):
// every active lane's operands allow the short sequences above.
__device__ __forceinline__ void decay_replay4(float4 &pv, float4 &sv, int n, float lr, float wd, float eps)
{
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n <= 0) return;
    const bool ordinary = n <= 64 && decay_params_ordinary(lr, wd, eps) && !__any(!decay_operands_ordinary(pv, sv));
    const float4 p0 = pv, s0 = sv;
    if (ordinary) decay_step4_ordinary(pv, sv, lr, wd, eps);
    else adagrad4(pv, zero, sv, lr, wd, eps);
    if (!__any(bits_differ(pv, p0) || bits_differ(sv, s0))) return;      // (checked once: a row that moves keeps moving)
    if (ordinary) {
        for (int i = 1; i < n; ++i) decay_step4_ordinary(pv, sv, lr, wd, eps);
    } else {
        for (int i = 1; i < n; ++i) adagrad4(pv, zero, sv, lr, wd, eps);
    }
}

// The rows a wave owes work on, one after the other with the NEXT row's loads in flight while the current one is replayed
// (a row is a dependent chain: counters -> row loads -> up to `window` correctly rounded sqrt / div steps -> stores; one row at a
// time left the wave waiting on HBM for most of its life).  Lane-held description of the wave's 64 candidate rows: `mask` (bit j:
// lane j's row needs work), row index, pending decay-only steps, and whether the step with the gradient follows.  Rows of up to
// 256 floats (lane = column quad); longer rows go through the plain column loop.
__device__ __forceinline__ void lazy_rows(uint64_t mask, int64_t row, int n_decay, bool with_grad, float *__restrict__ p,
                                          float *__restrict__ s, float *__restrict__ g, int row_len, int lane, float lr, float wd,
                                          float eps)
{
    const int row4 = row_len >> 2;
    float4 *p4 = reinterpret_cast<float4 *>(p), *s4 = reinterpret_cast<float4 *>(s), *g4 = reinterpret_cast<float4 *>(g);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row4 > 64) {
        while (mask) {
            const int j = __ffsll((unsigned long long)mask) - 1;
            mask &= mask - 1;
            const size_t base = (size_t)__shfl(row, j) * row4;
            const int n = __shfl(n_decay, j);
            const bool wg = __shfl((int)with_grad, j) != 0;
            for (int c = lane; c < row4; c += 64) {
                float4 pv = p4[base + c], sv = s4[base + c], gv = wg ? g4[base + c] : zero;
                const float4 p0 = pv, s0 = sv;
                decay_replay4(pv, sv, n, lr, wd, eps);
                if (wg) adagrad4(pv, gv, sv, lr, wd, eps);
                if (bits_differ(pv, p0)) p4[base + c] = pv;
                if (bits_differ(sv, s0)) s4[base + c] = sv;
                if (wg && ((__float_as_uint(gv.x) | __float_as_uint(gv.y) | __float_as_uint(gv.z) | __float_as_uint(gv.w)) != 0u)) g4[base + c] = zero;
            }
        }
        return;
    }
    const bool act = lane < row4;
    int j = mask ? __ffsll((unsigned long long)mask) - 1 : -1;
    if (j >= 0) mask &= mask - 1;
    size_t base = 0;
    int n = 0;
    bool wg = false;
    float4 pv = zero, sv = zero, gv = zero;
    if (j >= 0) {
        base = (size_t)__shfl(row, j) * row4 + lane;
        n = __shfl(n_decay, j);
        wg = __shfl((int)with_grad, j) != 0;
        if (act) { pv = p4[base]; sv = s4[base]; if (wg) gv = g4[base]; }
    }
    while (j >= 0) {
        const int jn = mask ? __ffsll((unsigned long long)mask) - 1 : -1;
        if (jn >= 0) mask &= mask - 1;
        size_t base_n = 0;
        int n_n = 0;
        bool wg_n = false;
        float4 pn = zero, sn = zero, gn = zero;
        if (jn >= 0) {                                   // the next row's loads leave before this row's arithmetic starts
            base_n = (size_t)__shfl(row, jn) * row4 + lane;
            n_n = __shfl(n_decay, jn);
            wg_n = __shfl((int)with_grad, jn) != 0;
            if (act) { pn = p4[base_n]; sn = s4[base_n]; if (wg_n) gn = g4[base_n]; }
        }
        if (act) {
            const float4 p0 = pv, s0 = sv;
            decay_replay4(pv, sv, n, lr, wd, eps);
            if (wg) adagrad4(pv, gv, sv, lr, wd, eps);
            if (bits_differ(pv, p0)) p4[base] = pv;
            if (bits_differ(sv, s0)) s4[base] = sv;
            if (wg && ((__float_as_uint(gv.x) | __float_as_uint(gv.y) | __float_as_uint(gv.z) | __float_as_uint(gv.w)) != 0u)) g4[base] = zero;
        }
        j = jn; base = base_n; n = n_n; wg = wg_n; pv = pn; sv = sn; gv = gn;
    }
}

// Padded leading dimension (floats) of an LDS tile whose rows hold D16 floats: D16 + 4 = 4 * odd, so
// (a) ds_read_b128 of 16 different rows at one column lands on 16 different 16-byte slots and
// (b) ds_read_b32 of rows r and r+4 at 16 consecutive columns lands on disjoint bank halves.
__host__ __device__ constexpr int lds_ld(int D16) { return D16 + 4; }

}  // namespace okge
