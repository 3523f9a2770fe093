// Token-pooled embedder: UnigramPoolingRelationEmbedder._encode (openkge/model.py:762-786)
//   id -> token ids (|vocab| x L table, right-padded with 0) -> sum / mean / max of the token embedding rows
//      -> [BatchNorm1d, training: statistics of THIS call's rows; evaluation: running statistics] -> rows
// and its backward (batch-norm backward + scatter-add into the token table's dense gradient).  Dropout, the last
// step of _encode, is applied by the consumers (the tile kernels drop rows as they gather them).
// All of it is HBM-bound gather / reduce work: one workgroup per row with 16-byte column accesses for the gathers,
// row-block partial sums + a double-precision finish for the column statistics (no float atomics on statistics).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "okge_device.h"
#include "okge_kernels.h"

namespace okge {

namespace {

constexpr int POOL_SUM = 0, POOL_MEAN = 1, POOL_MAX = 2;
constexpr int STAT_ROWS = 32;          // rows per pooling / partial-sum workgroup
constexpr int POOL_BWD_ROWS = 16, HOT_TOKENS = 32, POOL_MAX_LEN = 64;
constexpr int BN_ROWS = 16;            // rows per workgroup of bn_apply_kernel

// A step of the token-pooled models makes FIVE _encode calls (candidates, po relations, po objects, sp subjects, sp
// relations: trainer.py:75-91) and as many backward passes.  Run one after the other they were 35 small launches per step,
// each a chain of dependent loads on a few hundred workgroups (0.45 ms of a 1.2 ms step at BASELINE configs[4]).  Every
// kernel here therefore takes a BATCH of calls: a workgroup finds its call from the cumulative block counts, so one step is
// three launches forward (pool + partial statistics, finish, apply) and three backward, each filling the chip.
struct PoolBatch {
    PoolCall c[POOL_MAX_CALLS];
    int32_t  n_calls;
    int32_t  cum[POOL_MAX_CALLS + 1];      // cumulative workgroup counts of the kernel being launched
    int     *id_err;
};

__device__ __forceinline__ int locate_call(const PoolBatch &pb, int blk, int &local)
{
    int c = 0;
#pragma unroll
    for (int i = 1; i < POOL_MAX_CALLS; ++i)
        if (i < pb.n_calls && blk >= pb.cum[i]) c = i;
    local = blk - pb.cum[c];
    return c;
}

__device__ __forceinline__ int row_id(const int32_t *ids, int first_id, int i, int n_ids, int *id_err)
{
    return (int)checked_row(ids ? ids[i] : first_id + i, n_ids, id_err);      // row of the token-id table
}

// ---- scatter plan: the token-table gradient without float atomics (round 4) ---------------------------------------------
// dW[token] = sum over the (row, position) pairs that name the token of dx[row].  The atomics version ran at the chip's
// float-atomic rate (~1.3 TB/s of added bytes: 51 us at BASELINE configs[4]) and its sums depended on arrival order.  Now
// every row's dx is STORED once (DX) and every token row is summed by one owner in ascending (row, position) order through an
// inverted index built on the device from the batch's token ids -- four launches, no host work, bit-reproducible:
//   1 (beside bn_partial2)  the pairs of token ids >= SC_HOT ("cold": one pair in eight at configs[4]) are compacted into a
//                           list per workgroup and counted per token (integer atomics without return, full waves)
//   2 (beside bn_finish<2>) list the tokens that have pairs, give each a segment of `pairs` (its base is arbitrary: only the
//                           ORDER INSIDE a segment reaches the sums, and that is fixed below by sorting)
//   3 pool_dx_kernel        dx rows -> DX; the SC_HOT most frequent token ids (the reference's ids are frequency-ranked,
//                           index_mapper.py:95-108; BOS / EOS sit in every row) summed per workgroup, stored as slabs;
//                           workgroups behind those: the cold pairs -> their token's segment (arrival order)
//   4 pool_sum_kernel       per token: sort the segment (<= 64 pairs: by rank in one wave; <= 1024: by rank in one workgroup;
//                           more: bitmap over the pair space), add the rows in that order, one read-modify-write of the dW
//                           row; hot tokens: the workgroups' slabs in block order
// The per-token counters live in a caller-provided STATE buffer that is all-zero between calls (pass 3 counts them back down).
// What the passes cost is dependent-access chains and the per-CU rate of scattered atomics (~1 us per 64-lane instruction),
// not bytes: hence the compaction (atomics with all lanes live), the chains of several items side by side, the counters of a
// workgroup added up in LDS before ONE global atomic.
constexpr int SC_HOT = 32;             // == HOT_TOKENS
constexpr int SC_ROWS = 32;            // rows per workgroup of pool_dx_kernel = rows per hot-token slab
constexpr int SC_HIST_PAIRS = 1024;    // (row, position) pairs per workgroup of pass 1: four per thread, all loads in flight together
constexpr int SC_SHORT = 64;           // segments up to this many pairs are sorted and summed by ONE wave
constexpr int SC_WIN_WORDS = 1024;     // long segments: bitmap window of 32 * 1024 pair indices (4 KB of LDS)
constexpr int SC_LIST = 256;           // long segments / hot slabs: a wave's list of rows to add (8 bitmap words' set bits)
constexpr int SC_SUM_WGS = 1024;       // pass 4: workgroups (4 waves each) striding over the listed tokens (one round of the chip at 4 per CU)
constexpr int SC_LONG_WGS = 64;        // pass 4: workgroups striding over the long segments
constexpr int SC_SORT_MAX = 1024;      // pass 4: long segments up to this many pairs are rank-sorted in LDS

struct ScatterCall {
    const int32_t *tokens, *ids;
    int32_t first_id, n, n_ids, group;
    int32_t row0;                      // first row of the call among its group's rows (DX rows, pair index = row * L + position)
    int32_t blk0;                      // first hot-slab block of the call within its group
    int32_t hblk0;                     // first pass-1 workgroup (= chunk of the cold-pair list) of the call within its group
};
struct ScatterGroup {                  // one token table of the batch (calls with the same dW)
    float    *dW;
    uint8_t  *touched;                 // [vocab] or nullptr
    int32_t  *cnt;                     // STATE [vocab]: pairs per token, zero between calls
    int32_t  *ctr;                     // STATE [16]: 0 tokens with several pairs, 1 pairs allotted, 2 long segments, 3 tokens with one pair (zero between calls); 4, 5, 6: copies of 0, 2, 3 for pass 4
    int2     *cold;                    // [hist_blocks][SC_HIST_PAIRS] (token, pair index) of the cold pairs each pass-1 workgroup met
    int32_t  *cold_n;                  // [hist_blocks] how many
    int2     *single;                  // [cap] (token, its only pair index)
    int32_t  *start;                   // [vocab] segment base of a listed token
    int32_t  *pairs;                   // [n_rows * L]
    int32_t  *seg;                     // [cap][4]: token, base, count, -
    int32_t  *long_list;               // [cap] indices into seg
    float    *hot_slab;                // [blocks][SC_HOT][d]
    uint32_t *hot_seen;                // [blocks]
    float    *DX;                      // [n_rows][d]
    int32_t   vocab, d, L, n_rows, blocks, stamp;
    int32_t   P, cap;                  // n_rows * L pair indices; list capacity min(P, vocab)
    int32_t   hist_blocks;
};
struct ScatterBatch {
    ScatterCall  c[POOL_MAX_CALLS];
    ScatterGroup g[POOL_MAX_CALLS];
    int32_t n_calls, n_groups;
    int32_t hist_rows;                 // rows per workgroup of pass 1 (SC_HIST_PAIRS / L)
    int32_t ablate;                    // diagnostics (OKGE_SC_ABLATE): leave parts out to time the rest; results are then WRONG
    int32_t hist_cum[POOL_MAX_CALLS + 1];
};

__device__ __forceinline__ int sc_tok(int tok, int vocab) { return (unsigned)tok < (unsigned)vocab ? tok : 0; }

// pass 1.  A thread takes four (row, position) pairs and walks their chains of dependent loads (row id -> token id) side by
// side; the cold pairs are compacted through LDS into the workgroup's own chunk of the group's list (no global counter) and
// counted per token by full waves.
__device__ __forceinline__ void scatter_count_block(const ScatterBatch &sb, int blk)
{
    __shared__ int2 found[SC_HIST_PAIRS];
    __shared__ int n_found;
    int c = 0;
#pragma unroll
    for (int i = 1; i < POOL_MAX_CALLS; ++i)
        if (i < sb.n_calls && blk >= sb.hist_cum[i]) c = i;
    const int lb = blk - sb.hist_cum[c];
    const ScatterCall &q = sb.c[c];
    const ScatterGroup &g = sb.g[q.group];
    const int L = g.L, r0 = lb * sb.hist_rows, nr = min(q.n, r0 + sb.hist_rows) - r0;
    if (sb.ablate & 128) return;
    constexpr int PER = SC_HIST_PAIRS / 256;
    if (threadIdx.x == 0) n_found = 0;
    __syncthreads();
    int row[PER], tok[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + 256 * j;
        row[j] = i < nr * L ? row_id(q.ids, q.first_id, r0 + i / L, q.n_ids, nullptr) : -1;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = threadIdx.x + 256 * j;
        tok[j] = row[j] >= 0 ? sc_tok(q.tokens[(size_t)row[j] * L + i % L], g.vocab) : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j)
        if (tok[j] >= SC_HOT) {                          // (below SC_HOT: padding and the hot tokens of pool_dx_kernel)
            const int i = threadIdx.x + 256 * j;
            found[atomicAdd(&n_found, 1)] = make_int2(tok[j], (q.row0 + r0 + i / L) * L + i % L);
        }
    __syncthreads();
    const int n = n_found, chunk = q.hblk0 + lb;
    if (threadIdx.x == 0) g.cold_n[chunk] = n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int2 f = found[i];
        g.cold[(size_t)chunk * SC_HIST_PAIRS + i] = f;
        atomicAdd(&g.cnt[f.x], 1);                       // (result unused: an atomic without return, the wave does not wait)
    }
}

// pass 2: the tokens with pairs get a list entry.  Tokens with ONE pair (four in five) go to a list of their own and need
// no segment: the counter is zeroed here and start[token] = -(list index + 1) tells pass 3 to write the pair index into the
// entry itself.  The others get a segment of `pairs`.
// A workgroup scans SC_ALLOC_TOKENS counters (eight per thread, all loads in flight), adds its four totals up through LDS and
// takes its shares with FOUR atomics (one returning atomic per token and thread-loop iteration made this a 40 us chain).
constexpr int SC_ALLOC_TOKENS = 2048;
__device__ __forceinline__ void scatter_alloc_block(const ScatterBatch &sb, int blk)
{
    __shared__ int wave_tot[4][4], wg_base[4];
    constexpr int PER = SC_ALLOC_TOKENS / 256;
    int gi = 0, lb = blk;
    while (gi + 1 < sb.n_groups && lb >= (sb.g[gi].vocab + SC_ALLOC_TOKENS - 1) / SC_ALLOC_TOKENS) {
        lb -= (sb.g[gi].vocab + SC_ALLOC_TOKENS - 1) / SC_ALLOC_TOKENS;
        ++gi;
    }
    const ScatterGroup &g = sb.g[gi];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (sb.ablate & 256) return;
    int c[PER];
    int mine[4] = {0, 0, 0, 0};                          // tokens with several pairs, their pairs, long segments, tokens with one pair
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int tok = lb * SC_ALLOC_TOKENS + j * 256 + (int)threadIdx.x;
        c[j] = (tok >= SC_HOT && tok < g.vocab) ? g.cnt[tok] : 0;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        mine[0] += c[j] > 1;
        mine[1] += c[j] > 1 ? c[j] : 0;
        mine[2] += c[j] > SC_SHORT;
        mine[3] += c[j] == 1;
    }
    // exclusive prefix of the four counts over the workgroup's threads
    int inc[4] = {mine[0], mine[1], mine[2], mine[3]};
#pragma unroll
    for (int o = 1; o < 64; o <<= 1)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int up = __shfl_up(inc[k], o);
            if (lane >= o) inc[k] += up;
        }
    if (lane == 63)
        for (int k = 0; k < 4; ++k) wave_tot[w][k] = inc[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int k = threadIdx.x;
        const int tot = wave_tot[0][k] + wave_tot[1][k] + wave_tot[2][k] + wave_tot[3][k];
        wg_base[k] = tot ? atomicAdd(&g.ctr[k], tot) : 0;
    }
    __syncthreads();
    int off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        off[k] = wg_base[k] + inc[k] - mine[k];
        for (int ww = 0; ww < w; ++ww) off[k] += wave_tot[ww][k];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (c[j] == 0) continue;
        const int tok = lb * SC_ALLOC_TOKENS + j * 256 + (int)threadIdx.x;
        if (c[j] == 1) {
            const int i = off[3]++;
            if (i >= g.cap) continue;                    // (cannot happen with a zeroed state buffer)
            g.cnt[tok] = 0;
            g.start[tok] = -(i + 1);
            g.single[i].x = tok;
        } else {
            const int i = off[0]++, base = off[1];
            off[1] += c[j];
            if (i >= g.cap) continue;
            g.start[tok] = base;
            if (c[j] > SC_SHORT && off[2] < g.cap) g.long_list[off[2]++] = i;
            *reinterpret_cast<int4 *>(g.seg + 4 * i) = make_int4(tok, base, c[j], 0);
        }
    }
}

// raw[i][k] = pool_t W[tok(i,t)][k] for the 32 rows of the workgroup, and -- training-mode batch-norm -- the block's partial
// column statistics: partial[b][0][k] = mean of the block's rows, partial[b][1][k] = their sum of squared deviations from it
// (merged without cancellation by bn_finish_kernel<0>).  Thread = (column quad cq = tid & 63, row group rg = tid >> 6):
// 16-byte loads of four columns for the group's 8 rows -- 80 gathers per thread at 10 tokens instead of 320 four-byte
// ones with a thread per column (the launch is a chain of gathers on two waves per SIMD: 95 -> 50 us at configs[4]) --,
// the four groups' (mean, M2) merged exactly (Chan) through LDS.  Padded positions (token 0) take part: the table's row 0 is
// an ordinary row whose gradient is suppressed (padding_idx), not a zero row (model.py:660-661 re-initialises the whole
// weight).  Slot sizes that are not a multiple of 4 take the scalar path (thread = column).
__global__ __launch_bounds__(256) void pool_stats_kernel(const PoolBatch pb, int training)
{
    __shared__ int32_t toks[STAT_ROWS][POOL_MAX_LEN];
    __shared__ float inv_len[STAT_ROWS];
    extern __shared__ float grp_stats[];                // [2][4 row groups][d]: per-group mean, M2
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    const int L = q.L, d = q.d, r0 = lb * STAT_ROWS, nr = min(q.n, r0 + STAT_ROWS) - r0;
    for (int i = threadIdx.x; i < nr * L; i += blockDim.x)
        toks[i / L][i % L] = q.tokens[(size_t)row_id(q.ids, q.first_id, r0 + i / L, q.n_ids, (i % L) ? nullptr : pb.id_err) * L + i % L];
    __syncthreads();
    if (training && q.touched)                           // every token row this call reads (row 0 of the padded positions too)
        for (int i = threadIdx.x; i < nr * L; i += blockDim.x) {
            const int tok = toks[i / L][i % L];
            if ((unsigned)tok < (unsigned)q.vocab) q.touched[tok] = (uint8_t)q.touched_stamp;
        }
    if (threadIdx.x < nr) {
        int len = 0;
        for (int t = 0; t < L; ++t) len += toks[threadIdx.x][t] > 0;
        inv_len[threadIdx.x] = q.pool == POOL_MEAN ? 1.f / ((float)len + 1e-12f) : 1.f;     // torch divides: sum / (len + 1e-12)
    }
    __syncthreads();
    const bool stats = training && q.saved;
    const bool copy = !q.bn_weight && q.out != q.raw;
    const bool vec = (d & 3) == 0 && (q.ld & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(q.W) | reinterpret_cast<uintptr_t>(q.raw) | reinterpret_cast<uintptr_t>(q.out)) & 15) == 0;
    if (vec) {
        constexpr int RG = 8;                            // rows per row group
        const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6, rbase = RG * rg;
        const int cnt = max(0, min(nr - rbase, RG));     // this group's live rows
        for (int k = 4 * cq; k < d; k += 256) {
            float4 v[RG];
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                v[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < cnt) {
                    const float ini = q.pool == POOL_MAX ? -INFINITY : 0.f;
                    float4 acc = make_float4(ini, ini, ini, ini);
                    for (int t = 0; t < L; ++t) {
                        const float4 w = *reinterpret_cast<const float4 *>(q.W + (size_t)toks[rbase + r][t] * d + k);
                        if (q.pool == POOL_MAX) { acc.x = fmaxf(acc.x, w.x); acc.y = fmaxf(acc.y, w.y); acc.z = fmaxf(acc.z, w.z); acc.w = fmaxf(acc.w, w.w); }
                        else { acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w; }
                    }
                    if (q.pool == POOL_MEAN) { const float il = inv_len[rbase + r]; acc.x *= il; acc.y *= il; acc.z *= il; acc.w *= il; }
                    v[r] = acc;
                    *reinterpret_cast<float4 *>(q.raw + (size_t)(r0 + rbase + r) * q.ld + k) = acc;
                    if (copy) *reinterpret_cast<float4 *>(q.out + (size_t)(r0 + rbase + r) * q.ld + k) = acc;
                    a.x += acc.x; a.y += acc.y; a.z += acc.z; a.w += acc.w;
                }
            }
            if (stats) {
                const float ic = cnt ? 1.f / (float)cnt : 0.f;
                a.x *= ic; a.y *= ic; a.z *= ic; a.w *= ic;
                float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < RG; ++r)
                    if (r < cnt) {
                        const float dx = v[r].x - a.x, dy = v[r].y - a.y, dz = v[r].z - a.z, dw = v[r].w - a.w;
                        b.x += dx * dx; b.y += dy * dy; b.z += dz * dz; b.w += dw * dw;
                    }
                *reinterpret_cast<float4 *>(grp_stats + (0 * 4 + rg) * d + k) = a;
                *reinterpret_cast<float4 *>(grp_stats + (1 * 4 + rg) * d + k) = b;
            }
        }
        if (!stats) return;
        __syncthreads();
        // merge the four groups' (count, mean, M2) per column: mean = sum n_g mean_g / n, M2 = sum [M2_g + n_g (mean_g - mean)^2]
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            float mean = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) mean += grp_stats[(0 * 4 + g) * d + k] * (float)max(0, min(nr - RG * g, RG));
            mean /= (float)nr;
            float m2 = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float ng = (float)max(0, min(nr - RG * g, RG)), dm = grp_stats[(0 * 4 + g) * d + k] - mean;
                m2 += grp_stats[(1 * 4 + g) * d + k] + ng * dm * dm;
            }
            q.partial[((size_t)lb * 2 + 0) * d + k] = mean;
            q.partial[((size_t)lb * 2 + 1) * d + k] = m2;
        }
        return;
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float v[STAT_ROWS];
        float a = 0.f;
#pragma unroll 4
        for (int r = 0; r < STAT_ROWS; ++r) {
            v[r] = 0.f;
            if (r < nr) {
                float acc = q.pool == POOL_MAX ? -INFINITY : 0.f;
                for (int t = 0; t < L; ++t) {
                    const float w = q.W[(size_t)toks[r][t] * d + k];
                    acc = q.pool == POOL_MAX ? fmaxf(acc, w) : acc + w;
                }
                v[r] = q.pool == POOL_MEAN ? acc * inv_len[r] : acc;
                q.raw[(size_t)(r0 + r) * q.ld + k] = v[r];
                if (copy) q.out[(size_t)(r0 + r) * q.ld + k] = v[r];
                a += v[r];
            }
        }
        if (stats) {
            a /= (float)nr;
            float b = 0.f;
#pragma unroll 4
            for (int r = 0; r < STAT_ROWS; ++r) {
                const float dx = v[r] - a;
                b += r < nr ? dx * dx : 0.f;
            }
            q.partial[((size_t)lb * 2 + 0) * d + k] = a;
            q.partial[((size_t)lb * 2 + 1) * d + k] = b;
        }
    }
}

// Catch-up of the deferred decay-only Adagrad steps (okge_adagrad_lazy, okge_misc.hip): every token row this batch names is
// brought to the current step BEFORE the pooling forward reads it.  Lane = (row, position) pair; a pair whose token row lags
// claims it with an integer atomicMax on the row's step counter (the first claim wins: one owner per row however many pairs
// name it), then the wave replays its claimed rows, lane = column quad (lazy_rows, okge_device.h).  Rows named every step (the
// frequent tokens) never lag.  Token 0 -- the padding id of every short sequence (model.py:579-586): half the pairs at
// configs[4], and a row the backward never stamps -- is claimed by ONE lane per call instead (tens of thousands of atomics on
// one address took 0.4 ms).
constexpr int CATCH_ROWS = 8, CATCH_PAIRS = 16;

__global__ __launch_bounds__(256, 8) void pool_catch_up_kernel(const PoolBatch pb, const int32_t *__restrict__ counters, float lr, float wd,
                                                            float eps, int catch_rows, int catch_pairs)
{
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    if (!q.steps) return;
    const int T = counters[0];
    const int L = q.L, r0 = lb * catch_rows, nr = min(q.n, r0 + catch_rows) - r0, lane = threadIdx.x & 63;
    float *W = const_cast<float *>(q.W);
    // CATCH_PAIRS pairs per wave and turn: many short waves, like adagrad_lazy_kernel
    for (int i0 = (threadIdx.x >> 6) * catch_pairs; i0 < nr * L; i0 += 4 * catch_pairs) {
        const int i = i0 + lane;
        int tok = -1, from = T;
        if (lane < catch_pairs && i < nr * L) tok = q.tokens[(size_t)row_id(q.ids, q.first_id, r0 + i / L, q.n_ids, nullptr) * L + i % L];
        if (tok == 0) tok = -1;                                  // (below: the call's first workgroup looks after row 0)
        bool claim = false;
        if ((unsigned)tok < (unsigned)q.vocab && q.steps[tok] < T) {
            from = atomicMax(&q.steps[tok], T);
            claim = from < T;
        }
        lazy_rows(__ballot(claim), tok, T - from, false, W, q.sumW, q.sumW, q.d, lane, lr, wd, eps);
    }
    if (lb == 0 && threadIdx.x < 64) {                           // row 0, whether or not the first pair names it
        int from = T;
        bool claim = false;
        if (lane == 0 && q.vocab > 0 && q.steps[0] < T) {
            from = atomicMax(&q.steps[0], T);
            claim = from < T;
        }
        lazy_rows(__ballot(claim), 0, T - from, false, W, q.sumW, q.sumW, q.d, lane, lr, wd, eps);
    }
}

// backward partials: partial[b][0][k] = sum_i dy,  partial[b][1][k] = sum_i dy * xhat      (xhat = (x - mean) * rstd)
// Thread = (column quad, group of 8 rows), 16-byte loads, the four groups added through LDS in a fixed order (a thread per
// column walked its 32 rows one 4-byte load after the other: 15 us for 33 MB at configs[4]).  Workgroups behind the
// batch-norm blocks run pass 1 of the scatter plan.
__global__ __launch_bounds__(256) void bn_partial2_kernel(const PoolBatch pb, const ScatterBatch sb, int bn_blocks)
{
    extern __shared__ float grp_sums[];                 // [2][4 row groups][d]
    const int count_blocks = (int)gridDim.x - bn_blocks;   // (they come first in the grid: the longer dependency chain)
    if ((int)blockIdx.x < count_blocks) {
        scatter_count_block(sb, (int)blockIdx.x);
        return;
    }
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, (int)blockIdx.x - count_blocks, lb)];
    const int d = q.d, r0 = lb * STAT_ROWS, r1 = min(q.n, r0 + STAT_ROWS);
    const bool vec = (d & 3) == 0 && (q.ld & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(q.raw) | reinterpret_cast<uintptr_t>(q.dY)) & 15) == 0;
    if (vec) {
        constexpr int RG = 8;
        const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6;
        for (int k = 4 * cq; k < d; k += 256) {
            const float4 m = *reinterpret_cast<const float4 *>(q.saved + k), rs = *reinterpret_cast<const float4 *>(q.saved + d + k);
            float4 dy[RG], x[RG];
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const int i = r0 + RG * rg + r;
                dy[r] = x[r] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < r1) {
                    dy[r] = *reinterpret_cast<const float4 *>(q.dY + (size_t)i * q.ld + k);
                    x[r] = *reinterpret_cast<const float4 *>(q.raw + (size_t)i * q.ld + k);
                }
            }
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                a.x += dy[r].x; a.y += dy[r].y; a.z += dy[r].z; a.w += dy[r].w;
                b.x += dy[r].x * ((x[r].x - m.x) * rs.x); b.y += dy[r].y * ((x[r].y - m.y) * rs.y);
                b.z += dy[r].z * ((x[r].z - m.z) * rs.z); b.w += dy[r].w * ((x[r].w - m.w) * rs.w);
            }
            *reinterpret_cast<float4 *>(grp_sums + (0 * 4 + rg) * d + k) = a;
            *reinterpret_cast<float4 *>(grp_sums + (1 * 4 + rg) * d + k) = b;
        }
        __syncthreads();
        for (int k = threadIdx.x; k < 2 * d; k += blockDim.x) {
            const int which = k >= d, kk = k - which * d;
            const float *gsum = grp_sums + which * 4 * d + kk;
            q.partial[((size_t)lb * 2 + which) * d + kk] = (gsum[0] + gsum[d]) + (gsum[2 * d] + gsum[3 * d]);
        }
        return;
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        float a = 0.f, b = 0.f;
        const float m = q.saved[k], rs = q.saved[d + k];
        for (int i = r0; i < r1; ++i) {
            const float dy = q.dY[(size_t)i * q.ld + k];
            a += dy;
            b += dy * ((q.raw[(size_t)i * q.ld + k] - m) * rs);
        }
        q.partial[((size_t)lb * 2 + 0) * d + k] = a;
        q.partial[((size_t)lb * 2 + 1) * d + k] = b;
    }
}

// finish in double, one workgroup per (call, FIN_COLS columns), FIN_PARTS threads per column each folding its share of the
// row-block partials (4 per column and 64 columns per workgroup made this a 20-workgroup launch of 64-step dependent chains:
// 34 us; 16 x 16: 13 us; 64 threads per column on 4 columns: 7 us at configs[4]):
// STEP 0 -> saved = {mean, rstd, unbiased variance (for the running statistics, applied in call order by bn_apply_kernel's
//           last workgroup)} of all n rows from the per-block (mean, M2) pairs;
// STEP 2 -> this call's (dbias, dweight) into saved[2..3] (added to the parameter gradients in call order by
//           pool_backward_kernel's last workgroup)
constexpr int FIN_COLS = 4, FIN_PARTS = 64;      // (16 x 16 until round 4: 80 workgroups of 16-step dependent chains, 13 us; now 320 of 4 steps)
template <int STEP>
__global__ __launch_bounds__(256) void bn_finish_kernel(const PoolBatch pb, const ScatterBatch sb, int fin_blocks)
{
    __shared__ double red[2][FIN_PARTS][FIN_COLS];
    if ((int)blockIdx.x >= fin_blocks) {                 // (STEP 2 only) pass 2 of the scatter plan
        scatter_alloc_block(sb, (int)blockIdx.x - fin_blocks);
        return;
    }
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    const int n = q.n, d = q.d, blocks = (n + STAT_ROWS - 1) / STAT_ROWS;
    const float *partial = q.partial;
    float *saved = q.saved;
    const int c = threadIdx.x & (FIN_COLS - 1), grp = threadIdx.x / FIN_COLS, k = lb * FIN_COLS + c;
    const bool live = k < d;
    auto fold = [&](int which) {                      // fixed order: the result does not depend on scheduling
        double t = 0.0;
#pragma unroll
        for (int g = 0; g < FIN_PARTS; ++g) t += red[which][g][c];
        return t;
    };
    double a = 0.0, b = 0.0;
    if (STEP == 0) {
        // parallel-variance merge: mean = sum n_b mean_b / n,  M2 = sum [M2_b + n_b (mean_b - mean)^2]
        if (live)
            for (int p = grp; p < blocks; p += FIN_PARTS)
                a += (double)partial[((size_t)p * 2 + 0) * d + k] * (double)min(STAT_ROWS, n - p * STAT_ROWS);
        red[0][grp][c] = a;
        __syncthreads();
        const double mean = fold(0) / n;
        if (live)
            for (int p = grp; p < blocks; p += FIN_PARTS) {
                const double dm = (double)partial[((size_t)p * 2 + 0) * d + k] - mean;
                b += (double)partial[((size_t)p * 2 + 1) * d + k] + dm * dm * (double)min(STAT_ROWS, n - p * STAT_ROWS);
            }
        red[1][grp][c] = b;
        __syncthreads();
        if (!live || grp != 0) return;
        const double var = fold(1) / n;                                  // biased: normalises the batch
        saved[k] = (float)mean;
        saved[d + k] = (float)(1.0 / sqrt(var + (double)q.eps));
        saved[2 * d + k] = (float)(var * n / (n > 1 ? n - 1 : 1));       // running statistics use the unbiased variance
    } else {
        if (live)
            for (int p = grp; p < blocks; p += FIN_PARTS) {
                a += partial[((size_t)p * 2 + 0) * d + k];
                b += partial[((size_t)p * 2 + 1) * d + k];
            }
        red[0][grp][c] = a;
        red[1][grp][c] = b;
        __syncthreads();
        if (!live || grp != 0) return;
        saved[2 * d + k] = (float)fold(0);
        saved[3 * d + k] = (float)fold(1);
    }
}

// y = (x - mean) * rstd * weight + bias     (training: saved = this call's statistics; evaluation: running ones).
// The LAST workgroup of a training launch instead updates the running statistics, call after call in the order given --
// the reference's _encode calls update one module's buffers in sequence (momentum 0.1: the order matters).
__global__ __launch_bounds__(256) void bn_apply_kernel(const PoolBatch pb, int training)
{
    if ((int)blockIdx.x == pb.cum[pb.n_calls]) {
        for (int c = 0; c < pb.n_calls; ++c) {
            const PoolCall &q = pb.c[c];
            if (!q.saved || !q.run_mean) continue;
            for (int k = threadIdx.x; k < q.d; k += blockDim.x) {
                q.run_mean[k] += q.momentum * (q.saved[k] - q.run_mean[k]);
                q.run_var[k] += q.momentum * (q.saved[2 * q.d + k] - q.run_var[k]);
            }
            __syncthreads();             // (calls of one slot share the buffers: finish a call before the next reads them)
        }
        return;
    }
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    // a workgroup takes BN_ROWS rows; thread = column (quad): the per-column factors are loaded once per workgroup
    const int d = q.d, r0 = lb * BN_ROWS, r1 = min(q.n, r0 + BN_ROWS);
    const bool vec = (d & 3) == 0 && (q.ld & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(q.raw) | reinterpret_cast<uintptr_t>(q.out)) & 15) == 0;
    if (vec) {
        for (int k = 4 * threadIdx.x; k < d; k += 4 * blockDim.x) {
            float rs[4], mu[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mu[e] = training ? q.saved[k + e] : q.run_mean[k + e];
                rs[e] = training ? q.saved[d + k + e] : 1.0f / sqrtf(q.run_var[k + e] + q.eps);
            }
            const float4 wt = *reinterpret_cast<const float4 *>(q.bn_weight + k), bias = *reinterpret_cast<const float4 *>(q.bn_bias + k);
            for (int i = r0; i < r1; ++i) {
                const float4 x = *reinterpret_cast<const float4 *>(q.raw + (size_t)i * q.ld + k);
                float4 o;                         // (x - mean) * rstd * weight + bias, the scalar path's order: same bits
                o.x = (x.x - mu[0]) * rs[0] * wt.x + bias.x;
                o.y = (x.y - mu[1]) * rs[1] * wt.y + bias.y;
                o.z = (x.z - mu[2]) * rs[2] * wt.z + bias.z;
                o.w = (x.w - mu[3]) * rs[3] * wt.w + bias.w;
                *reinterpret_cast<float4 *>(q.out + (size_t)i * q.ld + k) = o;
            }
        }
        return;
    }
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        const float mean = training ? q.saved[k] : q.run_mean[k];
        const float rs = training ? q.saved[d + k] : 1.0f / sqrtf(q.run_var[k] + q.eps);
        for (int i = r0; i < r1; ++i)
            q.out[(size_t)i * q.ld + k] = (q.raw[(size_t)i * q.ld + k] - mean) * rs * q.bn_weight[k] + q.bn_bias[k];
    }
}

// dx = weight * rstd * (dy - dbias/n - xhat * dweight/n)   (or dx = dy without batch-norm), then scattered into the
// token table gradient: sum -> every token of the row, mean -> scaled by 1/(len + 1e-12), max -> the first token that
// attains the maximum of its column.  Token 0 (padding_idx) receives nothing.
// One workgroup takes POOL_BWD_ROWS rows, one thread per column.  The reference's token ids are frequency-ranked
// (index_mapper.py:95-108 writes the maps sorted by count; BOS = 2 and EOS = 3 sit in every row), so the first
// HOT_TOKENS ids would take thousands of same-address atomics per step: their rows are accumulated in LDS (a thread
// owns its column, so plain read-modify-write) and flushed with one atomic per touched (token, column) per workgroup.
// The LAST workgroup adds the calls' batch-norm parameter gradients (bn_finish_kernel<2>) up in call order.
__global__ __launch_bounds__(256) void pool_backward_kernel(const PoolBatch pb)
{
    extern __shared__ float hot[];                      // [HOT_TOKENS][d]
    __shared__ uint32_t hot_seen;
    __shared__ int32_t toks[POOL_BWD_ROWS][POOL_MAX_LEN];
    __shared__ float inv_len[POOL_BWD_ROWS];
    if ((int)blockIdx.x == pb.cum[pb.n_calls]) {
        for (int c = 0; c < pb.n_calls; ++c) {
            const PoolCall &q = pb.c[c];
            if (!q.saved) continue;
            for (int k = threadIdx.x; k < q.d; k += blockDim.x) {
                q.d_bias[k] += q.saved[2 * q.d + k];
                q.d_weight[k] += q.saved[3 * q.d + k];
            }
            __syncthreads();
        }
        return;
    }
    int lb;
    const PoolCall &q = pb.c[locate_call(pb, blockIdx.x, lb)];
    const int d = q.d, L = q.L, n = q.n, pool = q.pool;
    const float *saved = q.saved, *X = q.raw, *DY = q.dY, *W = q.W;
    float *dW = q.dW;
    const int64_t ldx = q.ld, lddy = q.ld;
    const int r0 = lb * POOL_BWD_ROWS, nr = min(n, r0 + POOL_BWD_ROWS) - r0;
    for (int i = threadIdx.x; i < HOT_TOKENS * d; i += blockDim.x) hot[i] = 0.f;
    for (int i = threadIdx.x; i < nr * L; i += blockDim.x)
        toks[i / L][i % L] = q.tokens[(size_t)row_id(q.ids, q.first_id, r0 + i / L, q.n_ids, nullptr) * L + i % L];
    if (threadIdx.x == 0) hot_seen = 0;
    __syncthreads();
    if (threadIdx.x < nr) {
        int len = 0;
        for (int t = 0; t < L; ++t) len += toks[threadIdx.x][t] > 0;
        inv_len[threadIdx.x] = pool == POOL_MEAN ? 1.f / ((float)len + 1e-12f) : 1.f;
    }
    __syncthreads();
    const float inv_n = 1.f / (float)n;
    uint32_t seen = 0;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        // all of this column's gradients first (independent loads in flight), then the scatter
        float g[POOL_BWD_ROWS];
#pragma unroll
        for (int r = 0; r < POOL_BWD_ROWS; ++r) g[r] = r < nr ? DY[(size_t)(r0 + r) * lddy + k] : 0.f;
        if (saved) {
            const float m = saved[k], rs = saved[d + k], db = saved[2 * d + k], dw = saved[3 * d + k], wk = q.bn_weight[k];
#pragma unroll
            for (int r = 0; r < POOL_BWD_ROWS; ++r) {
                const float x = r < nr ? X[(size_t)(r0 + r) * ldx + k] : 0.f;
                g[r] = wk * rs * (g[r] - db * inv_n - (x - m) * rs * dw * inv_n);
            }
        }
#pragma unroll
        for (int r = 0; r < POOL_BWD_ROWS; ++r) {
            if (r >= nr) continue;
            if (pool == POOL_MAX) {
                int best = 0;
                float bw = -INFINITY;
                for (int t = 0; t < L; ++t) {
                    const float w = W[(size_t)toks[r][t] * d + k];
                    if (w > bw) { bw = w; best = t; }
                }
                const int tk = toks[r][best];
                if (tk == 0) continue;
                if (q.touched) q.touched[tk] = (uint8_t)q.touched_stamp;
                if (tk < HOT_TOKENS) { hot[tk * d + k] += g[r]; seen |= 1u << tk; }
                else atomicAdd(dW + (size_t)tk * d + k, g[r]);
            } else {
                const float gg = g[r] * inv_len[r];
                for (int t = 0; t < L; ++t) {
                    const int tk = toks[r][t];
                    if (tk == 0) continue;
                    if (k == (int)threadIdx.x && threadIdx.x == 0 && q.touched) q.touched[tk] = (uint8_t)q.touched_stamp;
                    if (tk < HOT_TOKENS) { hot[tk * d + k] += gg; seen |= 1u << tk; }
                    else atomicAdd(dW + (size_t)tk * d + k, gg);
                }
            }
        }
    }
    if (seen) atomicOr(&hot_seen, seen);
    __syncthreads();
    const uint32_t all = hot_seen;
    for (int tk = 1; tk < HOT_TOKENS; ++tk) {
        if (!(all >> tk & 1u)) continue;
        for (int k = threadIdx.x; k < d; k += blockDim.x) {
            const float v = hot[tk * d + k];
            if (v != 0.f) atomicAdd(dW + (size_t)tk * d + k, v);
        }
    }
}

// pass 3 of the scatter plan: dx = batch-norm backward of dy (times 1/(len + 1e-12) for mean pooling) -> DX, the pair indices
// of the row's cold tokens -> their token's segment, the hot tokens' sums over the workgroup's rows -> one slab per workgroup.
// Thread = column: its 32 dx values stay in REGISTERS and the hot tokens' sums are register dot products with a
// [hot token][row] multiplicity table in LDS (read as broadcasts, no dependent chain).  (First version: per (row, position) a
// token-id read from LDS, a branch and a read-modify-write of an LDS accumulator -- 320 dependent LDS round trips per wave,
// 55 us; the atomics kernel it replaced spent its time the same way.)
// The LAST workgroup adds the calls' batch-norm parameter gradients up in call order (as pool_backward_kernel does) and
// hands the counters of passes 1-2 to pass 4.
__global__ __launch_bounds__(256) void pool_dx_kernel(const PoolBatch pb, const ScatterBatch sb, int fill_wgs)
{
    __shared__ float mult[SC_HOT][SC_ROWS];             // how often hot token t stands in row r of this workgroup
    __shared__ int32_t lens[SC_ROWS];
    __shared__ uint32_t hot_seen;
    const int blk = (int)blockIdx.x - fill_wgs;          // the (short) fill workgroups come first in the grid
    if (blk < 0) {
        // the cold pairs go to their token's segment: (token, pair) -> start[token] -> [returning atomic on the token's
        // counter, which pass 1 left at the pair count and which returns to zero here ->] one 4-byte store.  Tokens with a
        // single pair (four in five) need no atomic: the pair goes into the token's list entry.
        if (sb.ablate & 1) return;
        int chunk = (int)blockIdx.x, gi = 0;
        while (gi + 1 < sb.n_groups && chunk >= sb.g[gi].hist_blocks) chunk -= sb.g[gi++].hist_blocks;
        const ScatterGroup &g = sb.g[gi];
        const int n = min(g.cold_n[chunk], SC_HIST_PAIRS);
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int2 f = g.cold[(size_t)chunk * SC_HIST_PAIRS + i];
            const int tok = sc_tok(f.x, g.vocab);
            int slot = g.start[tok];
            // (bounds: a state buffer that was not all-zero must never become an out-of-bounds access)
            if (slot >= 0) {
                slot += atomicSub(&g.cnt[tok], 1) - 1;
                if ((unsigned)slot < (unsigned)g.P) g.pairs[slot] = f.y;
            } else if (-slot - 1 < g.cap) g.single[-slot - 1].y = f.y;
        }
        return;
    }
    if (blk == pb.cum[pb.n_calls]) {
        if (threadIdx.x == 0)
            for (int gi = 0; gi < sb.n_groups; ++gi) {   // passes 1 and 2 are over: hand their counts to pass 4, zero the state
                int32_t *ctr = sb.g[gi].ctr;
                ctr[4] = ctr[0]; ctr[5] = ctr[2]; ctr[6] = ctr[3];
                ctr[0] = ctr[1] = ctr[2] = ctr[3] = 0;
            }
        for (int c = 0; c < pb.n_calls; ++c) {
            const PoolCall &q = pb.c[c];
            if (!q.saved) continue;
            for (int k = threadIdx.x; k < q.d; k += blockDim.x) {
                q.d_bias[k] += q.saved[2 * q.d + k];
                q.d_weight[k] += q.saved[3 * q.d + k];
            }
            __syncthreads();
        }
        return;
    }
    int lb;
    const int ci = locate_call(pb, blk, lb);
    const PoolCall &q = pb.c[ci];
    const ScatterCall &sc = sb.c[ci];
    const ScatterGroup &g = sb.g[sc.group];
    const int d = q.d, L = q.L, n = q.n;
    const float *saved = q.saved, *X = q.raw, *DY = q.dY;
    const int64_t ld = q.ld;
    const int r0 = lb * SC_ROWS, nr = min(n, r0 + SC_ROWS) - r0;
    // (requesting the first column's data before the token phase -- it does not depend on the tokens -- was measured: the 64
    //  values alive across the barrier cost 40 registers and spills, 28.7 -> 32.4 us)
    float gr[SC_ROWS], x[SC_ROWS];
    float m = 0.f, rs = 0.f, db = 0.f, dw = 0.f, wk = 0.f;
    auto load_column = [&](int k) {
        if (saved) { m = saved[k]; rs = saved[d + k]; db = saved[2 * d + k]; dw = saved[3 * d + k]; wk = q.bn_weight[k]; }
#pragma unroll
        for (int r = 0; r < SC_ROWS; ++r) gr[r] = (r < nr && !(sb.ablate & 8)) ? DY[(size_t)(r0 + r) * ld + k] : 0.f;
        if (saved) {
#pragma unroll
            for (int r = 0; r < SC_ROWS; ++r) x[r] = (r < nr && !(sb.ablate & 8)) ? X[(size_t)(r0 + r) * ld + k] : 0.f;
        }
    };
    for (int i = threadIdx.x; i < SC_HOT * SC_ROWS; i += blockDim.x) (&mult[0][0])[i] = 0.f;
    if (threadIdx.x < SC_ROWS) lens[threadIdx.x] = 0;
    if (threadIdx.x == 0) hot_seen = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < nr * L; i += blockDim.x) {
        const int r = i / L;
        const int tok = sc_tok(q.tokens[(size_t)row_id(q.ids, q.first_id, r0 + r, q.n_ids, nullptr) * L + i % L], g.vocab);
        if (tok > 0) atomicAdd(&lens[r], 1);
        if (tok > 0 && tok < SC_HOT) {
            atomicAdd(&mult[tok][r], 1.f);               // (small integers: exact in any order)
            atomicOr(&hot_seen, 1u << tok);
        }
    }
    __syncthreads();
    const float inv_n = 1.f / (float)n;
    const uint32_t seen = hot_seen;
    float *DX = g.DX + (size_t)(sc.row0 + r0) * d;
    float *slab = g.hot_slab + (size_t)(sc.blk0 + lb) * SC_HOT * d;
    for (int k = threadIdx.x; k < d; k += blockDim.x) {
        load_column(k);
        if (saved) {
#pragma unroll
            for (int r = 0; r < SC_ROWS; ++r) gr[r] = wk * rs * (gr[r] - db * inv_n - (x[r] - m) * rs * dw * inv_n);
        }
        if (q.pool == POOL_MEAN) {
#pragma unroll
            for (int r = 0; r < SC_ROWS; ++r) gr[r] *= 1.f / ((float)lens[r] + 1e-12f);
        }
#pragma unroll
        for (int r = 0; r < SC_ROWS; ++r)
            if (r < nr && !(sb.ablate & 4)) DX[(size_t)r * d + k] = gr[r];
        for (int tk = 1; tk < ((sb.ablate & 2) ? 0 : SC_HOT); ++tk) {
            if (!(seen >> tk & 1u)) continue;            // (uniform)
            float acc = 0.f;
#pragma unroll
            for (int r4 = 0; r4 < SC_ROWS; r4 += 4) {
                const float4 mu = *reinterpret_cast<const float4 *>(&mult[tk][r4]);
                acc = fmaf(mu.x, gr[r4], acc); acc = fmaf(mu.y, gr[r4 + 1], acc);
                acc = fmaf(mu.z, gr[r4 + 2], acc); acc = fmaf(mu.w, gr[r4 + 3], acc);
            }
            slab[tk * d + k] = acc;
        }
    }
    if (threadIdx.x == 0) g.hot_seen[sc.blk0 + lb] = seen;
}

__device__ __forceinline__ void add4(float4 &a, const float4 &v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }

// LDS traffic between the lanes of ONE wave (a list written by some lanes, read by all): the wave's DS instructions execute in
// order, the compiler only has to keep them in order
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// acc += rows[list[0]] + rows[list[1]] + ... in list order for the lane's four columns (row i at base + list[i] * stride);
// sixteen row loads in flight
__device__ __forceinline__ void sum_listed_rows(const float *__restrict__ base, size_t stride, const int32_t *list, int n, float4 &acc)
{
    int i = 0;
    for (; i + 16 <= n; i += 16) {
        float4 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const float4 *>(base + (size_t)list[i + j] * stride);
#pragma unroll
        for (int j = 0; j < 16; ++j) add4(acc, v[j]);
    }
    if (i + 8 <= n) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(base + (size_t)list[i + j] * stride);
#pragma unroll
        for (int j = 0; j < 8; ++j) add4(acc, v[j]);
        i += 8;
    }
    if (i + 4 <= n) {
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4 *>(base + (size_t)list[i + j] * stride);
#pragma unroll
        for (int j = 0; j < 4; ++j) add4(acc, v[j]);
        i += 4;
    }
    for (; i < n; ++i) add4(acc, *reinterpret_cast<const float4 *>(base + (size_t)list[i] * stride));
}

// the four waves' partial sums of one destination row, added in a fixed order, then one read-modify-write of the row
__device__ __forceinline__ void combine_and_add(float4 (*part)[64], int w, int lane, bool live, const float4 &acc, float *dst_row)
{
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && live) {
        float4 t0 = part[0][lane], t1 = part[2][lane];
        add4(t0, part[1][lane]); add4(t1, part[3][lane]); add4(t0, t1);
        float4 *dst = reinterpret_cast<float4 *>(dst_row);
        float4 o = *dst;
        add4(o, t0);
        *dst = o;
    }
    __syncthreads();
}

// set bits of `word` (lane < n_words holds word j0 + lane of a bitmap) -> list, ascending; returns how many
template <typename F>
__device__ __forceinline__ int compact_bits(uint32_t word, int lane, int n_words, int32_t *list, F value_of)
{
    const int c = __popc(word);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const int up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    const int total = __shfl(incl, n_words - 1);
    int off = incl - c;
    while (word) {
        const int bit = __ffs(word) - 1;
        word &= word - 1;
        list[off++] = value_of(lane, bit);
    }
    return total;
}

// pass 4 of the scatter plan.  Workgroups [0, n_hot): one hot token of one group each, the slabs added in block order;
// [n_hot, n_hot + n_long): segments of more than SC_SHORT pairs, one workgroup per segment; the rest: one WAVE per segment.
__global__ __launch_bounds__(256, 4) void pool_sum_kernel(const ScatterBatch sb, int n_hot, int n_long)
{
    __shared__ uint32_t bm[SC_WIN_WORDS];
    __shared__ int32_t lists[4][SC_LIST];
    __shared__ float4 part[4][64];
    static_assert(SC_WIN_WORDS >= SC_SORT_MAX && 4 * SC_LIST >= SC_SORT_MAX, "the rank sort of long segments reuses bm / lists");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((int)blockIdx.x < n_hot) {
        if (sb.ablate & 16) return;
        // the slabs of the workgroups of pool_dx_kernel that met the token, in block order: wave w takes a quarter of the
        // blocks, lists the ones whose mask has the token's bit (8 mask words = 256 blocks at a time) and adds their rows
        const ScatterGroup &g = sb.g[blockIdx.x / (SC_HOT - 1)];
        const int tk = 1 + (int)blockIdx.x % (SC_HOT - 1), d = g.d;
        const int per = (((g.blocks + 3) / 4) + 7) & ~7, b_lo = min(g.blocks, w * per), b_hi = min(g.blocks, b_lo + per);
        bool any = false;
        for (int b = threadIdx.x; b < g.blocks; b += blockDim.x) any = any || (g.hot_seen[b] >> tk & 1u);
        if (!__syncthreads_or(any)) return;              // no row of the batch holds this token
        for (int col0 = 0; col0 < d; col0 += 256) {
            const int col = col0 + 4 * lane;
            const bool live = col < d;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int b0 = b_lo; b0 < b_hi; b0 += SC_LIST) {
                // lane l < 8 gathers the token's bit of blocks b0 + 32 l .. b0 + 32 l + 31 into one word
                uint32_t word = 0;
                if (lane < 8) {
#pragma unroll 8
                    for (int j = 0; j < 32; ++j) {
                        const int b = b0 + 32 * lane + j;
                        if (b < b_hi && (g.hot_seen[b] >> tk & 1u)) word |= 1u << j;
                    }
                }
                wave_lds_sync();
                const int total = compact_bits(word, lane, 8, lists[w], [&](int l, int bit) { return b0 + 32 * l + bit; });
                wave_lds_sync();
                if (live && total) sum_listed_rows(g.hot_slab + (size_t)tk * d + col, (size_t)SC_HOT * d, lists[w], total, acc);
            }
            combine_and_add(part, w, lane, live, acc, g.dW + (size_t)tk * d + col);
        }
        if (g.touched && threadIdx.x == 0) g.touched[tk] = (uint8_t)g.stamp;
        return;
    }
    if ((int)blockIdx.x < n_hot + n_long) {
        const int lw = (int)blockIdx.x - n_hot;
        if (sb.ablate & 32) return;
        for (int gi = 0; gi < sb.n_groups; ++gi) {
            const ScatterGroup &g = sb.g[gi];
            const int n_l = min(g.ctr[5], g.cap), d = g.d, L = g.L, P = g.P;
            for (int li = lw; li < n_l; li += n_long) {
                const int si = min((int)g.long_list[li], g.cap - 1), tok = sc_tok(g.seg[4 * si], g.vocab), base = g.seg[4 * si + 1];
                const int n = ((unsigned)base < (unsigned)P) ? min((int)g.seg[4 * si + 2], P - base) : 0;
                if (n <= SC_SORT_MAX) {
                    // rank sort through LDS: bm = the pair indices as they arrived, lists = their rows in ascending pair order;
                    // wave w then adds the w-th quarter of the sorted rows, the four partial sums are added in a fixed order
                    int32_t *arrived = reinterpret_cast<int32_t *>(bm), *sorted = &lists[0][0];
                    for (int j = threadIdx.x; j < n; j += blockDim.x) arrived[j] = g.pairs[base + j];
                    __syncthreads();
                    for (int j = threadIdx.x; j < n; j += blockDim.x) {
                        const int mine = arrived[j];
                        int rank = 0;
                        for (int i = 0; i < n; ++i) rank += arrived[i] < mine;     // (broadcast reads; pair indices are distinct)
                        sorted[min(rank, SC_SORT_MAX - 1)] = (unsigned)mine < (unsigned)P ? mine / L : 0;
                    }
                    __syncthreads();
                    const int per = (n + 3) / 4, lo_ = min(n, w * per), cnt_ = min(n, lo_ + per) - lo_;
                    for (int col0 = 0; col0 < d; col0 += 256) {
                        const int col = col0 + 4 * lane;
                        const bool live = col < d;
                        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (live) sum_listed_rows(g.DX + col, (size_t)d, sorted + lo_, cnt_, acc);
                        combine_and_add(part, w, lane, live, acc, g.dW + (size_t)tok * d + col);
                    }
                } else
                for (int col0 = 0; col0 < d; col0 += 256) {
                    const int col = col0 + 4 * lane;
                    const bool live = col < d;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    for (int win0 = 0; win0 < P; win0 += 32 * SC_WIN_WORDS) {
                        const int words = min(SC_WIN_WORDS, (P - win0 + 31) >> 5);
                        for (int j = threadIdx.x; j < words; j += blockDim.x) bm[j] = 0u;
                        __syncthreads();
                        for (int j = threadIdx.x; j < n; j += blockDim.x) {
                            const int pr = g.pairs[base + j] - win0;
                            if ((unsigned)pr < 32u * (unsigned)words) atomicOr(&bm[pr >> 5], 1u << (pr & 31));
                        }
                        __syncthreads();
                        // wave w walks its quarter of the window's words in ascending order, 8 words at a time
                        const int wq = ((words + 3) / 4 + 7) & ~7, w_lo = min(words, w * wq), w_hi = min(words, w_lo + wq);
                        for (int j0 = w_lo; j0 < w_hi; j0 += 8) {
                            const uint32_t word = (lane < 8 && j0 + lane < w_hi) ? bm[j0 + lane] : 0u;
                            if (!__builtin_amdgcn_ballot_w64(word != 0u)) continue;
                            wave_lds_sync();
                            const int total = compact_bits(word, lane, 8, lists[w], [&](int l, int bit) { return (win0 + 32 * (j0 + l) + bit) / L; });
                            wave_lds_sync();
                            if (live) sum_listed_rows(g.DX + col, (size_t)d, lists[w], total, acc);
                        }
                        __syncthreads();
                    }
                    combine_and_add(part, w, lane, live, acc, g.dW + (size_t)tok * d + col);
                }
                if (g.touched && threadIdx.x == 0) g.touched[tok] = (uint8_t)g.stamp;
            }
        }
        return;
    }
    // short segments: wave-private work, no workgroup barrier below this line.  Per token a chain of dependent loads (segment
    // -> pairs -> rows): the next token's segment is requested before the current one is worked on.
    if (sb.ablate & 64) return;
    const int n_waves = 4 * ((int)gridDim.x - n_hot - n_long), wave = 4 * ((int)blockIdx.x - n_hot - n_long) + w;
    // tokens with one pair (four in five): dW[token] += DX[row], four tokens per wave side by side
    for (int gi = 0; gi < sb.n_groups; ++gi) {
        const ScatterGroup &g = sb.g[gi];
        const int n_s = min(g.ctr[6], g.cap), d = g.d, L = g.L;
        int2 first[4];                                   // (requested before n_s is known: one dependent round trip less)
#pragma unroll
        for (int u = 0; u < 4; ++u) first[u] = g.single[min(4 * wave + u, g.cap - 1)];
        for (int e0 = 4 * wave; e0 < n_s; e0 += 4 * n_waves) {
            int2 ent[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ent[u] = e0 == 4 * wave ? first[u] : g.single[min(e0 + u, g.cap - 1)];
                if (e0 + u >= n_s) ent[u] = make_int2(-1, 0);
            }
            for (int col = 4 * lane; col < d; col += 256) {
                float4 v[4], o[4];
                float4 *dst[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int tok = sc_tok(ent[u].x, g.vocab), row = (unsigned)ent[u].y < (unsigned)g.P ? ent[u].y / L : 0;
                    dst[u] = reinterpret_cast<float4 *>(g.dW + (size_t)tok * d + col);
                    if (ent[u].x >= 0) { v[u] = *reinterpret_cast<const float4 *>(g.DX + (size_t)row * d + col); o[u] = *dst[u]; }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (ent[u].x >= 0) { add4(o[u], v[u]); *dst[u] = o[u]; }
            }
            if (g.touched) {
                const int2 e = lane == 0 ? ent[0] : lane == 1 ? ent[1] : lane == 2 ? ent[2] : ent[3];
                if (lane < 4 && e.x >= 0) g.touched[sc_tok(e.x, g.vocab)] = (uint8_t)g.stamp;
            }
        }
    }
    for (int gi = 0; gi < sb.n_groups; ++gi) {
        const ScatterGroup &g = sb.g[gi];
        const int n_t = min(g.ctr[4], g.cap), d = g.d, L = g.L;
        const int4 *seg4 = reinterpret_cast<const int4 *>(g.seg);
        // (half a grid away from the waves that the single-pair tokens keep busy: all workgroups are resident at once)
        const int rwave = (wave + n_waves / 2) % n_waves;
        int4 nxt = seg4[min(rwave, g.cap - 1)];
        for (int si = rwave; si < n_t; si += n_waves) {
            const int4 sg = nxt;
            const int tok = sc_tok(sg.x, g.vocab), base = sg.y, n = sg.z;
            int pr = INT32_MAX;
            if (n <= SC_SHORT && lane < n && (unsigned)(base + lane) < (unsigned)g.P) pr = g.pairs[base + lane];
            if (si + n_waves < n_t) nxt = seg4[si + n_waves];
            if (n > SC_SHORT || n < 2) continue;
            const int col = 4 * lane;
            float4 *dst = reinterpret_cast<float4 *>(g.dW + (size_t)tok * d + col);
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (col < d) o = *dst;                                              // (needs the token only: in flight beside the rows)
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += __shfl(pr, j) < pr;          // pair indices are distinct: a permutation
            wave_lds_sync();                                                    // (the previous token's list has been read)
            if (lane < n) lists[w][min(rank, SC_SHORT - 1)] = (unsigned)pr < (unsigned)g.P ? pr / L : 0;
            wave_lds_sync();
            if (col < d) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                sum_listed_rows(g.DX + col, (size_t)d, lists[w], n, acc);
                add4(o, acc);
                *dst = o;
            }
            for (int c2 = col + 256; c2 < d; c2 += 256) {                       // slot sizes above 256
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                sum_listed_rows(g.DX + c2, (size_t)d, lists[w], n, acc);
                float4 *dst2 = reinterpret_cast<float4 *>(g.dW + (size_t)tok * d + c2);
                float4 o2 = *dst2;
                add4(o2, acc);
                *dst2 = o2;
            }
            if (g.touched && lane == 0) g.touched[tok] = (uint8_t)g.stamp;
        }
    }
}

template <typename F>
PoolBatch make_batch(const PoolCall *calls, int n_calls, int *id_err, F blocks_of)
{
    PoolBatch pb;
    pb.n_calls = n_calls;
    pb.id_err = id_err;
    pb.cum[0] = 0;
    for (int i = 0; i < POOL_MAX_CALLS; ++i) {
        if (i < n_calls) pb.c[i] = calls[i];
        pb.cum[i + 1] = pb.cum[i] + (i < n_calls ? blocks_of(calls[i]) : 0);
    }
    return pb;
}

}  // namespace

size_t pool_workspace_bytes(int n, int d)
{
    const size_t blocks = (size_t)(n + STAT_ROWS - 1) / STAT_ROWS;
    return (blocks * 2 * (size_t)d * sizeof(float) + 255) / 256 * 256;
}

// forward of a batch of _encode calls (each with n > 0): pooled rows -> raw; batch-norm (calls with `saved`): training-mode
// statistics of THAT call's rows into saved[0..2d), running statistics updated in call order, normalised rows -> out;
// evaluation mode: running statistics.  calls[i].partial must hold pool_workspace_bytes(n_i, d_i).
hipError_t launch_pool_encode_calls(const PoolCall *calls, int n_calls, int training, int *id_err, hipStream_t st)
{
    if (n_calls <= 0) return hipSuccess;
    if (n_calls > POOL_MAX_CALLS) return hipErrorInvalidValue;
    {
        const PoolBatch pb = make_batch(calls, n_calls, id_err, [](const PoolCall &q) { return (q.n + STAT_ROWS - 1) / STAT_ROWS; });
        int dmax = 0;
        for (int i = 0; i < n_calls; ++i) dmax = calls[i].d > dmax ? calls[i].d : dmax;
        hipLaunchKernelGGL(pool_stats_kernel, dim3(pb.cum[n_calls]), dim3(256), sizeof(float) * 2 * 4 * dmax, st, pb, training);
    }
    bool any_bn = false;
    for (int i = 0; i < n_calls; ++i) any_bn = any_bn || calls[i].bn_weight != nullptr;
    if (!any_bn) return hipGetLastError();
    // calls without batch-norm drop out of the two batch-norm launches
    PoolCall bn[POOL_MAX_CALLS];
    int nb = 0;
    for (int i = 0; i < n_calls; ++i)
        if (calls[i].bn_weight) bn[nb++] = calls[i];
    if (training) {
        const PoolBatch pf = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.d + FIN_COLS - 1) / FIN_COLS; });
        static const ScatterBatch no_plan = {};
        hipLaunchKernelGGL(bn_finish_kernel<0>, dim3(pf.cum[nb]), dim3(256), 0, st, pf, no_plan, pf.cum[nb]);
    }
    const PoolBatch pa = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.n + BN_ROWS - 1) / BN_ROWS; });
    hipLaunchKernelGGL(bn_apply_kernel, dim3(pa.cum[nb] + (training ? 1 : 0)), dim3(256), 0, st, pa, training);
    return hipGetLastError();
}

hipError_t launch_pool_catch_up(const PoolCall *calls, int n_calls, const int32_t *counters, float lr, float wd, float eps, int *id_err,
                                hipStream_t st)
{
    if (n_calls <= 0) return hipSuccess;
    if (n_calls > POOL_MAX_CALLS || !counters) return hipErrorInvalidValue;
    static const int rows_env = getenv("OKGE_CATCH_ROWS") ? atoi(getenv("OKGE_CATCH_ROWS")) : CATCH_ROWS;
    static const int pairs_env = getenv("OKGE_CATCH_PAIRS") ? atoi(getenv("OKGE_CATCH_PAIRS")) : CATCH_PAIRS;
    const int catch_rows = rows_env >= 1 && rows_env <= 64 ? rows_env : CATCH_ROWS;
    const int catch_pairs = (pairs_env == 8 || pairs_env == 32 || pairs_env == 64) ? pairs_env : CATCH_PAIRS;
    const PoolBatch pb = make_batch(calls, n_calls, id_err, [catch_rows](const PoolCall &q) { return (q.n + catch_rows - 1) / catch_rows; });
    hipLaunchKernelGGL(pool_catch_up_kernel, dim3(pb.cum[n_calls]), dim3(256), 0, st, pb, counters, lr, wd, eps, catch_rows, catch_pairs);
    return hipGetLastError();
}

namespace {

inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

// groups the calls by token table and lays out the state / scratch regions of the scatter plan; false: not applicable
// (max pooling scatters per column; the 16-byte paths need slot sizes that are a multiple of 4)
struct ScatterLayout {
    int n_groups;
    int group_of_call[POOL_MAX_CALLS], row0[POOL_MAX_CALLS];
    int rep[POOL_MAX_CALLS], n_rows[POOL_MAX_CALLS];       // per group: a representative call, rows over its calls
    size_t state_off[POOL_MAX_CALLS], state_bytes, scratch_bytes;
};
bool scatter_layout(const PoolCall *calls, int n_calls, ScatterLayout &lo)
{
    lo.n_groups = 0;
    lo.state_bytes = lo.scratch_bytes = 0;
    for (int i = 0; i < n_calls; ++i) {
        const PoolCall &q = calls[i];
        if (q.pool == POOL_MAX || (q.d & 3) || !q.dW || (reinterpret_cast<uintptr_t>(q.dW) & 15) || q.vocab <= 0) return false;
        int g = -1;
        for (int j = 0; j < lo.n_groups; ++j)
            if (calls[lo.rep[j]].dW == q.dW) g = j;
        if (g < 0) {
            g = lo.n_groups++;
            lo.rep[g] = i;
            lo.n_rows[g] = 0;
        } else {
            const PoolCall &r = calls[lo.rep[g]];
            if (r.tokens != q.tokens || r.L != q.L || r.vocab != q.vocab || r.d != q.d || r.touched != q.touched) return false;
        }
        lo.group_of_call[i] = g;
        lo.row0[i] = lo.n_rows[g];
        lo.n_rows[g] += q.n;
        if ((int64_t)lo.n_rows[g] * q.L > (int64_t)1 << 30) return false;
    }
    for (int g = 0; g < lo.n_groups; ++g) {
        const PoolCall &r = calls[lo.rep[g]];
        lo.state_off[g] = lo.state_bytes;
        lo.state_bytes += al256(sizeof(int32_t) * (16 + (size_t)r.vocab));
        const size_t P = (size_t)lo.n_rows[g] * r.L, cap = std::min<size_t>(P, (size_t)r.vocab);
        lo.scratch_bytes += al256(sizeof(int32_t) * (size_t)r.vocab) + al256(sizeof(int32_t) * P) + al256(16 * cap) + al256(4 * cap) + al256(8 * cap);
    }
    return true;
}
// the regions whose size depends on the number of workgroups of pool_dx_kernel (rows per workgroup: `rows_wg`)
size_t scatter_block_scratch(const PoolCall *calls, int n_calls, const ScatterLayout &lo, int rows_wg)
{
    size_t bytes = 0;
    int hist_L = 1;
    for (int i = 0; i < n_calls; ++i) hist_L = std::max(hist_L, calls[i].L);
    for (int g = 0; g < lo.n_groups; ++g) {
        size_t blocks = 0;
        for (int i = 0; i < n_calls; ++i)
            if (lo.group_of_call[i] == g) blocks += (size_t)(calls[i].n + rows_wg - 1) / rows_wg;
        const PoolCall &r = calls[lo.rep[g]];
        bytes += al256(sizeof(float) * blocks * SC_HOT * r.d) + al256(sizeof(uint32_t) * blocks) + al256(sizeof(float) * (size_t)lo.n_rows[g] * r.d);
        // the cold-pair chunks of pass 1: one per workgroup, SC_HIST_PAIRS / L rows each (at most one more workgroup per call than rows / that)
        size_t hist_blocks = 0;
        const int hist_rows = std::max(1, SC_HIST_PAIRS / hist_L);
        for (int i = 0; i < n_calls; ++i)
            if (lo.group_of_call[i] == g) hist_blocks += (size_t)(calls[i].n + hist_rows - 1) / hist_rows;
        bytes += al256(8 * hist_blocks * SC_HIST_PAIRS) + al256(4 * hist_blocks);
    }
    return bytes;
}
}  // namespace

size_t pool_scatter_state_bytes(const PoolCall *calls, int n_calls)
{
    ScatterLayout lo;
    return scatter_layout(calls, n_calls, lo) ? lo.state_bytes : 0;
}

size_t pool_scatter_workspace_bytes(const PoolCall *calls, int n_calls)
{
    ScatterLayout lo;
    return scatter_layout(calls, n_calls, lo) ? lo.scratch_bytes + scatter_block_scratch(calls, n_calls, lo, SC_ROWS) : 0;
}

hipError_t launch_pool_backward_calls(const PoolCall *calls, int n_calls, int *id_err, hipStream_t st, void *state,
                                      size_t state_bytes, void *scratch, size_t scratch_bytes)
{
    if (n_calls <= 0) return hipSuccess;
    if (n_calls > POOL_MAX_CALLS) return hipErrorInvalidValue;
    PoolCall bn[POOL_MAX_CALLS];
    int nb = 0, dmax = 0;
    for (int i = 0; i < n_calls; ++i) {
        if (calls[i].saved) bn[nb++] = calls[i];
        dmax = calls[i].d > dmax ? calls[i].d : dmax;
    }
    static ScatterBatch no_plan;                         // n_calls = n_groups = 0
    ScatterLayout lo;
    constexpr int rows_wg = SC_ROWS;
    const bool plan = state && scratch && scatter_layout(calls, n_calls, lo) && state_bytes >= lo.state_bytes &&
                      scratch_bytes >= lo.scratch_bytes + scatter_block_scratch(calls, n_calls, lo, rows_wg);
    if (state && !plan) return hipErrorInvalidValue;     // the caller asked for the plan: say so instead of silently using atomics
    if (!plan) {
        if (nb) {
            const PoolBatch p1 = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.n + STAT_ROWS - 1) / STAT_ROWS; });
            hipLaunchKernelGGL(bn_partial2_kernel, dim3(p1.cum[nb]), dim3(256), sizeof(float) * 2 * 4 * dmax, st, p1, no_plan, p1.cum[nb]);
            const PoolBatch p2 = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.d + FIN_COLS - 1) / FIN_COLS; });
            hipLaunchKernelGGL(bn_finish_kernel<2>, dim3(p2.cum[nb]), dim3(256), 0, st, p2, no_plan, p2.cum[nb]);
        }
        const PoolBatch p3 = make_batch(calls, n_calls, id_err, [](const PoolCall &q) { return (q.n + POOL_BWD_ROWS - 1) / POOL_BWD_ROWS; });
        static LdsOptIn lds_a;
        const size_t shmem = sizeof(float) * HOT_TOKENS * dmax;
        if (hipError_t e = ensure_dynamic_lds(lds_a, reinterpret_cast<const void *>(&pool_backward_kernel), shmem)) return e;
        hipLaunchKernelGGL(pool_backward_kernel, dim3(p3.cum[n_calls] + (nb ? 1 : 0)), dim3(256), shmem, st, p3);
        return hipGetLastError();
    }
    // ---- scatter plan
    ScatterBatch sb;
    std::memset(&sb, 0, sizeof(sb));
    sb.n_calls = n_calls;
    sb.n_groups = lo.n_groups;
    static const int ablate = getenv("OKGE_SC_ABLATE") ? atoi(getenv("OKGE_SC_ABLATE")) : 0;
    sb.ablate = ablate;
    char *sp = static_cast<char *>(scratch);
    auto take = [&](size_t bytes) { char *p = sp; sp += al256(bytes); return p; };
    for (int g = 0; g < lo.n_groups; ++g) {
        const PoolCall &r = calls[lo.rep[g]];
        ScatterGroup &G = sb.g[g];
        const size_t P = (size_t)lo.n_rows[g] * r.L, cap = std::min<size_t>(P, (size_t)r.vocab);
        size_t blocks = 0;
        for (int i = 0; i < n_calls; ++i)
            if (lo.group_of_call[i] == g) blocks += (size_t)(calls[i].n + rows_wg - 1) / rows_wg;
        G.dW = r.dW; G.touched = r.touched; G.stamp = r.touched_stamp;
        G.ctr = reinterpret_cast<int32_t *>(static_cast<char *>(state) + lo.state_off[g]);
        G.cnt = G.ctr + 16;
        G.start = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)r.vocab));
        G.pairs = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * P));
        G.seg = reinterpret_cast<int32_t *>(take(16 * cap));
        G.long_list = reinterpret_cast<int32_t *>(take(4 * cap));
        G.single = reinterpret_cast<int2 *>(take(8 * cap));
        G.hot_slab = reinterpret_cast<float *>(take(sizeof(float) * blocks * SC_HOT * r.d));
        G.hot_seen = reinterpret_cast<uint32_t *>(take(sizeof(uint32_t) * blocks));
        G.DX = reinterpret_cast<float *>(take(sizeof(float) * (size_t)lo.n_rows[g] * r.d));
        G.vocab = r.vocab; G.d = r.d; G.L = r.L; G.n_rows = lo.n_rows[g]; G.blocks = (int)blocks; G.P = (int)P; G.cap = (int)cap;
    }
    int blk_of_group[POOL_MAX_CALLS] = {0};
    int Lmax = 1;
    for (int i = 0; i < n_calls; ++i) Lmax = std::max(Lmax, calls[i].L);
    sb.hist_rows = std::max(1, SC_HIST_PAIRS / Lmax);
    sb.hist_cum[0] = 0;
    int hblk_of_group[POOL_MAX_CALLS] = {0};
    for (int i = 0; i < POOL_MAX_CALLS; ++i) {
        if (i < n_calls) {
            ScatterCall &c = sb.c[i];
            const int g = lo.group_of_call[i];
            c.tokens = calls[i].tokens; c.ids = calls[i].ids; c.first_id = calls[i].first_id; c.n = calls[i].n; c.n_ids = calls[i].n_ids;
            c.group = g; c.row0 = lo.row0[i]; c.blk0 = blk_of_group[g]; c.hblk0 = hblk_of_group[g];
            blk_of_group[g] += (calls[i].n + rows_wg - 1) / rows_wg;
            hblk_of_group[g] += (calls[i].n + sb.hist_rows - 1) / sb.hist_rows;
        }
        sb.hist_cum[i + 1] = sb.hist_cum[i] + (i < n_calls ? (calls[i].n + sb.hist_rows - 1) / sb.hist_rows : 0);
    }
    int fill_wgs = 0;
    for (int g = 0; g < lo.n_groups; ++g) {
        ScatterGroup &G = sb.g[g];
        G.hist_blocks = hblk_of_group[g];
        G.cold = reinterpret_cast<int2 *>(take(8 * (size_t)G.hist_blocks * SC_HIST_PAIRS));
        G.cold_n = reinterpret_cast<int32_t *>(take(4 * (size_t)G.hist_blocks));
        fill_wgs += G.hist_blocks;
    }
    {
        const PoolBatch p1 = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.n + STAT_ROWS - 1) / STAT_ROWS; });
        hipLaunchKernelGGL(bn_partial2_kernel, dim3(p1.cum[nb] + sb.hist_cum[n_calls]), dim3(256), sizeof(float) * 2 * 4 * dmax, st, p1, sb, p1.cum[nb]);
        const PoolBatch p2 = make_batch(bn, nb, id_err, [](const PoolCall &q) { return (q.d + FIN_COLS - 1) / FIN_COLS; });
        int alloc_wgs = 0;
        for (int g = 0; g < lo.n_groups; ++g) alloc_wgs += (sb.g[g].vocab + SC_ALLOC_TOKENS - 1) / SC_ALLOC_TOKENS;
        hipLaunchKernelGGL(bn_finish_kernel<2>, dim3(p2.cum[nb] + alloc_wgs), dim3(256), 0, st, p2, sb, p2.cum[nb]);
    }
    {
        const PoolBatch p3 = make_batch(calls, n_calls, id_err, [](const PoolCall &q) { return (q.n + SC_ROWS - 1) / SC_ROWS; });
        hipLaunchKernelGGL(pool_dx_kernel, dim3(p3.cum[n_calls] + 1 + fill_wgs), dim3(256), 0, st, p3, sb, fill_wgs);
    }
    const int n_hot = lo.n_groups * (SC_HOT - 1);
    hipLaunchKernelGGL(pool_sum_kernel, dim3(n_hot + SC_LONG_WGS + SC_SUM_WGS), dim3(256), 0, st, sb, n_hot, SC_LONG_WGS);
    return hipGetLastError();
}

}  // namespace okge
