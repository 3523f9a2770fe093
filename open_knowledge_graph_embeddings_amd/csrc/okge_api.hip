// C ABI (include/okge.h) over the gfx950 kernels.  Host-side only: argument checking, workspace carving,
// launch geometry, HIP-event timing.  No torch, no CPU fallback: every entry point either enqueues HIP
// kernels or returns an error.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/okge.h"
#include "okge_kernels.h"

using namespace okge;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return OKGE_ERR_HIP;
}

// ---- per-kernel HIP-event timing ----------------------------------------------------------------------
struct TimedLaunch {
    const char *name;
    hipEvent_t  start, stop;
};
std::mutex               g_tmu;
bool                     g_timing = false;
std::vector<TimedLaunch> g_launches;
std::vector<hipEvent_t>  g_event_pool;

hipEvent_t get_event()
{
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

constexpr int OKGE_MAX_DEVICES_SIDE = 64;

struct ScopedTimer {
    hipStream_t st;
    bool        on;
    TimedLaunch t;
    ScopedTimer(const char *name, hipStream_t s) : st(s), on(g_timing)
    {
        if (!on) return;
        std::lock_guard<std::mutex> lk(g_tmu);
        t.name = name;
        t.start = get_event();
        t.stop = get_event();
        (void)hipEventRecord(t.start, st);
    }
    ~ScopedTimer()
    {
        if (!on) return;
        (void)hipEventRecord(t.stop, st);
        std::lock_guard<std::mutex> lk(g_tmu);
        g_launches.push_back(t);
    }
};

// ---- a side stream per (host thread, device): the partial-slab reduction of a training call BESIDE the dQ kernel -----------
// An experiment, OFF unless OKGE_REDUCE_OVERLAP=1.  dc_reduce* is HBM-bound (17 us / 87 MB at configs[2], 9.5 us at configs[4]) and
// the dQ kernel that follows it is MFMA-bound and needs nothing the reduction writes (it reads G^T and the masked candidate rows;
// the reduction turns the tile kernel's partial slabs into dE rows): fork after the tile launch, join before whatever touches dE
// or the slabs next (the prefix backward, the next range's tile launch); event record / wait pairs, captured as plain dependencies
// when the caller's stream is capturing.  Correct (the whole GPU suite passes with it on) and SLOWER: next to the dQ kernel's
// resident workgroups the reduction stretches 17 -> 66 us and the dQ kernel 54 -> 59 us: configs[2] 0.257 -> 0.276 ms, configs[4]
// 0.728 -> 0.743, the 8-rank FB shape 0.154 -> 0.164 (profiles/round4_ablation.md section 7) -- like the two side-stream sweeps before.
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t  fork = nullptr, join = nullptr;
    bool        ok = false;
};
static SideStream *side_stream()
{
    static const bool enabled = std::getenv("OKGE_REDUCE_OVERLAP") && std::atoi(std::getenv("OKGE_REDUCE_OVERLAP")) == 1;
    if (!enabled) return nullptr;
    thread_local SideStream side[OKGE_MAX_DEVICES_SIDE];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OKGE_MAX_DEVICES_SIDE) return nullptr;
    SideStream &x = side[dev];
    if (!x.s) {
        x.ok = hipStreamCreateWithFlags(&x.s, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&x.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&x.join, hipEventDisableTiming) == hipSuccess;
        if (!x.s) x.s = reinterpret_cast<hipStream_t>(-1);       // (creation failed: never tried again, the reduction stays in line)
    }
    return x.ok ? &x : nullptr;
}

// ---- device word counting out-of-range ids (checked_row in the kernels); read and cleared by okge_id_errors ------------
// One word PER DEVICE (keyed by hipGetDevice() at the call): a process that drives several devices must never hand a kernel
// on device b a pointer that lives on device a.
constexpr int OKGE_MAX_DEVICES = 64;
static std::mutex g_dev_mu;
static int *g_id_err[OKGE_MAX_DEVICES] = {};
int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= OKGE_MAX_DEVICES) return -1;
    return dev;
}
int *id_err_ptr()
{
    const int dev = current_device();
    if (dev < 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (!g_id_err[dev]) {
        if (hipMalloc(reinterpret_cast<void **>(&g_id_err[dev]), sizeof(int)) != hipSuccess) { g_id_err[dev] = nullptr; return nullptr; }
        (void)hipMemset(g_id_err[dev], 0, sizeof(int));
    }
    return g_id_err[dev];
}

// ---- helpers ---------------------------------------------------------------------------------------------
DropDev to_dev(const okge_dropout &d)
{
    DropDev r;
    std::memset(&r, 0, sizeof(r));
    r.scale = 1.f;
    if (d.p > 0.f) {
        r.enabled = 1;
        r.keep = d.keep;
        r.scale = 1.0f / (1.0f - d.p);
        const double t = (double)d.p * 65536.0;
        r.thr = t <= 0 ? 0u : (t >= 65535.0 ? 65535u : (uint32_t)t);
        r.k0 = (uint32_t)d.seed;
        r.k1 = (uint32_t)(d.seed >> 32);
        r.stream = d.stream;
        r.step = d.step;
        r.step_dev = d.step_dev;
    }
    return r;
}

PrefixDev to_dev(const okge_prefix_batch &b, const okge_tables *t, const okge_shard *sh = nullptr)
{
    PrefixDev p;
    p.po_rel = b.po_rel; p.po_obj = b.po_obj; p.sp_subj = b.sp_subj; p.sp_rel = b.sp_rel;
    p.n_po = b.n_po; p.n_sp = b.n_sp;
    p.ent_lo = sh ? sh->ent_lo : 0;
    p.ent_hi = sh ? sh->ent_hi : t->n_ent;          // unsharded: the range IS the table, an id outside it is an error
    p.whole_table = sh ? 0 : 1;
    p.n_rel = t->n_rel;
    p.id_err = id_err_ptr();
    p.drop_po_ent = to_dev(b.drop_po_ent); p.drop_po_rel = to_dev(b.drop_po_rel);
    p.drop_sp_ent = to_dev(b.drop_sp_ent); p.drop_sp_rel = to_dev(b.drop_sp_rel);
    return p;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Geometry {
    int    B, N, d, Bpad, D16, KB, LDK, ldq, ldg, tiles, b_split, b_per_block, nsplit;
    int    tile_w, ktiles;      // fused train kernel: tile width (32 or 64 candidates) and tile count
    int    sk_wgs, sk_chunks;   // > 0: the train tile kernel is launched stream-K shaped over sk_wgs workgroups (okge_train64k.hip)
    // > 0: the LAST `tail_tiles` candidate tiles of the call (what is left over after whole rounds of one workgroup per CU)
    // are launched apart with the batch rows split over tail_split workgroups each, so the closing round is short
    int    tail_tiles, tail_split, tail_b_per_block;
    bool   dq8;
    // training sweeps the candidates in RANGES of range_n (a multiple of 64) so that the one (B, N)-shaped
    // intermediate, G^T, and everything sized like it (masked rows Cm, KL statistics) is O(B x range_n):
    // 41 GB -> 1 GB at |E| = 2.5 M, B = 4096.  A range is one tile-kernel launch + one dq launch (slabs accumulate).
    int    range_n, n_ranges, range_tiles, range_ktiles;
    size_t off_Q, off_tptr, off_stats, off_lse, off_ysum, off_run, off_loss, off_GT, off_Cm, off_slab, off_dcs;
    size_t score_bytes, lse_bytes, total;
};

int env_int(const char *name, int dflt)
{
    const char *v = std::getenv(name);
    return v && *v ? std::atoi(v) : dflt;
}

// compute units of the current device (stream-K launches: one workgroup per CU); 256 where no device answers
int cu_count()
{
    static std::mutex mu;
    static int cus[OKGE_MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return 256;
    std::lock_guard<std::mutex> lk(mu);
    if (!cus[dev]) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// Tail split of a launch of `tiles` equal tiles, one workgroup per CU (slot sizes up to 256): the `tail` tiles left over after
// whole rounds would cost a whole round's time; launched apart with the rows split `split` ways they cost 1 / split of it.
// Worth it when the rows per workgroup drop by at least two 64-row blocks.  (cfg4 shard: 4883 tiles = 19 rounds + 19 tiles.)
struct TailSplit { int tiles, split, b_per_block; };
TailSplit tail_split(int tiles, int Bpad, int slots)
{
    TailSplit t = {0, 0, Bpad};
    const int bblks = Bpad / BC, tail = tiles % slots;
    if (tiles <= slots || tail == 0 || env_int("OKGE_TAIL_SPLIT", 1) == 0) return t;
    const int c = std::min(slots / tail, bblks);
    if (c < 2) return t;
    const int per = (bblks + c - 1) / c;
    if (bblks - per < 2) return t;
    t.tiles = tail;
    t.b_per_block = per * BC;
    t.split = (Bpad + t.b_per_block - 1) / t.b_per_block;
    return t;
}

bool make_geometry(int B, int N, int d, Geometry &g)
{
    if (B <= 0 || N <= 0 || d <= 0) return false;
    g.B = B; g.N = N; g.d = d;
    g.Bpad = (B + BC - 1) / BC * BC;
    // slot size padded to a width the tile kernels are instantiated for (zero columns are free of error,
    // not of time: d in (128, 208] runs as 208, etc.)
    const int kb = (d + 15) / 16;
    g.KB = kb <= 4 ? 4 : kb <= 8 ? 8 : kb <= 13 ? 13 : kb <= 16 ? 16 : 32;
    g.D16 = 16 * g.KB;
    g.LDK = lds_ld(g.D16);
    g.ldq = g.D16;
    g.tiles = (N + NT - 1) / NT;
    const int bblks = g.Bpad / BC;
    // candidate ranges: G^T of one range stays under OKGE_GT_MBYTES (default 1024 MiB)
    {
        const int64_t budget = (int64_t)std::max(1, env_int("OKGE_GT_MBYTES", 1024)) << 20;
        int64_t rt = budget / ((int64_t)g.Bpad * NT * (int64_t)sizeof(float));     // 64-candidate tiles per range
        rt = std::max<int64_t>(1, std::min<int64_t>(rt, g.tiles));
        g.range_tiles = (int)rt;
        g.n_ranges = (g.tiles + g.range_tiles - 1) / g.range_tiles;
        const int cus = cu_count();
        if (g.n_ranges > 1 && g.KB <= 16 && rt >= cus) {
            // slot sizes up to 256 run one workgroup per CU and tile: ranges of whole rounds (a multiple of the CU count: 256
            // tiles on an MI355X), so that only the LAST range ends in a partial round -- the one the tail split below shortens
            g.range_tiles = (int)(rt / cus * cus);
        } else {
            g.range_tiles = (g.tiles + g.n_ranges - 1) / g.n_ranges;              // even out the ranges
        }
        g.n_ranges = (g.tiles + g.range_tiles - 1) / g.range_tiles;
        g.range_n = g.range_tiles * NT;
        g.range_ktiles = 2 * g.range_tiles;
    }
    g.ldg = g.range_tiles * NT;
    // fused train kernel: one 8-wave workgroup per CU on 64-candidate tiles (slot sizes up to 256), or the 32-candidate
    // cut for larger slots (an even tile count so that every 64-candidate chunk dq_kernel reads has been written)
    // (slot sizes above 256: fused_tile64k_kernel, candidate tile in registers; OKGE_TILE_W=32 selects the round-1/2 cut)
    g.tile_w = env_int("OKGE_TILE_W", 64) == 32 ? 32 : 64;
    g.ktiles = g.tiles * (NT / g.tile_w);
    const int slots = (g.tile_w == 64 || g.KB > 16) ? cu_count() : 2 * cu_count();   // one workgroup per CU, except the 32-wide cut at d <= 256
    // fill the CUs: if there are few candidate tiles, split the batch rows across blockIdx.y.  Cost of a split into c:
    // (rounds of workgroups over the slots) x (row blocks per workgroup + 1) -- the "+ 1" is a workgroup's fixed cost, the
    // candidate gather and the gradient write-back, about one 64-row block (profiles/round2_ablation.md §1) -- plus the c
    // partial-gradient slabs dc_reduce adds up (an eighth of a block each).  Measured at cfg3 (313 tiles of 32, d = 512):
    // c = 1 / 2 / 3 / 4 / 8 -> 0.299 / 0.288 / 0.294 / 0.303 / 0.364 ms per step; the model orders them 18 / 15 / 16 / 15 / 20.
    int bs = 1;
    {
        long best = -1;
        for (int c = 1; c <= bblks; ++c) {
            const long rounds = ((long)g.ktiles * c + slots - 1) / slots, per = (bblks + c - 1) / c;
            const long cost = 8 * rounds * (per + 1) + c;
            if (best < 0 || cost < best) { best = cost; bs = c; }
        }
    }
    bs = std::max(1, env_int("OKGE_B_SPLIT", bs));
    bs = std::min(bs, bblks);
    g.b_per_block = (bblks + bs - 1) / bs * BC;
    g.b_split = (g.Bpad + g.b_per_block - 1) / g.b_per_block;
    if (g.n_ranges > 1) { g.b_split = 1; g.b_per_block = g.Bpad; }   // many tiles: no batch split
    // slot sizes above 256 (fused_tile64k_kernel), one launch: (tile, 32-row chunk) units in equal runs over one workgroup
    // per CU instead of a (tiles, batch splits) grid -- see the kernel's header for the measurements behind it
    g.sk_wgs = g.sk_chunks = 0;
    if (g.KB == 32 && g.tile_w == 64 && g.n_ranges == 1 && env_int("OKGE_STREAMK", 1) != 0 && env_int("OKGE_B_SPLIT", 0) == 0) {
        g.sk_chunks = (B + 31) / 32;
        const int64_t units = (int64_t)g.tiles * g.sk_chunks;
        g.sk_wgs = (int)std::min<int64_t>(std::min(cu_count(), 511), units);
        g.b_split = 1;
        g.b_per_block = g.Bpad;
    }
    // tail split (slot sizes up to 256, more tiles than CUs): equal tiles in rounds of one per CU leave a closing round with
    // `tail` < 256 workgroups that still takes a whole round's time (cfg4 shard: 4883 tiles = 19 rounds + 19 tiles -> 20 rounds,
    // 4.6 % of the kernel idle).  The tail tiles are launched apart with the rows split floor(256 / tail) ways; their partial
    // candidate gradients go through the batch-split slabs + dc_reduce.  Worth it when the rows per workgroup drop by >= 2 blocks.
    g.tail_tiles = g.tail_split = 0;
    g.tail_b_per_block = g.Bpad;
    if (g.KB <= 16 && g.tile_w == 64 && g.b_split == 1) {
        const TailSplit ts = tail_split(g.tiles - (g.n_ranges - 1) * g.range_tiles, g.Bpad, slots);
        if (ts.split > 1) { g.tail_tiles = ts.tiles; g.tail_split = ts.split; g.tail_b_per_block = ts.b_per_block; }
    }
    // dQ kernel: (batch block, candidate range) workgroups: 8-wave workgroups, one per CU (d <= 256), else 4-wave, two per CU
    // (slot sizes above 256: dq8k_kernel, 32-candidate chunks double-buffered; OKGE_DQ8K=0 selects dq_kernel<32>)
    g.dq8 = (g.KB <= 16 && env_int("OKGE_DQ8", 1) != 0) || (g.KB == 32 && env_int("OKGE_DQ8K", 1) != 0);
    int ns = std::max(1, (g.dq8 ? cu_count() : 2 * cu_count()) / bblks);
    if (ns >= 8) ns = ns / 8 * 8;   // workgroups of one candidate range then share an XCD (blockIdx % 8)
    ns = env_int("OKGE_DQ_SPLIT", ns);
    ns = std::max(1, std::min(ns, g.range_tiles));
    g.nsplit = ns;
    size_t off = 0;
    // [score-only part] the query block
    g.off_Q = off;     off += align_up((size_t)g.Bpad * g.ldq * sizeof(float), 256);
    g.score_bytes = off;
    // [row log-sum-exp part] per-range (max, sum-exp) tile statistics + the running per-row state
    g.off_tptr = off;  off += align_up((size_t)(2 * g.tiles + 1) * sizeof(int32_t), 256);
    g.off_stats = off; off += align_up((size_t)2 * g.range_ktiles * g.Bpad * 2 * sizeof(float), 256);
    g.off_lse = off;   off += align_up((size_t)g.Bpad * sizeof(float), 256);
    g.off_ysum = off;  off += align_up((size_t)g.Bpad * sizeof(float), 256);
    g.off_run = off;   off += align_up((size_t)g.Bpad * 2 * sizeof(float), 256);
    g.lse_bytes = off;
    // [training part]
    g.off_loss = off;  off += align_up((size_t)std::max(2 * g.tiles * g.b_split + g.tail_tiles * g.tail_split, g.sk_wgs) * sizeof(double), 256);
    g.off_GT = off;    off += align_up((size_t)g.Bpad * g.ldg * sizeof(float), 256);
    g.off_Cm = off;    off += align_up((size_t)g.range_tiles * NT * g.D16 * sizeof(float), 256);
    g.off_slab = off;  off += align_up((size_t)g.nsplit * g.Bpad * g.ldq * sizeof(float), 256);
    g.off_dcs = off;   off += g.sk_wgs > 0 ? align_up((size_t)2 * g.sk_wgs * NT * g.D16 * sizeof(float), 256)
                                     : g.b_split > 1 ? align_up((size_t)g.b_split * g.tiles * NT * g.D16 * sizeof(float), 256)
                                     : g.tail_split > 1 ? align_up((size_t)g.tail_split * g.tail_tiles * NT * g.D16 * sizeof(float), 256) : 0;
    g.total = off;
    return true;
}

int check_common(const okge_tables *t, const okge_prefix_batch *b, const okge_candidates *c)
{
    if (!t || !b || !c) return fail(OKGE_ERR_INVALID, "null descriptor");
    if (!t->E || !t->R) return fail(OKGE_ERR_INVALID, "null embedding table");
    if (t->d <= 0 || t->n_ent <= 0 || t->n_rel <= 0) return fail(OKGE_ERR_INVALID, "bad table shape");
    if (t->scorer != OKGE_COMPLEX && t->scorer != OKGE_DISTMULT) return fail(OKGE_ERR_INVALID, "unknown scorer");
    if (t->scorer == OKGE_COMPLEX && (t->d & 1)) return fail(OKGE_ERR_INVALID, "ComplEx needs an even slot size");
    if (t->d > 512) return fail(OKGE_ERR_UNSUPPORTED, "slot sizes above 512 are not supported by the tile kernels");
    if (b->n_po < 0 || b->n_sp < 0 || b->n_po + b->n_sp <= 0) return fail(OKGE_ERR_INVALID, "empty batch");
    if (b->n_po > 0 && (!b->po_rel || !b->po_obj)) return fail(OKGE_ERR_INVALID, "null po ids");
    if (b->n_sp > 0 && (!b->sp_subj || !b->sp_rel)) return fail(OKGE_ERR_INVALID, "null sp ids");
    if (c->n <= 0) return fail(OKGE_ERR_INVALID, "no candidates");
    const int64_t cand_rows = c->table ? c->table_rows : t->n_ent;
    if (!c->ids && (c->first_id < 0 || (int64_t)c->first_id + c->n > cand_rows))
        return fail(OKGE_ERR_INVALID, "candidate range outside the candidate table");
    return OKGE_OK;
}

void fill_fused_common(FusedArgs &a, const Geometry &g, const okge_tables *t, const okge_candidates *c, char *ws)
{
    std::memset(&a, 0, sizeof(a));
    a.E = c->table ? c->table : t->E;       // table the candidate rows are gathered from
    a.n_table_rows = c->table ? c->table_rows : t->n_ent;
    a.id_err = id_err_ptr();
    a.cand_ids = c->ids;
    a.cand_first = c->first_id;
    a.Q = ws ? reinterpret_cast<const float *>(ws + g.off_Q) : nullptr;
    a.drop_c = to_dev(c->drop);
    a.d = g.d; a.KB = g.KB; a.LDK = g.LDK; a.N = g.N; a.B = g.B; a.Bpad = g.Bpad; a.ldq = g.ldq; a.ldg = g.ldg;
    a.b_per_block = g.b_per_block;
}


// Arguments of the tiles from `t0` on of a launch: every per-tile pointer and the candidate window move with them
FusedArgs tile_window(const FusedArgs &base, const Geometry &g, int t0)
{
    FusedArgs a = base;
    a.N = base.N - t0 * NT;
    a.cand_first += t0 * NT;
    if (a.cand_ids) a.cand_ids += t0 * NT;
    a.cand_col0 += t0 * NT;
    if (a.tile_ptr) a.tile_ptr += t0;
    if (a.loss_partial) a.loss_partial += t0;
    if (a.G) a.G += (size_t)t0 * (g.Bpad / BC) * (BC * NT);
    if (a.Cm) a.Cm += (size_t)t0 * NT * g.D16;
    if (a.X) a.X += (size_t)t0 * NT;
    if (a.stats) a.stats += (size_t)t0 * g.Bpad * 2;
    if (a.rk_slab) a.rk_slab += (size_t)t0 * a.rk_ngroups;
    return a;
}

// the score sweep (MODE_SCORE / _STATS / _COUNT) of `tiles` 64-candidate tiles over all B rows: the 64 x 64 kernel up to slot size
// 256 (rows are independent: the tail tiles simply run as (tile, row split) workgroups); above, the register-tile kernel in a
// stream-K launch (no partial outputs to add up), or the 32 x 32 cut it replaced
hipError_t launch_score_sweep(const Geometry &g, const FusedArgs &a0, int tiles, hipStream_t st, int mode = MODE_SCORE)
{
    if (g.KB <= 16) {
        const TailSplit ts = a0.b_per_block >= g.Bpad ? tail_split(tiles, g.Bpad, cu_count()) : TailSplit{0, 0, g.Bpad};
        if (ts.split < 2) return launch_fused(mode, a0, tiles, 1, st);
        FusedArgs am = a0;
        am.N = (tiles - ts.tiles) * NT;
        if (hipError_t e = launch_fused(mode, am, tiles - ts.tiles, 1, st); e != hipSuccess) return e;
        FusedArgs at = tile_window(a0, g, tiles - ts.tiles);
        at.b_per_block = ts.b_per_block;
        return launch_fused(mode, at, ts.tiles, ts.split, st);
    }
    if (g.tile_w == 32) return launch_fused32(mode, a0, 2 * tiles, 1, st);
    FusedArgs a = a0;
    a.sk_tiles = tiles;
    const int64_t units = (int64_t)tiles * ((a.B + 31) / 32);
    return launch_fused64k(mode, a, (int)std::min<int64_t>(std::min(cu_count(), 511), units), 1, st);
}

// Arguments of candidate range r (geometry: ranges of g.range_n candidates): local candidate 0 of the launch is
// candidate r * range_n of the call.  Positives / dropout keep their global columns through cand_col0.
FusedArgs range_args(const FusedArgs &base, const Geometry &g, int r, int &tiles_r)
{
    FusedArgs a = base;
    const int n_lo = r * g.range_n;
    a.N = std::min(g.range_n, g.N - n_lo);
    a.cand_first += n_lo;
    if (a.cand_ids) a.cand_ids += n_lo;
    a.cand_col0 += n_lo;
    if (a.tile_ptr) a.tile_ptr += n_lo / g.tile_w;
    if (a.loss_partial) a.loss_partial += (size_t)(n_lo / g.tile_w) * g.b_split;
    tiles_r = (a.N + NT - 1) / NT;
    return a;
}

// per-row log-sum-exp over all candidates of the call (KL loss): one score-statistics pass per candidate range, merged
// into a running (max, sum-exp) per row; the (B, N) scores are never materialised
int lse_pass(const Geometry &g, const FusedArgs &base, char *ws, float *row_lse, hipStream_t st,
             const int32_t *count_pos_row = nullptr, int count_nnz = 0, float *count_ysum = nullptr)
{
    FusedArgs b = base;
    b.stats = reinterpret_cast<float *>(ws + g.off_stats);
    b.b_per_block = g.Bpad;
    b.tile_ptr = nullptr;
    b.loss_partial = nullptr;
    for (int r = 0; r < g.n_ranges; ++r) {
        int tiles_r;
        const FusedArgs s = range_args(b, g, r, tiles_r);
        hipError_t e;
        {
            ScopedTimer tm("fused_tile_stats", st);
            e = launch_score_sweep(g, s, tiles_r, st, MODE_STATS);      // (slot sizes above 256: four 16-candidate blocks per tile)
            if (e != hipSuccess) return fail_hip(e, "fused_tile_kernel<stats>");
        }
        ScopedTimer tm("kl_row_lse", st);
        // the 64x64 cut emits one (max, sum-exp) per 64-candidate tile, the 32x32 cut one per 16-candidate block
        e = launch_kl_row_lse(s.stats, g.KB <= 16 ? tiles_r : 4 * tiles_r, g.B, g.Bpad,
                              reinterpret_cast<float *>(ws + g.off_run), r == 0, r == g.n_ranges - 1, row_lse, st,
                              r == 0 ? count_pos_row : nullptr, count_nnz, r == 0 ? count_ysum : nullptr);   // + the rows' label mass
        if (e != hipSuccess) return fail_hip(e, "kl_row_lse");
    }
    return OKGE_OK;
}

}  // namespace

int okge::report_error(int code, const std::string &msg) { return fail(code, msg); }

extern "C" {

int okge_abi_version(void) { return OKGE_ABI_VERSION; }

const char *okge_last_error(void) { return g_err.c_str(); }

size_t okge_train_workspace_bytes(int32_t B, int32_t N, int32_t d)
{
    Geometry g;
    if (!make_geometry(B, N, d, g)) return 0;
    return g.total;
}

size_t okge_score_workspace_bytes(int32_t B, int32_t d)
{
    Geometry g;
    if (!make_geometry(B, 1, d, g)) return 0;
    return g.score_bytes;
}

size_t okge_lse_workspace_bytes(int32_t B, int32_t N, int32_t d)
{
    Geometry g;
    if (!make_geometry(B, N, d, g)) return 0;
    return g.lse_bytes;
}

int okge_score_prefixes(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                        float *scores, int64_t ld_scores, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_common(t, batch, cand)) return rc;
    if (!scores || ld_scores < cand->n) return fail(OKGE_ERR_INVALID, "bad scores buffer");
    Geometry g;
    make_geometry(batch->n_po + batch->n_sp, cand->n, t->d, g);
    const size_t need = g.score_bytes;   // only the query block
    if (!workspace || workspace_bytes < need) return fail(OKGE_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    const PrefixDev p = to_dev(*batch, t);
    {
        ScopedTimer tm("encode_queries", st);
        hipError_t e = launch_encode_queries(t->E, t->R, t->d, t->scorer, p, reinterpret_cast<float *>(ws + g.off_Q),
                                             g.ldq, g.Bpad, nullptr, nullptr, 0, nullptr, g.tiles, NT, 0, st);
        if (e != hipSuccess) return fail_hip(e, "encode_queries");
    }
    FusedArgs a;
    fill_fused_common(a, g, t, cand, ws);
    a.X = scores;
    a.ldx = ld_scores;
    a.x_vec_ok = (ld_scores % 4 == 0) && (reinterpret_cast<uintptr_t>(scores) % 16 == 0);
    a.b_per_block = g.Bpad;     // rows are independent in score mode, but one pass per tile keeps C resident
    {
        ScopedTimer tm("fused_tile_score", st);
        hipError_t e = launch_score_sweep(g, a, g.tiles, st);
        if (e != hipSuccess) return fail_hip(e, "fused_tile_kernel<score>");
    }
    return OKGE_OK;
}

// Shared core of the single-device step and of the sharded okge_train_tiles.
//   q_ext   : folded queries computed elsewhere (sharded path) or nullptr (encode them here from `batch`)
//   dq_out  : if set, the dQ slabs are reduced into it and the prefix backward is left to the caller
static int train_core(const okge_tables *t, const okge_shard *sh, const okge_prefix_batch *batch, const float *q_ext,
                      int64_t ldq_ext, int32_t B, const okge_candidates *cand, const okge_positives *pos,
                      int32_t loss_kind, float label_smoothing, double normalizer, int32_t n_cand_global, int32_t flags,
                      double *loss_out, float *dE, float *dR, float *dq_out, float *scores, int64_t ld_scores,
                      const float *row_lse_ext, void *workspace, size_t workspace_bytes, void *stream,
                      const okge_adagrad *opt = nullptr)
{
    if (!pos || pos->nnz < 0 || (pos->nnz > 0 && (!pos->col || !pos->row)))
        return fail(OKGE_ERR_INVALID, "bad positives");
    if (loss_kind != OKGE_LOSS_BCE && loss_kind != OKGE_LOSS_KL) return fail(OKGE_ERR_INVALID, "unknown loss");
    const bool loss_only = (flags & OKGE_TRAIN_LOSS_ONLY) != 0;
    if (!loss_out || (!loss_only && !dE)) return fail(OKGE_ERR_INVALID, "null output");
    if (cand->table) return fail(OKGE_ERR_INVALID, "training needs candidates from the entity table");
    if (!(normalizer > 0)) return fail(OKGE_ERR_INVALID, "normalizer must be positive");
    if (scores && ld_scores < cand->n) return fail(OKGE_ERR_INVALID, "bad scores buffer");
    Geometry g;
    make_geometry(B, cand->n, t->d, g);
    if (!workspace || workspace_bytes < g.total) return fail(OKGE_ERR_WORKSPACE, "workspace too small");
    if (q_ext && ldq_ext != g.ldq) return fail(OKGE_ERR_INVALID, "query block leading dimension must be okge_query_ld(d)");
    const bool clear_grads = (flags & OKGE_TRAIN_CLEAR_GRADS) != 0 && !loss_only;
    if (clear_grads) {
        if (cand->ids || sh || !dR) return fail(OKGE_ERR_INVALID, "OKGE_TRAIN_CLEAR_GRADS needs a contiguous candidate range on an unsharded table");
        flags |= OKGE_TRAIN_GRADS_ZERO;          // the candidate rows are stored, everything else is cleared below
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    const int cand_col0 = sh ? sh->cand_col0 : 0;
    hipError_t e;
    {
        // folded queries (unless supplied) + per-tile offsets into the column-sorted positives
        ScopedTimer tm("encode_queries", st);
        PrefixDev p;
        std::memset(&p, 0, sizeof(p));
        if (!q_ext) p = to_dev(*batch, t, sh);
        ClearSpec clr = {};
        // the KL loss' label mass per row is cleared HERE, by workgroups of this launch -- counted in lse_pass (own row
        // log-sum-exp) or by kl_count_pos (sharded: the log-sum-exp arrives from the exchange).  Not with hipMemsetAsync: inside a
        // captured HIP graph the memset node was not re-executed / ordered on replay (ROCm 7.2), the mass doubled every replay
        const bool kl_own_lse = loss_kind == OKGE_LOSS_KL;
        if (kl_own_lse) { clr.p[3] = reinterpret_cast<float *>(ws + g.off_ysum); clr.n[3] = g.Bpad; }
        if (clear_grads) {                        // all of dR; the rows of dE in front of and behind the candidate range
            const int64_t d64 = t->d, hi = (int64_t)cand->first_id + cand->n;
            clr.p[0] = dR;                      clr.n[0] = (int64_t)t->n_rel * d64;
            clr.p[1] = dE;                      clr.n[1] = (int64_t)cand->first_id * d64;
            clr.p[2] = dE + hi * d64;           clr.n[2] = ((int64_t)t->n_ent - hi) * d64;
        }
        if (opt) clr.prefix_flags = opt->prefix_flags;
        e = launch_encode_queries(t->E, t->R, t->d, t->scorer, p, reinterpret_cast<float *>(ws + g.off_Q), g.ldq,
                                  q_ext ? 0 : g.Bpad, nullptr, pos->col, pos->nnz,
                                  reinterpret_cast<int32_t *>(ws + g.off_tptr), g.ktiles, g.tile_w, cand_col0, st,
                                  (clear_grads || kl_own_lse || opt) ? &clr : nullptr);
        if (e != hipSuccess) return fail_hip(e, "encode_queries");
    }
    FusedArgs a;
    fill_fused_common(a, g, t, cand, ws);
    if (q_ext) a.Q = q_ext;
    a.cand_col0 = cand_col0;
    a.pos_col = pos->col; a.pos_row = pos->row; a.nnz = pos->nnz;
    a.tile_ptr = reinterpret_cast<const int32_t *>(ws + g.off_tptr);
    a.grads_zero = (flags & OKGE_TRAIN_GRADS_ZERO) ? 1 : 0;
    // an explicit id list may name an entity twice (precompute_batch_shared_inputs takes any list): its rows are then
    // accumulated with atomics unless the caller vouches for uniqueness (the collator's lists are unique)
    a.cand_exclusive = (!cand->ids || (flags & OKGE_TRAIN_UNIQUE_CANDIDATES)) ? 1 : 0;
    a.loss_only = loss_only ? 1 : 0;
    a.G = reinterpret_cast<float *>(ws + g.off_GT);
    a.Cm = reinterpret_cast<float *>(ws + g.off_Cm);
    a.dE = dE;
    a.dC_slab = (g.b_split > 1 || g.sk_wgs > 0 || g.tail_split > 1) ? reinterpret_cast<float *>(ws + g.off_dcs) : nullptr;
    a.loss_partial = reinterpret_cast<double *>(ws + g.off_loss);
    a.loss_kind = loss_kind;
    a.inv_norm = (float)(1.0 / normalizer);
    // label smoothing (trainer.py:103-105): y <- (y + 1/N) * (1 - eps), bce branch only; N = all candidates
    a.y_pos = 1.f; a.y_neg = 0.f;
    if (loss_kind == OKGE_LOSS_BCE && label_smoothing > 0.f) {
        const float invn = 1.0f / (float)(n_cand_global > 0 ? n_cand_global : cand->n);
        a.y_pos = (1.0f + invn) * (1.0f - label_smoothing);
        a.y_neg = (0.0f + invn) * (1.0f - label_smoothing);
    }
    if (scores) {
        FusedArgs s = a;
        s.X = scores; s.ldx = ld_scores;
        s.x_vec_ok = (ld_scores % 4 == 0) && (reinterpret_cast<uintptr_t>(scores) % 16 == 0);
        s.b_per_block = g.Bpad;
        ScopedTimer tm("fused_tile_score", st);
        e = launch_score_sweep(g, s, g.tiles, st);
        if (e != hipSuccess) return fail_hip(e, "fused_tile_kernel<score>");
    }
    if (loss_kind == OKGE_LOSS_KL) {
        if (sh && !row_lse_ext)
            return fail(OKGE_ERR_INVALID, "sharded KL loss: pass the all-shard row log-sum-exp (okge_row_logsumexp + exchange)");
        if (!row_lse_ext) {
            // (the label mass per row, trainer.py:99-101, is counted by extra workgroups of the pass' first kl_row_lse launch;
            //  its buffer was cleared by the encode launch above)
            if (int rc = lse_pass(g, a, ws, reinterpret_cast<float *>(ws + g.off_lse), st, pos->row, pos->nnz,
                                  reinterpret_cast<float *>(ws + g.off_ysum)))
                return rc;
        } else {
            ScopedTimer tm("kl_count_pos", st);   // label mass per row (trainer.py:99-101: y is not normalised)
            e = launch_kl_count_pos(pos->row, pos->nnz, g.Bpad, reinterpret_cast<float *>(ws + g.off_ysum), st);
            if (e != hipSuccess) return fail_hip(e, "kl_count_pos");
        }
        a.row_lse = row_lse_ext ? row_lse_ext : reinterpret_cast<const float *>(ws + g.off_lse);
        a.row_ysum = reinterpret_cast<const float *>(ws + g.off_ysum);
    }
#ifdef OKGE_STAMPS
    if (const char *sp = std::getenv("OKGE_STAMPS_PTR")) a.stamps_dbg = reinterpret_cast<unsigned long long *>(std::strtoull(sp, nullptr, 0));
#endif
    DqArgs q;
    std::memset(&q, 0, sizeof(q));
    q.G = a.G; q.Cm = a.Cm;
    q.slab = reinterpret_cast<float *>(ws + g.off_slab);
    q.d = g.d; q.KB = g.KB; q.LDK = g.LDK; q.Bpad = g.Bpad; q.ldq = g.ldq; q.ldg = g.ldg;
    q.nsplit = g.nsplit;
    q.waves8 = g.dq8 ? 1 : 0;
    const int mode = loss_kind == OKGE_LOSS_KL ? MODE_TRAIN_KL : MODE_TRAIN_BCE;
    // (tail split: the tail launch's partials of row split y > 0 follow the per-tile ones -- index y * tail + x from the tail's
    //  first tile is contiguous with them)
    const int n_loss_partials = g.sk_wgs > 0 ? g.sk_wgs : g.ktiles * g.b_split + g.tail_tiles * std::max(0, g.tail_split - 1);
    for (int r = 0; r < g.n_ranges; ++r) {
        int tiles_r;
        const FusedArgs ar = range_args(a, g, r, tiles_r);
        const int tail_r = (r == g.n_ranges - 1 && g.tail_split > 1) ? g.tail_tiles : 0;     // tiles of this range launched apart
        FusedArgs at = ar;
        {
            ScopedTimer tm("fused_tile_train", st);
            if (g.sk_wgs > 0) {                     // stream-K: a.sk_tiles tiles x sk_chunks chunks over sk_wgs workgroups
                FusedArgs as = ar;
                as.sk_tiles = tiles_r;
                e = launch_fused64(mode, as, g.sk_wgs, 1, st);
            } else if (tail_r > 0) {
                const int main_r = tiles_r - tail_r;
                e = hipSuccess;
                if (main_r > 0) {
                    FusedArgs am = ar;
                    am.N = main_r * NT;
                    e = launch_fused64(mode, am, main_r, 1, st);
                }
                if (e == hipSuccess) {
                    at = tile_window(ar, g, main_r);
                    at.b_per_block = g.tail_b_per_block;
                    e = launch_fused64(mode, at, tail_r, g.tail_split, st);
                }
            } else {
                e = g.tile_w == 64 ? launch_fused64(mode, ar, tiles_r, g.b_split, st) : launch_fused32(mode, ar, 2 * tiles_r, g.b_split, st);
            }
            if (e != hipSuccess) return fail_hip(e, "fused_tile_kernel<train>");
        }
        if (loss_only) continue;
        // the partial slabs' reduction: on the side stream beside the dQ launch where there is one (see SideStream)
        const bool reduce = tail_r > 0 || g.sk_wgs > 0 || g.b_split > 1;
        SideStream *side = reduce ? side_stream() : nullptr;
        hipStream_t rs = st;
        if (side) {
            if (hipEventRecord(side->fork, st) == hipSuccess && hipStreamWaitEvent(side->s, side->fork, 0) == hipSuccess) rs = side->s;
            else side = nullptr;
        }
        if (tail_r > 0) {
            ScopedTimer tm("dc_reduce", rs);
            e = launch_dc_reduce(a.dC_slab, g.tail_split, tail_r * NT, g.D16, at.N, g.d, at.cand_ids, at.cand_first, a.cand_exclusive,
                                 a.grads_zero, dE, a.n_table_rows, a.id_err, rs);
            if (e != hipSuccess) return fail_hip(e, "dc_reduce");
        }
        if (g.sk_wgs > 0) {
            ScopedTimer tm("dc_reduce", rs);
            e = launch_dc_reduce_streamk(a.dC_slab, tiles_r, g.sk_chunks, g.sk_wgs, g.D16, ar.N, g.d, cand->ids, cand->first_id,
                                         a.cand_exclusive, a.grads_zero, dE, a.n_table_rows, a.id_err, rs);
            if (e != hipSuccess) return fail_hip(e, "dc_reduce_streamk");
        } else if (g.b_split > 1) {                 // (few candidate tiles: always a single range)
            ScopedTimer tm("dc_reduce", rs);
            e = launch_dc_reduce(a.dC_slab, g.b_split, g.tiles * NT, g.D16, g.N, g.d, cand->ids, cand->first_id, a.cand_exclusive,
                                 a.grads_zero, dE, a.n_table_rows, a.id_err, rs);
            if (e != hipSuccess) return fail_hip(e, "dc_reduce");
        }
        if (side && hipEventRecord(side->join, side->s) != hipSuccess) return fail(OKGE_ERR_HIP, "side stream: event record");
        {
            ScopedTimer tm("dq", st);
            q.N = ar.N;
            q.accumulate = r > 0 ? 1 : 0;           // later ranges add to the slabs of the first
            e = launch_dq(q, (g.Bpad / BC) * g.nsplit, st);
            if (e != hipSuccess) return fail_hip(e, "dq_kernel");
        }
        // join: the next range's tile launch rewrites the slabs, the prefix backward adds into dE rows the reduction stores
        if (side && hipStreamWaitEvent(st, side->join, 0) != hipSuccess) return fail(OKGE_ERR_HIP, "side stream: event wait");
    }
    if (loss_only) {
        ScopedTimer tm("loss_reduce", st);
        e = launch_loss_reduce(a.loss_partial, n_loss_partials, loss_out, st);
        if (e != hipSuccess) return fail_hip(e, "loss_reduce");
        return OKGE_OK;
    }
    if (dq_out) {
        ScopedTimer tm("slab_reduce", st);          // + the deterministic loss reduction (one extra workgroup)
        e = launch_slab_reduce(q.slab, g.nsplit, (int64_t)g.Bpad * g.ldq, dq_out, a.loss_partial, n_loss_partials,
                               loss_out, st);
        if (e != hipSuccess) return fail_hip(e, "slab_reduce");
        return OKGE_OK;
    }
    {
        ScopedTimer tm("prefix_backward", st);      // + the deterministic loss reduction (one extra workgroup)
        const PrefixDev p = to_dev(*batch, t, sh);
        AdagradFuse af = {};
        if (opt) {
            af.E = const_cast<float *>(t->E); af.sumE = opt->sum_E; af.dE = dE; af.R = const_cast<float *>(t->R); af.sumR = opt->sum_R; af.dR = dR;
            af.flags = opt->prefix_flags; af.n_ent = t->n_ent; af.n_rel = t->n_rel; af.d = t->d; af.zero_dE = opt->zero_entity_grad; af.on = 1;
            af.lr = opt->lr; af.wd = opt->weight_decay; af.eps = opt->eps;
        }
        e = launch_prefix_backward(t->E, t->R, t->d, t->scorer, p, q.slab, g.nsplit, g.Bpad, g.ldq, nullptr, dE, dR,
                                   a.loss_partial, n_loss_partials, loss_out, st, nullptr, nullptr, 0, nullptr, nullptr, 0, nullptr,
                                   (flags & OKGE_TRAIN_DISTINCT_PREFIX_ROWS) ? 1 : 0, opt ? &af : nullptr);
        if (e != hipSuccess) return fail_hip(e, "prefix_backward");
        if (opt) {
            ScopedTimer tf("adagrad_finish", st);
            e = launch_adagrad_finish(af, p, st);
            if (e != hipSuccess) return fail_hip(e, "adagrad_finish");
        }
    }
    return OKGE_OK;
}

int okge_train_forward_backward(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                                const okge_positives *pos, int32_t loss_kind, float label_smoothing,
                                double normalizer, int32_t flags, double *loss_out, float *dE, float *dR,
                                float *scores, int64_t ld_scores, void *workspace, size_t workspace_bytes,
                                void *stream)
{
    if (int rc = check_common(t, batch, cand)) return rc;
    if (!(flags & OKGE_TRAIN_LOSS_ONLY) && !dR) return fail(OKGE_ERR_INVALID, "null output");
    return train_core(t, nullptr, batch, nullptr, 0, batch->n_po + batch->n_sp, cand, pos, loss_kind, label_smoothing,
                      normalizer, cand->n, flags, loss_out, dE, dR, nullptr, scores, ld_scores, nullptr, workspace,
                      workspace_bytes, stream);
}

int okge_train_step(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand, const okge_positives *pos,
                    int32_t loss_kind, float label_smoothing, double normalizer, int32_t flags, const okge_adagrad *opt,
                    double *loss_out, float *dE, float *dR, void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_common(t, batch, cand)) return rc;
    if (!dR || !opt || !opt->sum_E || !opt->sum_R || !opt->prefix_flags) return fail(OKGE_ERR_INVALID, "null output / optimizer state");
    if (flags & (OKGE_TRAIN_LOSS_ONLY | OKGE_TRAIN_DISTINCT_PREFIX_ROWS))
        return fail(OKGE_ERR_INVALID, "okge_train_step updates the tables: not with LOSS_ONLY / DISTINCT_PREFIX_ROWS");
    if (t->scorer == OKGE_DISTMULT ? t->d % 4 : t->d % 8)
        return fail(OKGE_ERR_UNSUPPORTED, "okge_train_step needs a slot size that is a multiple of 4 (DistMult) / 8 (ComplEx)");
    if ((reinterpret_cast<uintptr_t>(t->E) | reinterpret_cast<uintptr_t>(t->R) | reinterpret_cast<uintptr_t>(dE) | reinterpret_cast<uintptr_t>(dR) |
         reinterpret_cast<uintptr_t>(opt->sum_E) | reinterpret_cast<uintptr_t>(opt->sum_R)) % 16)
        return fail(OKGE_ERR_INVALID, "tables, gradients and accumulators must be 16-byte aligned");
    return train_core(t, nullptr, batch, nullptr, 0, batch->n_po + batch->n_sp, cand, pos, loss_kind, label_smoothing, normalizer,
                      cand->n, flags, loss_out, dE, dR, nullptr, nullptr, 0, nullptr, workspace, workspace_bytes, stream, opt);
}

// ---- entity-sharded phases ---------------------------------------------------------------------------------------
int64_t okge_query_ld(int32_t d)
{
    Geometry g;
    return make_geometry(1, 1, d, g) ? g.ldq : 0;
}

int32_t okge_query_rows(int32_t B) { return B > 0 ? (B + BC - 1) / BC * BC : 0; }

static int check_shard(const okge_tables *t, const okge_shard *sh)
{
    if (!sh) return fail(OKGE_ERR_INVALID, "null shard descriptor");
    if (sh->ent_lo < 0 || sh->ent_hi <= sh->ent_lo || sh->ent_hi - sh->ent_lo != t->n_ent)
        return fail(OKGE_ERR_INVALID, "shard range must cover exactly the rows of the local entity table");
    return OKGE_OK;
}

int okge_encode_queries(const okge_tables *t, const okge_shard *sh, const okge_prefix_batch *batch, float *Q,
                        int64_t ldq, float *ent_rows, void *stream)
{
    okge_candidates none;
    std::memset(&none, 0, sizeof(none));
    none.n = 1; none.first_id = 0;
    if (int rc = check_common(t, batch, &none)) return rc;
    if (int rc = check_shard(t, sh)) return rc;
    if ((!Q && !ent_rows) || ldq != okge_query_ld(t->d)) return fail(OKGE_ERR_INVALID, "bad query block");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PrefixDev p = to_dev(*batch, t, sh);
    ScopedTimer tm("encode_queries", st);
    hipError_t e = launch_encode_queries(t->E, t->R, t->d, t->scorer, p, Q, (int)ldq,
                                         okge_query_rows(batch->n_po + batch->n_sp), ent_rows, nullptr, 0, nullptr, 0,
                                         NT, 0, st);
    if (e != hipSuccess) return fail_hip(e, "encode_queries");
    return OKGE_OK;
}

int okge_fold_queries(const okge_tables *t, const okge_prefix_batch *batch, const float *ent_rows, int64_t ldq, float *Q,
                      void *stream)
{
    okge_candidates none;
    std::memset(&none, 0, sizeof(none));
    none.n = 1; none.first_id = 0;
    if (int rc = check_common(t, batch, &none)) return rc;
    if (!ent_rows || !Q || ldq != okge_query_ld(t->d)) return fail(OKGE_ERR_INVALID, "bad query block");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    PrefixDev p = to_dev(*batch, t);
    p.ent_lo = 0; p.ent_hi = 0x7fffffff; p.whole_table = 0;     // entity ids are global here and not dereferenced: only relations are
    ScopedTimer tm("fold_queries", st);
    hipError_t e = launch_fold_queries(t->R, t->d, t->scorer, p, ent_rows, Q, (int)ldq,
                                       okge_query_rows(batch->n_po + batch->n_sp), st);
    if (e != hipSuccess) return fail_hip(e, "fold_queries");
    return OKGE_OK;
}

int okge_train_tiles(const okge_tables *t, const okge_shard *sh, const float *Q, int64_t ldq, int32_t B,
                     const okge_candidates *cand, const okge_positives *pos, int32_t loss_kind, float label_smoothing,
                     double normalizer, int32_t n_cand_global, int32_t flags, const float *row_lse, double *loss_out,
                     float *dE, float *dQ, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!t || !cand || !t->E || t->d <= 0 || t->n_ent <= 0) return fail(OKGE_ERR_INVALID, "bad tables");
    if (t->d > 512) return fail(OKGE_ERR_UNSUPPORTED, "slot sizes above 512 are not supported by the tile kernels");
    if (int rc = check_shard(t, sh)) return rc;
    if (!Q || !dQ || B <= 0 || cand->n <= 0) return fail(OKGE_ERR_INVALID, "bad query block / candidates");
    if (!cand->ids && (cand->first_id < 0 || (int64_t)cand->first_id + cand->n > t->n_ent))
        return fail(OKGE_ERR_INVALID, "candidate range outside the local entity table");
    return train_core(t, sh, nullptr, Q, ldq, B, cand, pos, loss_kind, label_smoothing, normalizer, n_cand_global,
                      flags, loss_out, dE, nullptr, dQ, nullptr, 0, row_lse, workspace, workspace_bytes, stream);
}

static int check_query_call(const okge_tables *t, const okge_shard *sh, const float *Q, int64_t ldq, int32_t B,
                            const okge_candidates *cand)
{
    if (!t || !cand || !t->E || t->d <= 0 || t->n_ent <= 0) return fail(OKGE_ERR_INVALID, "bad tables");
    if (t->d > 512) return fail(OKGE_ERR_UNSUPPORTED, "slot sizes above 512 are not supported by the tile kernels");
    if (int rc = check_shard(t, sh)) return rc;
    if (!Q || B <= 0 || cand->n <= 0 || ldq != okge_query_ld(t->d))
        return fail(OKGE_ERR_INVALID, "bad query block / candidates");
    if (cand->table) return fail(OKGE_ERR_INVALID, "sharded calls take candidates from the local entity table");
    if (!cand->ids && (cand->first_id < 0 || (int64_t)cand->first_id + cand->n > t->n_ent))
        return fail(OKGE_ERR_INVALID, "candidate range outside the local entity table");
    return OKGE_OK;
}

int okge_score_queries(const okge_tables *t, const okge_shard *sh, const float *Q, int64_t ldq, int32_t B,
                       const okge_candidates *cand, float *scores, int64_t ld_scores, void *stream)
{
    if (int rc = check_query_call(t, sh, Q, ldq, B, cand)) return rc;
    if (!scores || ld_scores < cand->n) return fail(OKGE_ERR_INVALID, "bad scores buffer");
    Geometry g;
    make_geometry(B, cand->n, t->d, g);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    FusedArgs a;
    fill_fused_common(a, g, t, cand, nullptr);
    a.Q = Q;
    a.cand_col0 = sh->cand_col0;
    a.X = scores;
    a.ldx = ld_scores;
    a.x_vec_ok = (ld_scores % 4 == 0) && (reinterpret_cast<uintptr_t>(scores) % 16 == 0);
    a.b_per_block = g.Bpad;
    ScopedTimer tm("fused_tile_score", st);
    hipError_t e = launch_score_sweep(g, a, g.tiles, st);
    if (e != hipSuccess) return fail_hip(e, "fused_tile_kernel<score>");
    return OKGE_OK;
}

int okge_row_logsumexp(const okge_tables *t, const okge_shard *sh, const float *Q, int64_t ldq, int32_t B,
                       const okge_candidates *cand, float *row_lse, void *workspace, size_t workspace_bytes,
                       void *stream)
{
    if (int rc = check_query_call(t, sh, Q, ldq, B, cand)) return rc;
    if (!row_lse) return fail(OKGE_ERR_INVALID, "null output");
    Geometry g;
    make_geometry(B, cand->n, t->d, g);
    if (!workspace || workspace_bytes < g.lse_bytes) return fail(OKGE_ERR_WORKSPACE, "workspace too small");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    FusedArgs a;
    fill_fused_common(a, g, t, cand, ws);
    a.Q = Q;
    a.cand_col0 = sh->cand_col0;
    return lse_pass(g, a, ws, row_lse, st);
}

int okge_prefix_backward(const okge_tables *t, const okge_shard *sh, const okge_prefix_batch *batch, const float *dQ,
                         int64_t ldq, const float *ent_rows, float *dE, float *dR, void *stream)
{
    okge_candidates none;
    std::memset(&none, 0, sizeof(none));
    none.n = 1; none.first_id = 0;
    if (int rc = check_common(t, batch, &none)) return rc;
    if (int rc = check_shard(t, sh)) return rc;
    if (!dQ || !dE || !dR || ldq != okge_query_ld(t->d)) return fail(OKGE_ERR_INVALID, "bad prefix_backward arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PrefixDev p = to_dev(*batch, t, sh);
    const int B = batch->n_po + batch->n_sp;
    ScopedTimer tm("prefix_backward", st);
    hipError_t e = launch_prefix_backward(t->E, t->R, t->d, t->scorer, p, dQ, 1, okge_query_rows(B), (int)ldq, ent_rows,
                                          dE, dR, nullptr, 0, nullptr, st);
    if (e != hipSuccess) return fail_hip(e, "prefix_backward");
    return OKGE_OK;
}

int okge_prefix_backward_segmented(const okge_tables *t, const okge_shard *sh, const okge_prefix_batch *batch, const float *dQ,
                                   int64_t ldq, const float *ent_rows, const int32_t *rel_order, const int32_t *rel_seg_ptr,
                                   int32_t n_rel_seg, const int32_t *ent_order, const int32_t *ent_seg_ptr, int32_t n_ent_seg,
                                   float *grad_rows, float *dE, float *dR, void *stream)
{
    okge_candidates none;
    std::memset(&none, 0, sizeof(none));
    none.n = 1; none.first_id = 0;
    if (int rc = check_common(t, batch, &none)) return rc;
    if (int rc = check_shard(t, sh)) return rc;
    if (!dQ || !dE || !dR || ldq != okge_query_ld(t->d)) return fail(OKGE_ERR_INVALID, "bad prefix_backward arguments");
    const int B = batch->n_po + batch->n_sp;
    const bool has_r = rel_order || rel_seg_ptr || n_rel_seg, has_e = ent_order || ent_seg_ptr || n_ent_seg;
    if (!grad_rows || (!has_r && !has_e) || (has_r && (!rel_order || !rel_seg_ptr || n_rel_seg <= 0 || n_rel_seg > B)) ||
        (has_e && (!ent_order || !ent_seg_ptr || n_ent_seg <= 0 || n_ent_seg > B)))
        return fail(OKGE_ERR_INVALID, "bad row segments (order[B], seg_ptr[n_seg + 1], 1 <= n_seg <= B; grad_rows[2][rows][ldq])");
    const bool vec = t->scorer == OKGE_DISTMULT ? (t->d % 4 == 0) : (t->d % 8 == 0);
    if (!vec) return fail(OKGE_ERR_UNSUPPORTED, "segmented gradients need d % 8 == 0 (ComplEx) / d % 4 == 0 (DistMult)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const PrefixDev p = to_dev(*batch, t, sh);
    ScopedTimer tm("prefix_backward", st);
    hipError_t e = launch_prefix_backward(t->E, t->R, t->d, t->scorer, p, dQ, 1, okge_query_rows(B), (int)ldq, ent_rows,
                                          dE, dR, nullptr, 0, nullptr, st, rel_order, rel_seg_ptr, n_rel_seg, ent_order,
                                          ent_seg_ptr, n_ent_seg, grad_rows);
    if (e != hipSuccess) return fail_hip(e, "prefix_backward_segmented");
    return OKGE_OK;
}

int okge_encode_rows(const float *table, int32_t table_rows, int32_t d, const int32_t *ids, int32_t first_id, int32_t n,
                     const okge_dropout *drop, float *out, int64_t ld_out, void *stream)
{
    if (!table || !out || d <= 0 || n < 0 || ld_out < d || table_rows <= 0)
        return fail(OKGE_ERR_INVALID, "bad encode_rows arguments");
    if (!ids && (first_id < 0 || (int64_t)first_id + n > table_rows))
        return fail(OKGE_ERR_INVALID, "row range outside the table");
    okge_dropout none;
    std::memset(&none, 0, sizeof(none));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("encode_rows", st);
    hipError_t e = launch_encode_rows(table, table_rows, d, ids, first_id, n, to_dev(drop ? *drop : none), out, ld_out, id_err_ptr(), st);
    if (e != hipSuccess) return fail_hip(e, "encode_rows");
    return OKGE_OK;
}

size_t okge_prefix_score_backward_workspace_bytes(int32_t b, int32_t n, int32_t d)
{
    return b > 0 && n > 0 && d > 0 ? score_backward_workspace_bytes(b, n, d) : 0;
}

int okge_prefix_score_backward(int32_t scorer, int32_t sp, const float *g, int64_t ld_g, int32_t b, int32_t n, const float *ent,
                               int64_t ld_ent, const float *rel, int64_t ld_rel, const float *cand, int64_t ld_cand, int32_t d,
                               float *d_ent, float *d_rel, float *d_cand, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!g || !ent || !rel || !cand || b <= 0 || n <= 0 || d <= 0 || ld_g < n || ld_ent < d || ld_rel < d || ld_cand < d)
        return fail(OKGE_ERR_INVALID, "bad prefix_score_backward arguments");
    if (scorer != OKGE_COMPLEX && scorer != OKGE_DISTMULT) return fail(OKGE_ERR_INVALID, "unknown scorer");
    if (scorer == OKGE_COMPLEX && (d & 1)) return fail(OKGE_ERR_INVALID, "ComplEx needs an even slot size");
    if (!d_ent && !d_rel && !d_cand) return OKGE_OK;
    if (!workspace || workspace_bytes < score_backward_workspace_bytes(b, n, d) || reinterpret_cast<uintptr_t>(workspace) % 16)
        return fail(OKGE_ERR_WORKSPACE, "workspace too small or unaligned (okge_prefix_score_backward_workspace_bytes)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("score_backward", st);
    hipError_t e = launch_score_backward(scorer, sp != 0, g, ld_g, b, n, ent, ld_ent, rel, ld_rel, cand, ld_cand, d, d_ent, d_rel, d_cand,
                                         workspace, st);
    if (e != hipSuccess) return fail_hip(e, "score_backward");
    return OKGE_OK;
}

int okge_scatter_rows(const float *rows, int64_t ld, const int32_t *ids, const int32_t *order, int32_t first_id, int32_t n, int32_t d,
                      const okge_dropout *drop, float *table_grad, int32_t table_rows, void *stream)
{
    if (!rows || !table_grad || n < 0 || d <= 0 || ld < d || table_rows <= 0) return fail(OKGE_ERR_INVALID, "bad scatter_rows arguments");
    if (!ids && (first_id < 0 || (int64_t)first_id + n > table_rows)) return fail(OKGE_ERR_INVALID, "row range outside the table");
    if (ids && n > 1 && !order) return fail(OKGE_ERR_INVALID, "scatter_rows with ids needs the positions sorted by id");
    okge_dropout none;
    std::memset(&none, 0, sizeof(none));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("scatter_rows", st);
    hipError_t e = launch_scatter_rows(rows, ld, ids, order, first_id, n, d, to_dev(drop ? *drop : none), table_grad, table_rows, id_err_ptr(), st);
    if (e != hipSuccess) return fail_hip(e, "scatter_rows");
    return OKGE_OK;
}

int okge_score_triples(int32_t scorer, const float *subj, int64_t ld_subj, const float *rel, int64_t ld_rel,
                       const float *obj, int64_t ld_obj, int32_t n, int32_t d, float *out, void *stream)
{
    if (!subj || !rel || !obj || !out || n < 0 || d <= 0 || ld_subj < d || ld_rel < d || ld_obj < d)
        return fail(OKGE_ERR_INVALID, "bad score_triples arguments");
    if (scorer != OKGE_COMPLEX && scorer != OKGE_DISTMULT) return fail(OKGE_ERR_INVALID, "unknown scorer");
    if (scorer == OKGE_COMPLEX && (d & 1)) return fail(OKGE_ERR_INVALID, "ComplEx needs an even slot size");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("score_triples", st);
    hipError_t e = launch_score_triples(subj, ld_subj, rel, ld_rel, obj, ld_obj, n, d, scorer, out, st);
    if (e != hipSuccess) return fail_hip(e, "score_triples");
    return OKGE_OK;
}

// ---- token-pooled embedder -----------------------------------------------------------------------------------------
static int check_token_embedder(const okge_token_embedder *e, const int32_t *ids, int32_t first_id, int32_t n)
{
    if (!e || !e->W || !e->token_ids || e->d <= 0 || e->vocab <= 0 || e->n_ids <= 0 || e->max_len <= 0)
        return fail(OKGE_ERR_INVALID, "bad token embedder");
    if (e->pool < 0 || e->pool > 2) return fail(OKGE_ERR_INVALID, "pool must be 0 (sum), 1 (mean) or 2 (max)");
    if (e->max_len > 64) return fail(OKGE_ERR_UNSUPPORTED, "token sequences longer than 64 are not supported");
    if (n < 0 || (!ids && (first_id < 0 || (int64_t)first_id + n > e->n_ids)))
        return fail(OKGE_ERR_INVALID, "row range outside the token-id table");
    if (e->bn_weight && (!e->bn_bias || !e->bn_running_mean || !e->bn_running_var))
        return fail(OKGE_ERR_INVALID, "batch-norm needs weight, bias and running statistics");
    return OKGE_OK;
}

size_t okge_pool_workspace_bytes(int32_t n, int32_t d) { return n > 0 && d > 0 ? pool_workspace_bytes(n, d) : 0; }

// one okge_pool_call -> the kernels' descriptor; `partial` = this call's share of the workspace
static int pool_call_dev(const okge_pool_call &c, bool backward, bool training, char *&ws, size_t &ws_left, PoolCall &q)
{
    const okge_token_embedder *e = c.e;
    if (int rc = check_token_embedder(e, c.ids, c.first_id, c.n)) return rc;
    if (c.n <= 0) return fail(OKGE_ERR_INVALID, "a pooled call needs n > 0 (leave empty calls out of the batch)");
    if (!c.raw || c.ld < e->d) return fail(OKGE_ERR_INVALID, "bad row block");
    const bool bn = e->bn_weight != nullptr;
    std::memset(&q, 0, sizeof(q));
    q.W = e->W; q.tokens = e->token_ids; q.ids = c.ids; q.raw = c.raw; q.out = c.out; q.ld = c.ld;
    q.d = e->d; q.L = e->max_len; q.first_id = c.first_id; q.n = c.n; q.pool = e->pool; q.n_ids = e->n_ids; q.vocab = e->vocab;
    q.eps = e->bn_eps; q.momentum = e->bn_momentum;
    q.bn_weight = e->bn_weight; q.bn_bias = e->bn_bias; q.run_mean = e->bn_running_mean; q.run_var = e->bn_running_var;
    if (!backward) {
        if (!c.out) return fail(OKGE_ERR_INVALID, "bad output rows");
        if (bn && c.out == c.raw)
            return fail(OKGE_ERR_INVALID, "with batch-norm the raw pooled rows are kept for backward: out must differ from raw");
        if (bn && training && !c.saved)
            return fail(OKGE_ERR_INVALID, "training-mode batch-norm needs the saved-statistics buffer (4*d floats)");
        q.saved = bn && training ? c.saved : nullptr;
        if (training && c.row_touched) {                    // forward in training mode: every token row the call READS is stamped
            if (c.touched_stamp < 1 || c.touched_stamp > 255) return fail(OKGE_ERR_INVALID, "touched_stamp must lie in 1..255");
            q.touched = c.row_touched; q.touched_stamp = c.touched_stamp;
        }
    } else {
        if (!c.d_out || !c.dW) return fail(OKGE_ERR_INVALID, "bad backward arguments");
        if (bn && (!c.saved || !c.d_bn_weight || !c.d_bn_bias))
            return fail(OKGE_ERR_INVALID, "batch-norm backward needs saved statistics and gradient buffers");
        q.saved = bn ? c.saved : nullptr;
        q.dY = c.d_out; q.dW = c.dW; q.d_weight = c.d_bn_weight; q.d_bias = c.d_bn_bias;
        if (c.row_touched && (c.touched_stamp < 1 || c.touched_stamp > 255))
            return fail(OKGE_ERR_INVALID, "touched_stamp must lie in 1..255");
        q.touched = c.row_touched; q.touched_stamp = c.touched_stamp;
    }
    if (q.saved) {
        const size_t need = pool_workspace_bytes(c.n, e->d);
        if (!ws || ws_left < need) return fail(OKGE_ERR_WORKSPACE, "workspace too small (sum of okge_pool_workspace_bytes over the calls)");
        q.partial = reinterpret_cast<float *>(ws);
        ws += need;
        ws_left -= need;
    }
    return OKGE_OK;
}

int okge_pool_encode_calls(const okge_pool_call *calls, int32_t n_calls, int32_t training, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    if (!calls || n_calls <= 0 || n_calls > POOL_MAX_CALLS) return fail(OKGE_ERR_INVALID, "1 to 8 pooled calls per batch");
    PoolCall q[POOL_MAX_CALLS];
    char *ws = static_cast<char *>(workspace);
    size_t left = workspace_bytes;
    for (int i = 0; i < n_calls; ++i)
        if (int rc = pool_call_dev(calls[i], false, training != 0, ws, left, q[i])) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("pool_encode", st);
    hipError_t err = launch_pool_encode_calls(q, n_calls, training != 0, id_err_ptr(), st);
    if (err != hipSuccess) return fail_hip(err, "pool_encode");
    return OKGE_OK;
}

// the calls as the kernels see them, without workspace shares (sizing only)
static int pool_calls_for_sizing(const okge_pool_call *calls, int32_t n_calls, PoolCall *q)
{
    if (!calls || n_calls <= 0 || n_calls > POOL_MAX_CALLS) return fail(OKGE_ERR_INVALID, "1 to 8 pooled calls per batch");
    for (int i = 0; i < n_calls; ++i) {
        const okge_token_embedder *e = calls[i].e;
        if (int rc = check_token_embedder(e, calls[i].ids, calls[i].first_id, calls[i].n)) return rc;
        std::memset(&q[i], 0, sizeof(PoolCall));
        q[i].tokens = e->token_ids; q[i].d = e->d; q[i].L = e->max_len; q[i].n = calls[i].n; q[i].pool = e->pool;
        q[i].vocab = e->vocab; q[i].dW = calls[i].dW; q[i].touched = calls[i].row_touched;
    }
    return OKGE_OK;
}

size_t okge_pool_scatter_state_bytes(const okge_pool_call *calls, int32_t n_calls)
{
    PoolCall q[POOL_MAX_CALLS];
    if (pool_calls_for_sizing(calls, n_calls, q)) return 0;
    return pool_scatter_state_bytes(q, n_calls);
}

size_t okge_pool_backward_workspace_bytes(const okge_pool_call *calls, int32_t n_calls)
{
    PoolCall q[POOL_MAX_CALLS];
    if (pool_calls_for_sizing(calls, n_calls, q)) return 0;
    size_t bytes = pool_scatter_workspace_bytes(q, n_calls);
    for (int i = 0; i < n_calls; ++i) bytes += pool_workspace_bytes(calls[i].n, calls[i].e->d);
    return bytes;
}

int okge_pool_backward_calls(const okge_pool_call *calls, int32_t n_calls, void *workspace, size_t workspace_bytes,
                             void *scatter_state, size_t scatter_state_bytes, void *stream)
{
    if (!calls || n_calls <= 0 || n_calls > POOL_MAX_CALLS) return fail(OKGE_ERR_INVALID, "1 to 8 pooled calls per batch");
    PoolCall q[POOL_MAX_CALLS];
    char *ws = static_cast<char *>(workspace);
    size_t left = workspace_bytes;
    for (int i = 0; i < n_calls; ++i)
        if (int rc = pool_call_dev(calls[i], true, true, ws, left, q[i])) return rc;
    if (scatter_state) {
        const size_t need_state = pool_scatter_state_bytes(q, n_calls), need_ws = pool_scatter_workspace_bytes(q, n_calls);
        if (!need_state)
            return fail(OKGE_ERR_UNSUPPORTED, "the scatter plan needs sum / mean pooling, slot sizes that are a multiple of 4 and calls "
                                              "of one token table to agree in token matrix and touched map (pass scatter_state = NULL)");
        if (scatter_state_bytes < need_state) return fail(OKGE_ERR_WORKSPACE, "scatter state too small (okge_pool_scatter_state_bytes)");
        if (!ws || left < need_ws) return fail(OKGE_ERR_WORKSPACE, "workspace too small (okge_pool_backward_workspace_bytes)");
        if (reinterpret_cast<uintptr_t>(scatter_state) % 16 || reinterpret_cast<uintptr_t>(ws) % 16)
            return fail(OKGE_ERR_INVALID, "scatter state and workspace must be 16-byte aligned");
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("pool_backward", st);
    hipError_t err = launch_pool_backward_calls(q, n_calls, id_err_ptr(), st, scatter_state, scatter_state_bytes, scatter_state ? ws : nullptr, left);
    if (err != hipSuccess) return fail_hip(err, "pool_backward");
    return OKGE_OK;
}

int okge_pool_encode(const okge_token_embedder *e, const int32_t *ids, int32_t first_id, int32_t n, int32_t training,
                     float *raw, float *out, int64_t ld, float *saved, void *workspace, size_t workspace_bytes,
                     void *stream)
{
    if (int rc = check_token_embedder(e, ids, first_id, n)) return rc;
    if (n == 0) return OKGE_OK;
    okge_pool_call c;
    std::memset(&c, 0, sizeof(c));
    c.e = e; c.ids = ids; c.first_id = first_id; c.n = n; c.raw = raw; c.out = out; c.ld = ld; c.saved = saved;
    return okge_pool_encode_calls(&c, 1, training, workspace, workspace_bytes, stream);
}

int okge_pool_backward(const okge_token_embedder *e, const int32_t *ids, int32_t first_id, int32_t n, const float *raw,
                       const float *d_out, int64_t ld, float *saved, float *dW, float *d_bn_weight, float *d_bn_bias,
                       void *workspace, size_t workspace_bytes, void *stream)
{
    if (int rc = check_token_embedder(e, ids, first_id, n)) return rc;
    if (n == 0) return OKGE_OK;
    okge_pool_call c;
    std::memset(&c, 0, sizeof(c));
    c.e = e; c.ids = ids; c.first_id = first_id; c.n = n; c.raw = const_cast<float *>(raw); c.ld = ld; c.saved = saved;
    c.d_out = d_out; c.dW = dW; c.d_bn_weight = d_bn_weight; c.d_bn_bias = d_bn_bias;
    return okge_pool_backward_calls(&c, 1, workspace, workspace_bytes, nullptr, 0, stream);
}

int okge_scale_inplace(float *x, int64_t n, const float *alpha_dev, void *stream)
{
    if (!x || !alpha_dev || n < 0) return fail(OKGE_ERR_INVALID, "bad scale arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("scale", st);
    hipError_t e = launch_scale(x, n, alpha_dev, st);
    if (e != hipSuccess) return fail_hip(e, "scale");
    return OKGE_OK;
}

int okge_rescale_gradients(float *g0, int64_t n0, float *g1, int64_t n1, const float *alpha_dev, float applied, void *stream)
{
    if (!g0 || !alpha_dev || n0 < 0 || n1 < 0 || (n1 > 0 && !g1) || !(applied > 0.f))
        return fail(OKGE_ERR_INVALID, "bad rescale arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("rescale", st);
    hipError_t e = launch_rescale2(g0, n0, g1 ? g1 : g0, g1 ? n1 : 0, alpha_dev, applied, st);
    if (e != hipSuccess) return fail_hip(e, "rescale");
    return OKGE_OK;
}

int okge_adagrad_step(float *p, float *g, float *state_sum, int64_t n, float lr, float weight_decay, float eps,
                      int32_t zero_grad, void *stream)
{
    if (!p || !g || !state_sum || n < 0) return fail(OKGE_ERR_INVALID, "bad adagrad arguments");
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(state_sum)) % 16)
        return fail(OKGE_ERR_INVALID, "adagrad buffers must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("adagrad", st);
    hipError_t e = launch_adagrad(p, g, state_sum, n, lr, weight_decay, eps, zero_grad, st);
    if (e != hipSuccess) return fail_hip(e, "adagrad");
    return OKGE_OK;
}

int okge_adagrad_step2(float *p0, float *g0, float *sum0, int64_t n0, float *p1, float *g1, float *sum1, int64_t n1,
                       float lr, float weight_decay, float eps, int32_t zero_grad, void *stream)
{
    if (!p0 || !g0 || !sum0 || n0 < 0 || !p1 || !g1 || !sum1 || n1 < 0)
        return fail(OKGE_ERR_INVALID, "bad adagrad arguments");
    if ((reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(g0) | reinterpret_cast<uintptr_t>(sum0) |
         reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(g1) | reinterpret_cast<uintptr_t>(sum1)) % 16)
        return fail(OKGE_ERR_INVALID, "adagrad buffers must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("adagrad", st);
    hipError_t e = launch_adagrad2(p0, g0, sum0, n0, p1, g1, sum1, n1, lr, weight_decay, eps, zero_grad, st);
    if (e != hipSuccess) return fail_hip(e, "adagrad");
    return OKGE_OK;
}

int okge_adagrad_multi(const okge_adagrad_tensor *tensors, int32_t n_tensors, float lr, float weight_decay, float eps, void *stream)
{
    if (!tensors || n_tensors <= 0 || n_tensors > ADAGRAD_MAX_SEGS) return fail(OKGE_ERR_INVALID, "1 to 4 tensors per adagrad launch");
    AdagradSegM segs[ADAGRAD_MAX_SEGS];
    for (int i = 0; i < n_tensors; ++i) {
        const okge_adagrad_tensor &t = tensors[i];
        if (!t.p || !t.g || !t.state_sum || t.n < 0) return fail(OKGE_ERR_INVALID, "bad adagrad arguments");
        if ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.state_sum)) % 16)
            return fail(OKGE_ERR_INVALID, "adagrad buffers must be 16-byte aligned");
        if (t.row_touched) {
            if (t.row_len <= 0 || t.row_len % 4 || t.n % t.row_len || t.n / t.row_len > INT32_MAX)
                return fail(OKGE_ERR_INVALID, "a touched-row map needs rows of a multiple of 4 floats that tile the tensor");
            if (t.touched_stamp < 1 || t.touched_stamp > 255) return fail(OKGE_ERR_INVALID, "touched_stamp must lie in 1..255");
            if (!t.zero_grad) return fail(OKGE_ERR_INVALID, "a touched-row map needs zero_grad (rows not stamped must hold zero gradients)");
        }
        if (t.rows < 0 || t.rows > 2 || (t.rows && !t.row_touched)) return fail(OKGE_ERR_INVALID, "rows: 0 all, 1 unstamped only, 2 stamped only (1 / 2 need a map)");
        segs[i] = AdagradSegM{t.p, t.g, t.state_sum, t.n, t.row_touched, t.row_len, t.touched_stamp, t.zero_grad, t.rows};
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("adagrad", st);
    hipError_t e = launch_adagrad_multi(segs, n_tensors, lr, weight_decay, eps, st);
    if (e != hipSuccess) return fail_hip(e, "adagrad");
    return OKGE_OK;
}

static int lazy_tensor_dev(const okge_lazy_tensor &t, LazySeg &sg)
{
    if (!t.p || !t.g || !t.state_sum || t.rows < 0 || t.row_len <= 0) return fail(OKGE_ERR_INVALID, "bad lazy adagrad tensor");
    if (t.row_steps) {
        if (t.row_len % 4 || t.rows > INT32_MAX) return fail(OKGE_ERR_INVALID, "deferred rows need a row length that is a multiple of 4 and fewer than 2^31 rows");
        if ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(t.g) | reinterpret_cast<uintptr_t>(t.state_sum)) % 16)
            return fail(OKGE_ERR_INVALID, "adagrad buffers must be 16-byte aligned");
        if (t.row_touched && (t.touched_stamp < 1 || t.touched_stamp > 255)) return fail(OKGE_ERR_INVALID, "touched_stamp must lie in 1..255");
    } else if (t.row_touched) return fail(OKGE_ERR_INVALID, "a touched-row map needs row_steps");
    sg = LazySeg{t.p, t.g, t.state_sum, t.row_steps, t.row_touched, t.rows, t.row_len, t.touched_stamp};
    return OKGE_OK;
}

int okge_adagrad_lazy(const okge_lazy_tensor *tensors, int32_t n_tensors, int32_t *counters, int32_t window, int32_t mode, float lr,
                      float weight_decay, float eps, void *stream)
{
    if (!tensors || n_tensors <= 0 || n_tensors > ADAGRAD_MAX_SEGS) return fail(OKGE_ERR_INVALID, "1 to 4 tensors per adagrad launch");
    if (!counters || window < 1 || (mode != OKGE_LAZY_STEP && mode != OKGE_LAZY_FLUSH))
        return fail(OKGE_ERR_INVALID, "okge_adagrad_lazy needs the counters, window >= 1 and mode OKGE_LAZY_STEP / OKGE_LAZY_FLUSH");
    LazySeg segs[ADAGRAD_MAX_SEGS];
    for (int i = 0; i < n_tensors; ++i)
        if (int rc = lazy_tensor_dev(tensors[i], segs[i])) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm(mode == OKGE_LAZY_STEP ? "adagrad" : "adagrad_flush", st);
    hipError_t e = launch_adagrad_lazy(segs, n_tensors, counters, window, mode, lr, weight_decay, eps, st);
    if (e != hipSuccess) return fail_hip(e, "adagrad_lazy");
    return OKGE_OK;
}

int okge_pool_catch_up_calls(const okge_pool_call *calls, int32_t n_calls, const okge_lazy_tensor *tables, int32_t n_tables,
                             const int32_t *counters, float lr, float weight_decay, float eps, void *stream)
{
    if (!calls || n_calls <= 0 || n_calls > POOL_MAX_CALLS) return fail(OKGE_ERR_INVALID, "1 to 8 pooled calls per batch");
    if (!tables || n_tables <= 0 || n_tables > ADAGRAD_MAX_SEGS || !counters) return fail(OKGE_ERR_INVALID, "bad catch-up arguments");
    PoolCall q[POOL_MAX_CALLS];
    int nq = 0;
    for (int i = 0; i < n_calls; ++i) {
        const okge_token_embedder *e = calls[i].e;
        if (int rc = check_token_embedder(e, calls[i].ids, calls[i].first_id, calls[i].n)) return rc;
        if (calls[i].n <= 0) return fail(OKGE_ERR_INVALID, "a pooled call needs n > 0 (leave empty calls out of the batch)");
        for (int j = 0; j < n_tables; ++j) {
            if (tables[j].p != e->W || !tables[j].row_steps) continue;
            LazySeg sg;
            if (int rc = lazy_tensor_dev(tables[j], sg)) return rc;
            if (tables[j].rows != e->vocab || tables[j].row_len != e->d) return fail(OKGE_ERR_INVALID, "lazy tensor and token table differ in shape");
            PoolCall &c = q[nq++];
            std::memset(&c, 0, sizeof(c));
            c.W = e->W; c.tokens = e->token_ids; c.ids = calls[i].ids; c.d = e->d; c.L = e->max_len; c.first_id = calls[i].first_id;
            c.n = calls[i].n; c.n_ids = e->n_ids; c.vocab = e->vocab; c.sumW = sg.s; c.steps = sg.steps;
            break;
        }
    }
    if (!nq) return OKGE_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("pool_catch_up", st);
    hipError_t err = launch_pool_catch_up(q, nq, counters, lr, weight_decay, eps, id_err_ptr(), st);
    if (err != hipSuccess) return fail_hip(err, "pool_catch_up");
    return OKGE_OK;
}

int okge_filtered_ranks(const float *scores, int64_t ld_scores, int32_t B, int32_t N, const int64_t *filt_ptr,
                        const int32_t *filt_col, const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                        int64_t *ranks, void *stream)
{
    if (!scores || !filt_ptr || !row_ptr || !grp_ptr || !ids || !ranks || B <= 0 || N <= 0 || ld_scores < N)
        return fail(OKGE_ERR_INVALID, "bad rank arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("filtered_ranks", st);
    hipError_t e = launch_ranks(scores, ld_scores, B, N, filt_ptr, filt_col, row_ptr, grp_ptr, ids, ranks, 0, nullptr,
                                nullptr, nullptr, st);
    if (e != hipSuccess) return fail_hip(e, "ranks");
    return OKGE_OK;
}

int okge_rank_metrics(const int64_t *ranks, int64_t n, double *acc, void *stream)
{
    if (!ranks || !acc || n < 0) return fail(OKGE_ERR_INVALID, "bad rank_metrics arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("rank_metrics", st);
    hipError_t e = launch_rank_metrics(ranks, n, acc, st);
    if (e != hipSuccess) return fail_hip(e, "rank_metrics");
    return OKGE_OK;
}

// One evaluation batch on two streams: scores on `stream`, ranks + meters on `rank_stream`, ordered by events the
// library owns.  A score buffer is reused only after the ranks of the batch that last used it are counted.
}  // extern "C"
namespace {
struct EvalSlot { const float *buf; hipEvent_t scored, ranked; bool used; uint64_t last_use; };
std::mutex g_eval_mu;
std::vector<EvalSlot> g_eval_slots;
uint64_t g_eval_clock = 0;
constexpr size_t EVAL_SLOTS_MAX = 8;      // a caller alternates two (or a few) score buffers; older entries are recycled
EvalSlot &eval_slot(const float *buf)
{
    ++g_eval_clock;
    for (auto &sl : g_eval_slots)
        if (sl.buf == buf) { sl.last_use = g_eval_clock; return sl; }
    if (g_eval_slots.size() >= EVAL_SLOTS_MAX) {          // recycle the least recently used entry (its events are reused:
        size_t lru = 0;                                   // a wait on the old `ranked` event is harmless, never wrong)
        for (size_t i = 1; i < g_eval_slots.size(); ++i)
            if (g_eval_slots[i].last_use < g_eval_slots[lru].last_use) lru = i;
        g_eval_slots[lru].buf = buf;
        g_eval_slots[lru].last_use = g_eval_clock;
        return g_eval_slots[lru];
    }
    EvalSlot sl{buf, nullptr, nullptr, false, g_eval_clock};
    (void)hipEventCreateWithFlags(&sl.scored, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&sl.ranked, hipEventDisableTiming);
    g_eval_slots.push_back(sl);
    return g_eval_slots.back();
}
}  // namespace
extern "C" {

int okge_evaluate_batch(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                        const int64_t *filt_ptr, const int32_t *filt_col, const int64_t *row_ptr, const int64_t *grp_ptr,
                        const int32_t *ids, int64_t n_groups, float *scores, int64_t ld_scores, int64_t *ranks, double *acc,
                        void *workspace, size_t workspace_bytes, void *stream, void *rank_stream)
{
    if (!ranks || !acc || n_groups < 0) return fail(OKGE_ERR_INVALID, "bad evaluate arguments");
    hipStream_t s0 = reinterpret_cast<hipStream_t>(stream), s1 = reinterpret_cast<hipStream_t>(rank_stream);
    std::lock_guard<std::mutex> lk(g_eval_mu);
    EvalSlot &sl = eval_slot(scores);
    hipError_t e;
    if (sl.used && s0 != s1) {
        e = hipStreamWaitEvent(s0, sl.ranked, 0);
        if (e != hipSuccess) return fail_hip(e, "wait for the buffer's previous ranks");
    }
    if (int rc = okge_score_prefixes(t, batch, cand, scores, ld_scores, workspace, workspace_bytes, stream)) return rc;
    if (s0 != s1) {
        e = hipEventRecord(sl.scored, s0);
        if (e == hipSuccess) e = hipStreamWaitEvent(s1, sl.scored, 0);
        if (e != hipSuccess) return fail_hip(e, "order ranks after scores");
    }
    const int32_t B = batch->n_po + batch->n_sp;
    if (int rc = okge_filtered_ranks(scores, ld_scores, B, cand->n, filt_ptr, filt_col, row_ptr, grp_ptr, ids, ranks, rank_stream))
        return rc;
    if (int rc = okge_rank_metrics(ranks, n_groups, acc, rank_stream)) return rc;
    if (s0 != s1) {
        e = hipEventRecord(sl.ranked, s1);
        if (e != hipSuccess) return fail_hip(e, "record ranks done");
        sl.used = true;
    }
    return OKGE_OK;
}

// ---- fused evaluation: no (B, N) score block ------------------------------------------------------------------------
namespace {
struct EvalGeometry { size_t off_Q, off_true, off_filt, off_rps, off_gshift, off_grow, off_counts, total; bool slab; };
constexpr size_t EVAL_SLAB_MAX = (size_t)256 << 20;      // per-tile count slabs up to 256 MB; beyond that: atomics
bool eval_geometry(int B, int N, int d, int64_t n_groups, int64_t n_filter, Geometry &g, EvalGeometry &e)
{
    if (!make_geometry(B, N, d, g) || n_groups < 0 || n_filter < 0) return false;
    size_t off = 0;
    e.off_Q = off;      off += align_up((size_t)g.Bpad * g.ldq * sizeof(float), 256);
    e.off_true = off;   off += align_up((size_t)std::max<int64_t>(n_groups, 1) * sizeof(float), 256);
    e.off_filt = off;   off += align_up((size_t)std::max<int64_t>(n_filter, 1) * sizeof(float), 256);
    e.off_rps = off;    off += align_up((size_t)(g.Bpad + 1) * sizeof(int64_t), 256);       // row_ptr of the batch sorted by group count
    e.off_gshift = off; off += align_up((size_t)g.Bpad * sizeof(int64_t), 256);             // original -> sorted group index, per row
    e.off_grow = off;   off += align_up((size_t)std::max<int64_t>(n_groups, 1) * sizeof(int32_t), 256);     // row of every group
    const size_t slab_bytes = (size_t)g.tiles * std::max<int64_t>(n_groups, 1) * sizeof(uint32_t);
    e.slab = slab_bytes <= EVAL_SLAB_MAX;
    e.off_counts = off; off += align_up(e.slab ? slab_bytes : (size_t)std::max<int64_t>(n_groups, 1) * 2 * sizeof(int32_t), 256);
    e.total = off;
    return true;
}
}  // namespace

size_t okge_eval_workspace_bytes(int32_t B, int32_t N, int32_t d, int64_t n_groups, int64_t n_filter)
{
    Geometry g;
    EvalGeometry e;
    return eval_geometry(B, N, d, n_groups, n_filter, g, e) ? e.total : 0;
}

// phases: 1 = point scores (+ queries), 2 = tile sweep, 4 = ranks + meters; okge_evaluate_fused runs all three on one
// stream, okge_evaluate_fused_phase one at a time, okge_evaluate_fused_batches pairs phase 4 of a batch with phase 1 of
// the next in one launch.
namespace {
struct EvalCall {                      // one batch's launch arguments, built once and used by whichever phases run
    Geometry g;
    EvalGeometry eg;
    EvalPointsArgs pts;
    EvalRanksArgs rk;
    FusedArgs sweep;
    int32_t *counts;
    int64_t n_groups;
};
}  // namespace

// Candidate-sharded call (okge_evaluate_fused_shard): folded queries from outside, the shard's candidate columns, the
// true scores in a caller-owned buffer (they are exchanged between the phases), counts instead of ranks.
struct EvalShard {
    const float *Q_in;
    int32_t      B, col_lo, n_cand_global;
    float       *true_scores;
    int64_t     *counts_out;
};

static int eval_call(EvalCall &c, const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                     const int64_t *filt_ptr, const int32_t *filt_col, int64_t n_filter, const int64_t *row_ptr,
                     const int64_t *grp_ptr, const int32_t *ids, int64_t n_groups, int64_t *ranks, double *acc,
                     void *workspace, size_t workspace_bytes, const EvalShard *es = nullptr)
{
    if (!es) {
        if (int rc = check_common(t, batch, cand)) return rc;
    }
    if (!filt_ptr || !row_ptr || !grp_ptr || (n_groups > 0 && !ids) || (!es && (!ranks || !acc)) || n_groups < 0 || n_filter < 0 ||
        (n_filter > 0 && !filt_col))                      // (a batch without answer groups has no ids array)
        return fail(OKGE_ERR_INVALID, "bad evaluate arguments");
    if (t->d > 512) return fail(OKGE_ERR_UNSUPPORTED, "slot sizes above 512 are not supported by the tile kernels");
    if (cand->table || cand->drop.p > 0.f || (batch && (batch->drop_po_ent.p > 0.f || batch->drop_sp_ent.p > 0.f ||
        batch->drop_po_rel.p > 0.f || batch->drop_sp_rel.p > 0.f)))
        return fail(OKGE_ERR_UNSUPPORTED, "fused evaluation is the eval-mode path (no dropout, candidates from the entity table)");
    c.n_groups = n_groups;
    if (n_groups == 0) return OKGE_OK;
    const int32_t B = es ? es->B : batch->n_po + batch->n_sp;
    Geometry &g = c.g;
    EvalGeometry &eg = c.eg;
    eval_geometry(B, cand->n, t->d, n_groups, n_filter, g, eg);
    if (!workspace || workspace_bytes < eg.total) return fail(OKGE_ERR_WORKSPACE, "workspace too small");
    char *ws = static_cast<char *>(workspace);
    float *Q = reinterpret_cast<float *>(ws + eg.off_Q), *tru = es ? es->true_scores : reinterpret_cast<float *>(ws + eg.off_true);
    float *fx = reinterpret_cast<float *>(ws + eg.off_filt);
    c.counts = reinterpret_cast<int32_t *>(ws + eg.off_counts);
    int64_t *rps = reinterpret_cast<int64_t *>(ws + eg.off_rps), *gshift = reinterpret_cast<int64_t *>(ws + eg.off_gshift);
    int32_t *grow = reinterpret_cast<int32_t *>(ws + eg.off_grow);
    EvalPointsArgs &p = c.pts;
    p.E = t->E; p.R = t->R; p.Q = Q; p.cand_ids = cand->ids;
    if (es) {                                             // rows come folded: no prefix ids here, only their number
        std::memset(&p.p, 0, sizeof(p.p));
        p.p.n_po = B;
        p.p.id_err = id_err_ptr();
    } else {
        p.p = to_dev(*batch, t);
    }
    p.Q_in = es ? es->Q_in : nullptr;
    p.col_lo = es ? es->col_lo : 0;
    p.n_cand_global = es ? es->n_cand_global : cand->n;
    p.row_ptr = row_ptr; p.grp_ptr = grp_ptr; p.filt_ptr = filt_ptr; p.ids = ids; p.filt_col = filt_col;
    p.true_out = tru; p.filt_x = fx; p.row_ptr_sorted = rps; p.gshift = gshift; p.group_row = grow;
    p.table_rows = t->n_ent; p.d = t->d; p.scorer = t->scorer; p.ldq = g.ldq; p.KB = g.KB; p.Bpad = g.Bpad;
    p.cand_first = cand->first_id; p.n_cand = cand->n;
    FusedArgs &a = c.sweep;
    fill_fused_common(a, g, t, cand, ws);
    a.Q = Q;
    a.rk_row_ptr = rps; a.rk_true = tru; a.rk_ngroups = n_groups;          // the sweep works on the sorted batch
    a.rk_counts = eg.slab ? nullptr : c.counts;
    a.rk_slab = eg.slab ? reinterpret_cast<uint32_t *>(c.counts) : nullptr;
    a.b_per_block = g.Bpad;
    EvalRanksArgs &r = c.rk;
    r.counts = a.rk_counts; r.slab = a.rk_slab; r.true_scores = tru; r.filt_x = fx; r.filt_ptr = filt_ptr; r.gshift = gshift;
    r.group_row = grow; r.ranks = ranks; r.acc = acc; r.n_groups = n_groups; r.tiles = g.tiles; r.B = B;
    r.counts_out = es ? es->counts_out : nullptr;
    return OKGE_OK;
}

static int eval_issue(int phases, const EvalCall &c, hipStream_t st)
{
    if (c.n_groups == 0) return OKGE_OK;
    hipError_t e = hipSuccess;
    if (phases & 1) {
        if (!c.eg.slab) {
            e = hipMemsetAsync(c.counts, 0, (size_t)c.n_groups * 2 * sizeof(int32_t), st);
            if (e != hipSuccess) return fail_hip(e, "clear rank counters");
        }
        ScopedTimer tm("eval_points", st);
        e = launch_eval_side(&c.pts, nullptr, st);
        if (e != hipSuccess) return fail_hip(e, "eval_points");
    }
    if (phases & 2) {
        ScopedTimer tm("fused_tile_count", st);
        // (slot sizes above 256: the register-tile kernel's counting mode in a stream-K launch; not on the OKGE_TILE_W=32 cut)
        if (c.g.KB > 16 && c.g.tile_w == 32) return fail(OKGE_ERR_UNSUPPORTED, "the 32-wide cut (OKGE_TILE_W=32) has no counting sweep above slot size 256");
        e = launch_score_sweep(c.g, c.sweep, c.g.tiles, st, MODE_COUNT);
        if (e != hipSuccess) return fail_hip(e, "fused_tile_kernel<count>");
    }
    if (phases & 4) {
        ScopedTimer tm("eval_ranks", st);
        e = launch_eval_side(nullptr, &c.rk, st);
        if (e != hipSuccess) return fail_hip(e, "eval_ranks");
    }
    return OKGE_OK;
}

static int evaluate_fused_impl(int phases, const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                               const int64_t *filt_ptr, const int32_t *filt_col, int64_t n_filter, const int64_t *row_ptr,
                               const int64_t *grp_ptr, const int32_t *ids, int64_t n_groups, int64_t *ranks, double *acc,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    EvalCall c;
    if (int rc = eval_call(c, t, batch, cand, filt_ptr, filt_col, n_filter, row_ptr, grp_ptr, ids, n_groups, ranks, acc, workspace,
                           workspace_bytes))
        return rc;
    return eval_issue(phases, c, reinterpret_cast<hipStream_t>(stream));
}

int okge_evaluate_fused(const okge_tables *t, const okge_prefix_batch *batch, const okge_candidates *cand,
                        const int64_t *filt_ptr, const int32_t *filt_col, int64_t n_filter, const int64_t *row_ptr,
                        const int64_t *grp_ptr, const int32_t *ids, int64_t n_groups, int64_t *ranks, double *acc,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    return evaluate_fused_impl(7, t, batch, cand, filt_ptr, filt_col, n_filter, row_ptr, grp_ptr, ids, n_groups, ranks, acc,
                               workspace, workspace_bytes, stream);
}

int okge_evaluate_fused_shard(int32_t phase, const okge_tables *t, const okge_shard *sh, const float *Q, int64_t ldq, int32_t B,
                              const okge_candidates *cand, int32_t n_cand_global, const int64_t *filt_ptr, const int32_t *filt_col,
                              int64_t n_filter, const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                              int64_t n_groups, float *true_scores, int64_t *counts, void *workspace, size_t workspace_bytes,
                              void *stream)
{
    if (phase != 1 && phase != 2 && phase != 4) return fail(OKGE_ERR_INVALID, "phase must be 1 (points), 2 (sweep) or 4 (counts)");
    if (!t || !t->E || !t->R || t->d <= 0 || t->n_ent <= 0 || !cand || cand->n <= 0 || B <= 0 || !Q || !true_scores || !counts)
        return fail(OKGE_ERR_INVALID, "bad sharded evaluate arguments");
    if (int rc = check_shard(t, sh)) return rc;
    if (ldq != okge_query_ld(t->d)) return fail(OKGE_ERR_INVALID, "query block leading dimension must be okge_query_ld(d)");
    if (sh->cand_col0 < 0 || (int64_t)sh->cand_col0 + cand->n > n_cand_global)
        return fail(OKGE_ERR_INVALID, "the shard's candidate columns must lie inside the global candidate list");
    if (!cand->ids && (cand->first_id < 0 || (int64_t)cand->first_id + cand->n > t->n_ent))
        return fail(OKGE_ERR_INVALID, "candidate range outside the local entity table");
    EvalShard es;
    es.Q_in = Q; es.B = B; es.col_lo = sh->cand_col0; es.n_cand_global = n_cand_global;
    es.true_scores = true_scores; es.counts_out = counts;
    EvalCall c;
    if (int rc = eval_call(c, t, nullptr, cand, filt_ptr, filt_col, n_filter, row_ptr, grp_ptr, ids, n_groups, nullptr, nullptr,
                           workspace, workspace_bytes, &es))
        return rc;
    return eval_issue(phase, c, reinterpret_cast<hipStream_t>(stream));
}

// events of okge_evaluate_fused_batches (fork / join of the extra streams): one set per device, created on first use on that
// device.  A set is only used between the fork and the join of ONE call, and calls on one device are issued by one host
// thread at a time (the first stream orders them), so evaluators sharing a device can share the set.
constexpr int EVAL_MAX_STREAMS = 4;
static hipError_t eval_events(hipEvent_t **out)
{
    static hipEvent_t ev[OKGE_MAX_DEVICES][EVAL_MAX_STREAMS];
    static bool ready[OKGE_MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_dev_mu);
    if (!ready[dev]) {
        for (int i = 0; i < EVAL_MAX_STREAMS; ++i) {
            hipError_t e = hipEventCreateWithFlags(&ev[dev][i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        ready[dev] = true;
    }
    *out = ev[dev];
    return hipSuccess;
}

int okge_evaluate_fused_batches(const okge_tables *t, const okge_eval_batch *batches, int32_t n_batches, int64_t *ranks,
                                double *acc, void *workspace, size_t workspace_bytes, void *const *streams, int32_t n_streams)
{
    if (!batches || n_batches < 0 || !ranks || !acc || !workspace || !streams || n_streams < 1 || n_streams > EVAL_MAX_STREAMS)
        return fail(OKGE_ERR_INVALID, "bad evaluate arguments (1 to 4 streams)");
    if (n_batches == 0) return OKGE_OK;
    hipStream_t st[EVAL_MAX_STREAMS];
    for (int i = 0; i < n_streams; ++i) {
        st[i] = reinterpret_cast<hipStream_t>(streams[i]);
        for (int j = 0; j < i; ++j)
            if (st[j] == st[i]) return fail(OKGE_ERR_INVALID, "okge_evaluate_fused_batches: the streams must differ");
    }
    const int S = n_streams, slots = 2 * S;
    const size_t slot_bytes = (workspace_bytes / slots) & ~(size_t)255;
    char *ws = static_cast<char *>(workspace);
    // everything that can be refused is refused before the first launch.  Stream, workspace slot and position in the chain
    // come from the batch's index IN THE CALL -- the caller numbers its rank regions by that index too --, so a batch without
    // answer groups keeps its place and simply launches nothing (dropping it would shift the batches behind it onto other
    // chains while their rank regions stay where the caller put them: two unordered chains writing one region).
    std::vector<EvalCall> calls((size_t)n_batches);
    int live = 0;
    for (int i = 0; i < n_batches; ++i) {
        const okge_eval_batch &b = batches[i];
        if (b.rank_offset < 0) return fail(OKGE_ERR_INVALID, "negative rank_offset");
        if (int rc = eval_call(calls[i], t, &b.batch, &b.cand, b.filt_ptr, b.filt_col, b.n_filter, b.row_ptr, b.grp_ptr, b.ids,
                               b.n_groups, ranks + b.rank_offset, acc, ws + (size_t)(i % slots) * slot_bytes, slot_bytes))
            return rc;
        live += calls[i].n_groups > 0;
    }
    const int n = n_batches;
    if (live == 0) return OKGE_OK;
    // Batch i runs on stream i % S; each stream is an independent chain [points i] [sweep i] [ranks i + points i+S]
    // [sweep i+S] ... with no dependency on the others, so the device fills one chain's small latency-bound launches (and
    // the CUs a sweep's tile grid leaves empty) with the other chains' work.  2 S workspace slots: two batches in flight
    // per chain.
    hipEvent_t *ev = nullptr;
    hipError_t e = hipSuccess;
#define OKGE_EV(call, what) do { e = (call); if (e != hipSuccess) return fail_hip(e, what); } while (0)
    if (S > 1) {
        OKGE_EV(eval_events(&ev), "create evaluation events");
        OKGE_EV(hipEventRecord(ev[0], st[0]), "record fork");          // the other streams join behind whatever produced
        for (int k = 1; k < S; ++k) OKGE_EV(hipStreamWaitEvent(st[k], ev[0], 0), "fork");   // the tables / batches on the first
    }
    auto clear_counts = [&](const EvalCall &c, hipStream_t s) {      // (atomics path of very large batches only)
        return c.eg.slab ? hipSuccess : hipMemsetAsync(c.counts, 0, (size_t)c.n_groups * 2 * sizeof(int32_t), s);
    };
    for (int i = 0; i < n; ++i) {
        hipStream_t s = st[i % S];
        const bool has = calls[i].n_groups > 0;
        if (i < S && has) {                                            // head of a chain
            OKGE_EV(clear_counts(calls[i], s), "clear rank counters");
            ScopedTimer tm("eval_points", s);
            OKGE_EV(launch_eval_side(&calls[i].pts, nullptr, s), "eval_points");
        }
        if (has)
            if (int rc = eval_issue(2, calls[i], s)) return rc;
        const EvalCall *nx = (i + S < n && calls[i + S].n_groups > 0) ? &calls[i + S] : nullptr;      // the chain's next batch
        if (nx) OKGE_EV(clear_counts(*nx, s), "clear rank counters");
        if (!nx && !has) continue;
        ScopedTimer tm("eval_ranks+points", s);
        OKGE_EV(launch_eval_side(nx ? &nx->pts : nullptr, has ? &calls[i].rk : nullptr, s), "eval_side");
    }
    for (int k = 1; k < S; ++k) {
        OKGE_EV(hipEventRecord(ev[k], st[k]), "record join");
        OKGE_EV(hipStreamWaitEvent(st[0], ev[k], 0), "join");
    }
#undef OKGE_EV
    return OKGE_OK;
}

int okge_group_true_scores(const float *scores, int64_t ld_scores, int32_t B, int32_t col0, int32_t n_local,
                           const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids, float *true_out,
                           void *stream)
{
    if (!scores || !row_ptr || !grp_ptr || !ids || !true_out || B <= 0 || n_local <= 0 || col0 < 0 || ld_scores < n_local)
        return fail(OKGE_ERR_INVALID, "bad rank arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("group_true_scores", st);
    hipError_t e = launch_ranks(scores, ld_scores, B, n_local, row_ptr /* unused */, nullptr, row_ptr, grp_ptr, ids, nullptr,
                                col0, nullptr, true_out, nullptr, st);
    if (e != hipSuccess) return fail_hip(e, "ranks<true>");
    return OKGE_OK;
}

int okge_rank_counts(const float *scores, int64_t ld_scores, int32_t B, int32_t col0, int32_t n_local,
                     const int64_t *filt_ptr, const int32_t *filt_col, const int64_t *row_ptr, const float *true_scores,
                     int64_t *counts, void *stream)
{
    if (!scores || !filt_ptr || !row_ptr || !true_scores || !counts || B <= 0 || n_local <= 0 || col0 < 0 ||
        ld_scores < n_local)
        return fail(OKGE_ERR_INVALID, "bad rank arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("rank_counts", st);
    hipError_t e = launch_ranks(scores, ld_scores, B, n_local, filt_ptr, filt_col, row_ptr, nullptr, nullptr, nullptr, col0,
                                true_scores, nullptr, counts, st);
    if (e != hipSuccess) return fail_hip(e, "ranks<counts>");
    return OKGE_OK;
}

int okge_id_errors(int64_t *n_out)
{
    int *p = id_err_ptr();
    if (!p || !n_out) return fail(OKGE_ERR_INVALID, "no id error word");
    int v = 0;
    hipError_t e = hipMemcpy(&v, p, sizeof(int), hipMemcpyDeviceToHost);       // synchronises: call it when you would sync anyway
    if (e != hipSuccess) return fail_hip(e, "read id error word");
    if (v) {
        e = hipMemset(p, 0, sizeof(int));
        if (e != hipSuccess) return fail_hip(e, "clear id error word");
    }
    *n_out = v;
    return OKGE_OK;
}

int okge_clip_grad_norm(float *g0, int64_t n0, float *g1, int64_t n1, float max_norm, double *norm_out_dev, void *workspace,
                        size_t workspace_bytes, void *stream)
{
    if (!g0 || n0 < 0 || n1 < 0 || (n1 > 0 && !g1) || !(max_norm > 0.f)) return fail(OKGE_ERR_INVALID, "bad clip_grad_norm arguments");
    constexpr int NP = 1024;
    if (!workspace || workspace_bytes < NP * sizeof(double) + 256) return fail(OKGE_ERR_WORKSPACE, "workspace too small (8448 bytes)");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double *partial = static_cast<double *>(workspace);
    float *coef = reinterpret_cast<float *>(static_cast<char *>(workspace) + NP * sizeof(double));
    hipError_t e;
    {
        ScopedTimer tm("clip_grad_norm", st);
        e = launch_clip_coef(g0, n0, g1 ? g1 : g0, g1 ? n1 : 0, max_norm, partial, NP, coef, norm_out_dev, st);
        if (e != hipSuccess) return fail_hip(e, "clip_coef");
        e = launch_scale(g0, n0, coef, st);
        if (e == hipSuccess && g1 && n1 > 0) e = launch_scale(g1, n1, coef, st);
    }
    if (e != hipSuccess) return fail_hip(e, "scale gradients");
    return OKGE_OK;
}

int okge_merge_logsumexp(const float *parts, int32_t world, int32_t B, float *out, void *stream)
{
    if (!parts || !out || world <= 0 || B <= 0) return fail(OKGE_ERR_INVALID, "bad merge_logsumexp arguments");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    ScopedTimer tm("merge_lse", st);
    hipError_t e = launch_merge_lse(parts, world, B, out, st);
    if (e != hipSuccess) return fail_hip(e, "merge_lse");
    return OKGE_OK;
}

int okge_timing_enable(int32_t on)
{
    std::lock_guard<std::mutex> lk(g_tmu);
    g_timing = on != 0;
    return OKGE_OK;
}

int okge_timing_reset(void)
{
    std::lock_guard<std::mutex> lk(g_tmu);
    for (auto &t : g_launches) {
        g_event_pool.push_back(t.start);
        g_event_pool.push_back(t.stop);
    }
    g_launches.clear();
    return OKGE_OK;
}

int okge_timing_collect(const char **names, double *total_ms, int64_t *launches, int32_t cap)
{
    std::lock_guard<std::mutex> lk(g_tmu);
    int n = 0;
    for (auto &t : g_launches) {
        if (hipEventSynchronize(t.stop) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.start, t.stop) != hipSuccess) continue;
        int i = 0;
        for (; i < n; ++i)
            if (names[i] == t.name) break;
        if (i == n) {
            if (n >= cap) continue;
            names[n] = t.name; total_ms[n] = 0; launches[n] = 0;
            ++n;
        }
        total_ms[i] += ms;
        launches[i] += 1;
    }
    return n;
}

}  // extern "C"
