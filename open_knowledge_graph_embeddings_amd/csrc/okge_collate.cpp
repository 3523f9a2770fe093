// Batch producer, HOST side of the C ABI (include/okge.h: okge_collate_batch).
// Replaces OneToNMentionRelationDataset_collate_func (openkge/dataset.py:724-940) and the packed answer-group
// decoding it relies on (utils/misc.py:72-89): from the (P,7) int32 prefix table
//     [a, b, this_start, this_end, all_start, all_end, slot]        (dataset.py:677-703; slot 0 = po, 2 = sp)
// straight to what the device path consumes -- prefix id columns (po rows first), positives as unique (col,row)
// coordinates sorted by column, the candidate id list (batch-shared mode), and for evaluation the answer groups
// and the all-splits filter as CSR.  No dense (B,N) label / filter tensor is ever built (30 MB per batch at the
// FB15k-237 shape in the reference).  Plain C++: integer work on a few thousand ids per batch.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/okge.h"
#include "okge_kernels.h"

namespace {

inline uint64_t splitmix64(uint64_t &s)
{
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// One packed slice (utils/misc.py:56-89): header of slice-relative positions, terminated by 0, then the ids.
struct Packed {
    const int32_t *p;
    int64_t len;
    int64_t n_header;          // header entries before the 0 (= groups + 1)
    bool decode()
    {
        n_header = 0;
        while (n_header < len && p[n_header] != 0) ++n_header;
        if (n_header == len || n_header < 1) return false;             // no terminator / empty header
        int64_t prev = p[0];
        if (prev != n_header + 1) return false;                        // ids start right after the 0
        for (int64_t i = 1; i < n_header; ++i) {
            if (p[i] < prev || p[i] > len) return false;
            prev = p[i];
        }
        return true;
    }
    int64_t groups() const { return n_header - 1; }
    int64_t flat_begin() const { return p[0]; }
    int64_t flat_end() const { return n_header > 1 ? p[n_header - 1] : p[0]; }
};

// entity id -> column map without hashing: one slot per vocabulary entry, validated by an epoch stamp so that a new
// batch costs nothing to "clear" (8 bytes x |E| per thread: 20 MB at the OLPBENCH vocabulary)
struct EntityMap {
    std::vector<uint32_t> stamp;
    std::vector<int32_t> col;
    uint32_t epoch = 0;
    void begin(size_t n)
    {
        if (stamp.size() < n) { stamp.assign(n, 0); col.resize(n); epoch = 0; }
        if (++epoch == 0) { std::fill(stamp.begin(), stamp.end(), 0); epoch = 1; }
    }
    bool has(int32_t e) const { return stamp[(size_t)e] == epoch; }
    int32_t at(int32_t e) const { return col[(size_t)e]; }
    void put(int32_t e, int32_t c) { stamp[(size_t)e] = epoch; col[(size_t)e] = c; }
};

thread_local EntityMap tl_index, tl_drawn;

}  // namespace

extern "C" int okge_collate_batch(const okge_prefix_table *t, const int64_t *rows, int32_t B, int32_t is_training,
                                  int32_t use_batch_shared, int32_t min_size_batch_labels, uint64_t seed,
                                  okge_collated *out)
{
    using okge::report_error;
    if (!t || !rows || !out || B <= 0) return report_error(OKGE_ERR_INVALID, "collate: null argument or empty batch");
    if (!t->prefixes || !t->seen_entities || (!is_training && !t->all_splits_entities && t->n_all > 0))
        return report_error(OKGE_ERR_INVALID, "collate: null prefix table");
    if (t->entity_offset < 0 || t->n_entities <= t->entity_offset)
        return report_error(OKGE_ERR_INVALID, "collate: bad entity vocabulary");
    const bool training = is_training != 0, shared = use_batch_shared != 0;

    struct Item { int32_t a, b; Packed pk; const int32_t *all; int64_t n_all; };
    std::vector<Item> items[2];                       // [0] = slot 0 (po), [1] = slot 2 (sp); batch order kept
    EntityMap &index = tl_index;                      // batch-shared: entity id -> column, first-seen order
    if (shared) index.begin((size_t)t->n_entities);
    std::vector<int32_t> cand;
    bool id_ok = true;
    auto see = [&](int32_t e) {
        if (e < 0 || e >= t->n_entities) { id_ok = false; return; }
        if (!index.has(e)) { index.put(e, (int32_t)cand.size()); cand.push_back(e); }
    };
    for (int32_t i = 0; i < B; ++i) {
        const int64_t r = rows[i];
        if (r < 0 || r >= t->n_prefixes) return report_error(OKGE_ERR_INVALID, "collate: prefix row out of range");
        const int32_t *f = t->prefixes + 7 * r;
        const int32_t slot = f[6];
        if (slot != 0 && slot != 2) return report_error(OKGE_ERR_INVALID, "collate: slot must be 0 (po) or 2 (sp)");
        Item it;
        it.a = f[0]; it.b = f[1];
        if (f[2] < 0 || f[3] < f[2] || f[3] > t->n_seen) return report_error(OKGE_ERR_INVALID, "collate: bad answer slice");
        it.pk.p = t->seen_entities + f[2];
        it.pk.len = f[3] - f[2];
        if (!it.pk.decode()) return report_error(OKGE_ERR_INVALID, "collate: malformed packed answer groups");
        if (f[4] < 0 || f[5] < f[4] || f[5] > t->n_all) return report_error(OKGE_ERR_INVALID, "collate: bad all-splits slice");
        it.all = t->all_splits_entities ? t->all_splits_entities + f[4] : nullptr;
        it.n_all = f[5] - f[4];
        if (shared) {
            if (training) for (int64_t j = it.pk.flat_begin(); j < it.pk.flat_end(); ++j) see(it.pk.p[j]);
            else          for (int64_t j = 0; j < it.n_all; ++j) see(it.all[j]);
        }
        items[slot == 0 ? 0 : 1].push_back(it);
    }
    if (!id_ok) return report_error(OKGE_ERR_INVALID, "collate: entity id outside the vocabulary");

    // ---- candidate list (dataset.py:826-876) --------------------------------------------------------------------
    int64_t N;
    if (shared) {
        const int64_t m = min_size_batch_labels > 0 ? min_size_batch_labels : 0;
        if ((int64_t)cand.size() < m) {
            // fill up with entities sampled without replacement from [offset, n_entities), seen ones removed,
            // truncated to m (the reference draws them with numpy.random.choice and appends them in the
            // iteration order of a Python set -- an implementation detail; here: in sampling order)
            const int64_t range = (int64_t)t->n_entities - t->entity_offset;
            if (m > range) return report_error(OKGE_ERR_INVALID, "collate: min_size_batch_labels exceeds the vocabulary");
            EntityMap &drawn = tl_drawn;
            drawn.begin((size_t)t->n_entities);
            uint64_t s = seed;
            const size_t n_seen = cand.size();
            for (int64_t n_drawn = 0; n_drawn < m; ) {        // m distinct draws, like numpy.random.choice(replace=False)
                const int32_t e = (int32_t)(splitmix64(s) % (uint64_t)range) + t->entity_offset;
                if (drawn.has(e)) continue;
                drawn.put(e, 0);
                ++n_drawn;
                // not entered in `index`: a fill-up entity is never a label or filter column
                if (!index.has(e) && (int64_t)cand.size() < m) cand.push_back(e);
            }
            (void)n_seen;
        }
        N = (int64_t)cand.size();
    } else {
        N = (int64_t)t->n_entities - t->entity_offset;
    }
    auto column = [&](int32_t e, int64_t &c) -> bool {
        if (shared) {
            if (e < 0 || e >= t->n_entities || !index.has(e)) return false;
            c = index.at(e);
            return true;
        }
        c = (int64_t)e - t->entity_offset;
        return c >= 0 && c < N;
    };

    // ---- rows: po first, then sp (dataset.py:884-932) ------------------------------------------------------------
    const int32_t n_po = (int32_t)items[0].size(), n_sp = (int32_t)items[1].size();
    std::vector<uint64_t> pos;                        // (col << 32) | row
    std::vector<int64_t> row_ptr{0}, grp_ptr{0}, filt_ptr{0};
    std::vector<int32_t> ids, filt;
    std::vector<int32_t> tmp;
    int32_t r = 0;
    for (int s = 0; s < 2; ++s) {
        for (const Item &it : items[s]) {
            for (int64_t j = it.pk.flat_begin(); j < it.pk.flat_end(); ++j) {
                int64_t c;
                if (!column(it.pk.p[j], c)) return report_error(OKGE_ERR_INVALID, "collate: answer id outside the candidate list");
                pos.push_back(((uint64_t)c << 32) | (uint32_t)r);
            }
            if (!training) {
                for (int64_t g = 0; g < it.pk.groups(); ++g) {
                    for (int64_t j = it.pk.p[g]; j < it.pk.p[g + 1]; ++j) {
                        int64_t c;
                        if (!column(it.pk.p[j], c)) return report_error(OKGE_ERR_INVALID, "collate: answer id outside the candidate list");
                        ids.push_back((int32_t)c);
                    }
                    grp_ptr.push_back((int64_t)ids.size());
                }
                row_ptr.push_back((int64_t)grp_ptr.size() - 1);
                tmp.clear();
                for (int64_t j = 0; j < it.n_all; ++j) {
                    int64_t c;
                    if (!column(it.all[j], c)) return report_error(OKGE_ERR_INVALID, "collate: filter id outside the candidate list");
                    tmp.push_back((int32_t)c);
                }
                std::sort(tmp.begin(), tmp.end());
                tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());     // a mask: each position once
                filt.insert(filt.end(), tmp.begin(), tmp.end());
                filt_ptr.push_back((int64_t)filt.size());
            }
            ++r;
        }
    }
    std::sort(pos.begin(), pos.end());
    pos.erase(std::unique(pos.begin(), pos.end()), pos.end());                 // labels are a set (label_tensor[..] = 1)

    out->n_po = n_po; out->n_sp = n_sp;
    out->nnz = (int64_t)pos.size();
    out->n_cand = N;
    out->n_groups = (int64_t)grp_ptr.size() - 1;
    out->n_ids = (int64_t)ids.size();
    out->n_filter = (int64_t)filt.size();
    out->normalizer_loss = (double)B * (double)N;                              // dataset.py:935
    out->normalizer_metric = (double)pos.size();                               // dataset.py:934: label_tensor.sum()
    if (out->cap_rows < B || out->cap_pos < out->nnz || (shared && out->cap_cand < N) ||
        (!training && (out->cap_groups < out->n_groups || out->cap_ids < out->n_ids || out->cap_filter < out->n_filter)))
        return report_error(OKGE_ERR_WORKSPACE, "collate: output buffers too small (needed sizes are in the descriptor)");
    if (!out->po_rel || !out->po_obj || !out->sp_subj || !out->sp_rel || !out->pos_row || !out->pos_col ||
        (shared && !out->cand_ids) ||
        (!training && (!out->row_ptr || !out->grp_ptr || !out->ids || !out->filt_ptr || !out->filt_col)))
        return report_error(OKGE_ERR_INVALID, "collate: null output buffer");
    for (int32_t i = 0; i < n_po; ++i) { out->po_rel[i] = items[0][i].a; out->po_obj[i] = items[0][i].b; }
    for (int32_t i = 0; i < n_sp; ++i) { out->sp_subj[i] = items[1][i].a; out->sp_rel[i] = items[1][i].b; }
    for (size_t i = 0; i < pos.size(); ++i) {
        out->pos_col[i] = (int32_t)(pos[i] >> 32);
        out->pos_row[i] = (int32_t)(pos[i] & 0xffffffffu);
    }
    if (shared) std::memcpy(out->cand_ids, cand.data(), sizeof(int32_t) * cand.size());
    if (!training) {
        std::memcpy(out->row_ptr, row_ptr.data(), sizeof(int64_t) * row_ptr.size());
        std::memcpy(out->grp_ptr, grp_ptr.data(), sizeof(int64_t) * grp_ptr.size());
        std::memcpy(out->ids, ids.data(), sizeof(int32_t) * ids.size());
        std::memcpy(out->filt_ptr, filt_ptr.data(), sizeof(int64_t) * filt_ptr.size());
        std::memcpy(out->filt_col, filt.data(), sizeof(int32_t) * filt.size());
    }
    return OKGE_OK;
}


// Several batches in one call, all arrays of all batches in ONE int32 arena (one pinned buffer, one H2D copy): the host
// side of a training loop then costs one library call per n_batches steps instead of one Python round trip per step,
// and the call runs without the interpreter lock.
extern "C" int okge_collate_batches(const okge_prefix_table *t, const int64_t *rows, int32_t n_batches, int32_t B,
                                    int32_t is_training, int32_t use_batch_shared, int32_t min_size_batch_labels,
                                    uint64_t seed, int32_t *arena, int64_t arena_cap, okge_arena_batch *layout,
                                    int64_t *arena_used)
{
    using okge::report_error;
    if (!t || !rows || !arena || !layout || !arena_used || n_batches <= 0 || B <= 0)
        return report_error(OKGE_ERR_INVALID, "collate: null argument or empty request");
    const bool training = is_training != 0, shared = use_batch_shared != 0;
    int64_t used = 0;
    auto carve = [&](int64_t n) {                      // n int32 elements, start kept 8-byte aligned
        const int64_t off = used;
        used += (n + 1) / 2 * 2;
        return off;
    };
    for (int32_t k = 0; k < n_batches; ++k) {
        const int64_t *r = rows + (int64_t)k * B;
        int64_t cap_this = 0, cap_all = 0;
        for (int32_t i = 0; i < B; ++i) {
            if (r[i] < 0 || r[i] >= t->n_prefixes) return report_error(OKGE_ERR_INVALID, "collate: prefix row out of range");
            const int32_t *f = t->prefixes + 7 * r[i];
            cap_this += (int64_t)f[3] - f[2];
            if (!training) cap_all += (int64_t)f[5] - f[4];
        }
        if (cap_this < 0 || cap_all < 0) return report_error(OKGE_ERR_INVALID, "collate: bad slices");
        const int64_t cap_cand = shared ? std::max<int64_t>(min_size_batch_labels > 0 ? min_size_batch_labels : 0,
                                                           training ? cap_this : cap_all) : 0;
        okge_arena_batch &L = layout[k];
        std::memset(&L, 0, sizeof(L));
        okge_collated c;
        std::memset(&c, 0, sizeof(c));
        c.cap_rows = B; c.cap_pos = cap_this; c.cap_cand = cap_cand;
        c.cap_groups = training ? 0 : cap_this; c.cap_ids = training ? 0 : cap_this; c.cap_filter = cap_all;
        // 64-bit arrays first (their int32 offsets are even)
        if (!training) {
            L.off_row_ptr = carve(2 * (B + 1)); L.off_grp_ptr = carve(2 * (cap_this + 1)); L.off_filt_ptr = carve(2 * (B + 1));
        }
        L.off_po_rel = carve(B); L.off_po_obj = carve(B); L.off_sp_subj = carve(B); L.off_sp_rel = carve(B);
        L.off_pos_row = carve(cap_this); L.off_pos_col = carve(cap_this);
        L.off_cand = carve(cap_cand);
        if (!training) { L.off_ids = carve(cap_this); L.off_filt_col = carve(cap_all); }
        if (used > arena_cap) {
            *arena_used = used;
            return report_error(OKGE_ERR_WORKSPACE, "collate: arena too small");
        }
        c.po_rel = arena + L.off_po_rel; c.po_obj = arena + L.off_po_obj;
        c.sp_subj = arena + L.off_sp_subj; c.sp_rel = arena + L.off_sp_rel;
        c.pos_row = arena + L.off_pos_row; c.pos_col = arena + L.off_pos_col;
        c.cand_ids = arena + L.off_cand;
        c.row_ptr = reinterpret_cast<int64_t *>(arena + L.off_row_ptr);
        c.grp_ptr = reinterpret_cast<int64_t *>(arena + L.off_grp_ptr);
        c.filt_ptr = reinterpret_cast<int64_t *>(arena + L.off_filt_ptr);
        c.ids = arena + L.off_ids; c.filt_col = arena + L.off_filt_col;
        if (int rc = okge_collate_batch(t, r, B, is_training, use_batch_shared, min_size_batch_labels,
                                        seed + 0x9E3779B97F4A7C15ull * (uint64_t)(k + 1), &c))
            return rc;
        L.n_po = c.n_po; L.n_sp = c.n_sp; L.nnz = c.nnz; L.n_cand = c.n_cand; L.n_groups = c.n_groups; L.n_ids = c.n_ids;
        L.n_filter = c.n_filter; L.normalizer_loss = c.normalizer_loss; L.normalizer_metric = c.normalizer_metric;
    }
    *arena_used = used;
    return OKGE_OK;
}
