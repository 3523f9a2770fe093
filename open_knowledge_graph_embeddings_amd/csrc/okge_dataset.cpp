// On-disk format -> dataset tensors, HOST side of the C ABI (include/okge.h: okge_dataset_*).
// Replaces OneToNMentionRelationDataset._collect_seen_triples / merge_all_splits_triples / create_data_tensors
// (openkge/dataset.py:480-710), which go text -> sorted jsonl -> merged jsonl -> pickled tensors through Python
// (about 30 min and 10-20 GB for OLPBENCH, README.md:74), by one in-memory pass per file:
//   5-column TSV  s \t p \t o \t subject-mention-ids \t object-mention-ids     (utils/map_dataset_to_ids.py:11-17)
//   -> seen_prefixes [P][7], packed seen_entities, all_splits_entities          (the tensors okge_collate_batch reads)
// Reference behaviour that is kept on purpose (tests/golden/g6_*): the STRING sort keys (dataset.py:497-500), the
// record still open at end of input never being written (:501-518: the last prefix in sort order is dropped), slot
// 0 = po / 2 = sp, sp_o rows before po_s rows, training rows carrying all_start = all_end = 0.
// Declared divergences: ids inside one all-splits slice are ascending (the reference keeps a Python set's iteration
// order); with max_size_prefix_label the tensors end after the last real row (the reference over-allocates and
// leaves an uninitialised tail, :628-640).
#include <algorithm>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <string_view>
#include <utility>
#include <vector>

#include "../../include/okge.h"
#include "okge_kernels.h"

namespace {

struct Record {                       // one prefix of one split, one direction
    int32_t a, b;
    std::vector<std::vector<int32_t>> groups;
};

struct Direction { int key2, key1, c1, c2, slot, ans; };
constexpr Direction kDirs[2] = {{1, 0, 0, 1, 2, 4},      // sp_o   (dataset.py:490)
                                {1, 2, 1, 2, 0, 3}};     // po_s   (dataset.py:491)

bool read_file(const char *path, std::string &out)
{
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
    const bool ok = !std::ferror(f);
    std::fclose(f);
    return ok;
}

bool parse_int(std::string_view s, int32_t &v)
{
    while (!s.empty() && (s.front() == ' ' || s.front() == '\r' || s.front() == '\n')) s.remove_prefix(1);
    while (!s.empty() && (s.back() == ' ' || s.back() == '\r' || s.back() == '\n')) s.remove_suffix(1);
    if (s.empty() || s.size() > 15) return false;
    char tmp[16];
    std::memcpy(tmp, s.data(), s.size());
    tmp[s.size()] = 0;
    char *end = nullptr;
    errno = 0;
    const long x = std::strtol(tmp, &end, 10);
    if (errno || end == tmp || *end != 0 || x < INT32_MIN || x > INT32_MAX) return false;
    v = (int32_t)x;
    return true;
}

struct Line { std::string_view f[5]; };

// text -> lines of 5 tab-separated fields (a trailing empty line is no line, like file.readlines())
int parse_lines(const std::string &text, std::vector<Line> &lines, std::string &err)
{
    size_t pos = 0;
    while (pos < text.size()) {
        size_t nl = text.find('\n', pos);
        if (nl == std::string::npos) nl = text.size();
        std::string_view ln(text.data() + pos, nl - pos);
        pos = nl + 1;
        Line L;
        size_t start = 0;
        int nf = 0;
        for (; nf < 5; ++nf) {
            size_t tab = ln.find('\t', start);
            if (nf < 4 && tab == std::string_view::npos) break;
            if (nf == 4) tab = ln.size();                      // the answer column keeps anything that follows
            L.f[nf] = ln.substr(start, tab - start);
            start = tab + 1;
        }
        if (nf < 5) { err = "dataset: a line has fewer than 5 tab-separated columns"; return OKGE_ERR_INVALID; }
        lines.push_back(L);
    }
    return OKGE_OK;
}

int collect(const std::vector<Line> &lines, const Direction &d, std::vector<Record> &out, std::string &err)
{
    std::vector<const Line *> order(lines.size());
    for (size_t i = 0; i < lines.size(); ++i) order[i] = &lines[i];
    // sorted(sorted(lines, key=relation string), key=entity string): both stable
    std::stable_sort(order.begin(), order.end(), [&](const Line *x, const Line *y) {
        const int c = x->f[d.key1].compare(y->f[d.key1]);
        return c != 0 ? c < 0 : x->f[d.key2].compare(y->f[d.key2]) < 0;
    });
    bool open = false;
    Record cur;
    for (const Line *L : order) {
        int32_t a, b;
        if (!parse_int(L->f[d.c1], a) || !parse_int(L->f[d.c2], b)) { err = "dataset: non-integer id column"; return OKGE_ERR_INVALID; }
        std::vector<int32_t> group;
        std::string_view m = L->f[d.ans];
        size_t i = 0;
        while (i < m.size()) {
            while (i < m.size() && std::isspace((unsigned char)m[i])) ++i;
            size_t j = i;
            while (j < m.size() && !std::isspace((unsigned char)m[j])) ++j;
            if (j > i) {
                int32_t v;
                if (!parse_int(m.substr(i, j - i), v)) { err = "dataset: non-integer mention id"; return OKGE_ERR_INVALID; }
                group.push_back(v);
            }
            i = j;
        }
        if (open && cur.a == a && cur.b == b) {
            cur.groups.push_back(std::move(group));
        } else {
            if (open) out.push_back(std::move(cur));
            cur = Record{a, b, {}};
            cur.groups.push_back(std::move(group));
            open = true;
        }
    }
    // the record still open here is dropped, as in the reference (no flush after its loop)
    return OKGE_OK;
}

struct SplitTensors { std::vector<int32_t> prefixes, seen; };

}  // namespace

struct okge_dataset {
    SplitTensors split[3];
    std::vector<int32_t> all_splits;
    int32_t max_entity_id = -1, max_relation_id = -1;
};

extern "C" {

int okge_dataset_open(const char *train_path, const char *valid_path, const char *test_path,
                      int32_t max_size_prefix_label, okge_dataset **out)
{
    using okge::report_error;
    if (!train_path || !valid_path || !test_path || !out) return report_error(OKGE_ERR_INVALID, "dataset: null argument");
    const char *paths[3] = {train_path, valid_path, test_path};
    std::vector<Record> rec[3][2];
    std::string err;
    for (int s = 0; s < 3; ++s) {
        std::string text;
        if (!read_file(paths[s], text)) return report_error(OKGE_ERR_INVALID, std::string("dataset: cannot read ") + paths[s]);
        std::vector<Line> lines;
        if (int rc = parse_lines(text, lines, err)) return report_error(rc, err + " (" + paths[s] + ")");
        for (int d = 0; d < 2; ++d)
            if (int rc = collect(lines, kDirs[d], rec[s][d], err)) return report_error(rc, err + " (" + paths[s] + ")");
    }
    auto *ds = new okge_dataset();
    // all splits merged per direction (dataset.py:520-565), sp_o slices first (dataset.py:590-607)
    std::map<std::pair<int32_t, int32_t>, std::pair<int64_t, int64_t>> coords[2];
    for (int d = 0; d < 2; ++d) {
        std::map<std::pair<int32_t, int32_t>, std::vector<int32_t>> merged;      // ordered by prefix, numerically
        for (int s = 0; s < 3; ++s)
            for (const Record &r : rec[s][d]) {
                auto &v = merged[{r.a, r.b}];
                for (const auto &g : r.groups) v.insert(v.end(), g.begin(), g.end());
            }
        for (auto &kv : merged) {
            auto &v = kv.second;
            std::sort(v.begin(), v.end());
            v.erase(std::unique(v.begin(), v.end()), v.end());
            coords[d][kv.first] = {(int64_t)ds->all_splits.size(), (int64_t)(ds->all_splits.size() + v.size())};
            ds->all_splits.insert(ds->all_splits.end(), v.begin(), v.end());
        }
    }
    // per split: prefix rows + packed answer groups (dataset.py:609-706); split 0 is the training split
    for (int s = 0; s < 3; ++s) {
        const bool training = s == 0;
        SplitTensors &t = ds->split[s];
        for (int d = 0; d < 2; ++d)
            for (const Record &r : rec[s][d]) {
                const int64_t k = (int64_t)r.groups.size();
                const bool chunked = training && max_size_prefix_label > 1 && k > max_size_prefix_label;
                const int64_t step = chunked ? max_size_prefix_label : std::max<int64_t>(1, k);
                for (int64_t off = 0; off < k; off += step) {
                    const int64_t kk = std::min(step, k - off);
                    const int64_t start = (int64_t)t.seen.size();
                    // packed groups (utils/misc.py:56-70): [b_0+L .. b_kk+L, 0, ids...], L = kk + 2
                    int64_t cum = 0;
                    t.seen.push_back((int32_t)(kk + 2));
                    for (int64_t g = 0; g < kk; ++g) {
                        cum += (int64_t)r.groups[off + g].size();
                        t.seen.push_back((int32_t)(cum + kk + 2));
                    }
                    t.seen.push_back(0);
                    for (int64_t g = 0; g < kk; ++g)
                        for (int32_t e : r.groups[off + g]) {
                            t.seen.push_back(e);
                            ds->max_entity_id = std::max(ds->max_entity_id, e);
                        }
                    const auto c = training ? std::pair<int64_t, int64_t>{0, 0} : coords[d][{r.a, r.b}];
                    if (t.seen.size() > (size_t)INT32_MAX || c.second > INT32_MAX) {
                        delete ds;
                        return report_error(OKGE_ERR_UNSUPPORTED, "dataset: more than 2^31 packed ids (int32 offsets, as in the reference)");
                    }
                    const int32_t row[7] = {r.a, r.b, (int32_t)start, (int32_t)t.seen.size(), (int32_t)c.first,
                                            (int32_t)c.second, kDirs[d].slot};
                    t.prefixes.insert(t.prefixes.end(), row, row + 7);
                    const int32_t ent = d == 0 ? r.a : r.b, rel = d == 0 ? r.b : r.a;
                    ds->max_entity_id = std::max(ds->max_entity_id, ent);
                    ds->max_relation_id = std::max(ds->max_relation_id, rel);
                }
            }
    }
    *out = ds;
    return OKGE_OK;
}

int okge_dataset_sizes(const okge_dataset *ds, int32_t split, int64_t *n_prefixes, int64_t *n_seen, int64_t *n_all,
                       int32_t *max_entity_id, int32_t *max_relation_id)
{
    if (!ds || split < 0 || split > 2) return okge::report_error(OKGE_ERR_INVALID, "dataset: bad handle or split");
    if (n_prefixes) *n_prefixes = (int64_t)ds->split[split].prefixes.size() / 7;
    if (n_seen) *n_seen = (int64_t)ds->split[split].seen.size();
    if (n_all) *n_all = (int64_t)ds->all_splits.size();
    if (max_entity_id) *max_entity_id = ds->max_entity_id;
    if (max_relation_id) *max_relation_id = ds->max_relation_id;
    return OKGE_OK;
}

int okge_dataset_copy(const okge_dataset *ds, int32_t split, int32_t *prefixes, int32_t *seen_entities,
                      int32_t *all_splits_entities)
{
    if (!ds || split < 0 || split > 2) return okge::report_error(OKGE_ERR_INVALID, "dataset: bad handle or split");
    const SplitTensors &t = ds->split[split];
    if (prefixes && !t.prefixes.empty()) std::memcpy(prefixes, t.prefixes.data(), sizeof(int32_t) * t.prefixes.size());
    if (seen_entities && !t.seen.empty()) std::memcpy(seen_entities, t.seen.data(), sizeof(int32_t) * t.seen.size());
    if (all_splits_entities && !ds->all_splits.empty())
        std::memcpy(all_splits_entities, ds->all_splits.data(), sizeof(int32_t) * ds->all_splits.size());
    return OKGE_OK;
}

void okge_dataset_close(okge_dataset *ds) { delete ds; }

}  // extern "C"
