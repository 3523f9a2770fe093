/*
 * CPU ORACLE (plain C part) for the open-KGE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It restates, in scalar C, the integer/bit-exact pieces of the path plus a scalar fp32 scorer:
 *
 *   okge_oracle_filtered_ranks  -- openkge/dataset.py:423-453 (compute_metrics rank rule)
 *   okge_oracle_philox_keep     -- this build's counter-based dropout mask (Philox4x32-10)
 *   okge_oracle_score_prefix    -- openkge/model.py:198-229 (ComplEx) / :268-274 (DistMult), prefix branch
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks these against golden vectors generated
 * from the reference (tests/golden/g1_*, g5_*) and against oracle/kge_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC)
 */
#include <stdint.h>
#include <stddef.h>
#include <math.h>

/* ---- filtered ranks: dataset.py:436-446 -------------------------------------------------------
 * true_g = max_{j in group g} pred[b][j];  pred'[j] = filter[b][j] ? -1e8 : pred[b][j];
 * rank_g = #(pred' > true_g) + (#(pred' == true_g)) / 2            (integer division)
 */
int okge_oracle_filtered_ranks(const float *pred, const uint8_t *filt, int64_t B, int64_t N,
                               const int64_t *row_ptr, const int64_t *grp_ptr, const int32_t *ids,
                               int64_t *ranks)
{
    for (int64_t b = 0; b < B; ++b) {
        const float *p = pred + b * N;
        const uint8_t *f = filt + b * N;
        for (int64_t g = row_ptr[b]; g < row_ptr[b + 1]; ++g) {
            float t = -INFINITY;
            for (int64_t j = grp_ptr[g]; j < grp_ptr[g + 1]; ++j) {
                float v = p[ids[j]];
                if (v > t) t = v;
            }
            int64_t gt = 0, eq = 0;
            for (int64_t n = 0; n < N; ++n) {
                float v = f[n] ? -1e8f : p[n];
                gt += (v > t);
                eq += (v == t);
            }
            ranks[g] = gt + eq / 2;
        }
    }
    return 0;
}

/* ---- Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011) ------------------------------------------ */
static inline void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void okge_oracle_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

/* keep[row][k]: philox(counter = (row_key, k>>3, stream, step), key = seed) gives 4 words = 8 u16;
 * element k uses word (k&7)>>1, low half for even k, high half for odd k; keep <=> u16 >= floor(p*65536) */
int okge_oracle_philox_keep(uint64_t seed, uint32_t stream, uint32_t step, int64_t nrows, int64_t d,
                            double p, const uint32_t *row_keys, uint8_t *keep)
{
    double t = p * 65536.0;
    uint32_t thr = t <= 0 ? 0u : (t >= 65535.0 ? 65535u : (uint32_t)t);
    for (int64_t r = 0; r < nrows; ++r) {
        uint32_t key = row_keys ? row_keys[r] : (uint32_t)r;
        for (int64_t o = 0; o * 8 < d; ++o) {
            uint32_t c[4] = { key, (uint32_t)o, stream, step };
            okge_oracle_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            for (int e = 0; e < 8 && o * 8 + e < d; ++e) {
                uint32_t u16 = (c[e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
                keep[r * d + o * 8 + e] = u16 >= thr;
            }
        }
    }
    return 0;
}

/* ---- scalar prefix scorer: model.py:205-216, :269-272 ------------------------------------------
 * kind 0 = ComplEx, 1 = DistMult; dir 0 = po, 1 = sp.  ent/rel: (b, d) already-encoded prefix rows,
 * cand: (n, d) candidate rows, out: (b, n).  Accumulates the four ComplEx products exactly as the
 * reference writes them (four separate dot products, then + + -), in fp32.
 */
int okge_oracle_score_prefix(int kind, int dir, const float *ent, const float *rel, const float *cand,
                             int64_t b, int64_t n, int64_t d, float *out)
{
    int64_t h = d / 2;
    for (int64_t i = 0; i < b; ++i) {
        const float *e = ent + i * d, *r = rel + i * d;
        for (int64_t j = 0; j < n; ++j) {
            const float *c = cand + j * d;
            if (kind == 1) {
                float acc = 0.f;
                for (int64_t k = 0; k < d; ++k) acc += (e[k] * r[k]) * c[k];
                out[i * n + j] = acc;
            } else {
                float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
                for (int64_t k = 0; k < h; ++k) {
                    float e1 = e[k], e2 = e[h + k], r1 = r[k], r2 = r[h + k], c1 = c[k], c2 = c[h + k];
                    if (dir == 1) { /* sp */
                        t0 += (e1 * r1) * c1; t1 += (e2 * r1) * c2; t2 += (e1 * r2) * c2; t3 += (e2 * r2) * c1;
                    } else {        /* po */
                        t0 += (e1 * r1) * c1; t1 += (e2 * r1) * c2; t2 += (e2 * r2) * c1; t3 += (e1 * r2) * c2;
                    }
                }
                out[i * n + j] = t0 + t1 + t2 - t3;
            }
        }
    }
    return 0;
}
