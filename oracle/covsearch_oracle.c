/*
 * covsearch_oracle.c -- CPU oracle for segment_juncs' coverage search.
 *
 * TEST INFRASTRUCTURE ONLY (see thj_oracle.h).  Plain-C restatement, on ASCII data and byte-at-a-time loops, of
 * DaehwanKimLab/tophat v2.1.2 src/segment_juncs.cpp:
 *   MerExtension, store_read_extensions, index_read_mers    :146-180, :240-360, :396-571   (10-mer extension table of
 *                                                            the first 32 bases of the initially unmapped reads)
 *   build_coverage_map                                       :4140-4176
 *   capture_island_ends                                      :4268-4543   (islands, look-left / look-right windows)
 *   juncs_from_ref_segs<RecordExtendableJuncs>               :2052-2377   (POINT_DIR_LEFT / POINT_DIR_RIGHT windows)
 *   IntronMotifs::unique / attach_mers                       :700-833
 *   junction_key, left/right_extendable_junction,
 *   extendable_junction, mismatching_bases                   :576-648, :1464-1566
 *   RecordExtendableJuncs::record                            :1568-1626   (incl. the max_cov_juncs cap, :56)
 * Colour-space branches are out of scope and omitted.
 *
 * PARITY: unpinned by the reference's own tests (none reach the coverage search); checked informally against the
 * survey-stage scratch build (oracle/README.md).
 */
#define _POSIX_C_SOURCE 200809L          /* strdup */
#include "thj_oracle.h"
#include <stdlib.h>
#include <string.h>

/* charToDna5 & 3 (segment_juncs.cpp:166-211, dna5str_to_idx :218-229): A/a 0, C/c 1, G/g 2, T/t 3, N and anything else 0 */
static inline uint32_t base2(char c) {
    switch (c) { case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 0; }
}

/* ---- the extension table (:146-180, :240-360) ----------------------------------------------------------- */
#define MAX_EXTENSION_BP 14
typedef struct { uint32_t left_str, right_str; uint8_t left_len, right_len; } mer_ext;
typedef struct { int64_t* off; mer_ext* ext; } mer_table;      /* CSR over the 4^10 keys */
#define N_KEYS (1u << 20)

/* one pass of store_read_extensions (:240-360) over the read's first 32 bases; emit(key, ext) per seed position.
 * seq_key_len = 5: the seed is 10 bases; `left` are the bases before it, `right` the bases after it. */
static void read_extensions(const char* seq, int len, int64_t* counts, mer_table* t) {
    if (len > 32) len = 32;                                   /* count_read_mers / store_read_mers :425, :520 */
    if (len < 10) return;                                     /* the reference reads past the string here: undefined */
    for (int i = 0; i + 10 <= len; ++i) {
        uint32_t seed = 0;
        for (int k = 0; k < 10; ++k) seed = (seed << 2) | base2(seq[i + k]);
        if (!t) { counts[seed]++; continue; }
        mer_ext e;
        /* right: the min(len - 10 - i, 14) bases following the seed, first base most significant (:281-306) */
        int rl = len - 10 - i; if (rl > MAX_EXTENSION_BP) rl = MAX_EXTENSION_BP;
        uint32_t r = 0;
        for (int k = 0; k < rl; ++k) r = (r << 2) | base2(seq[i + 10 + k]);
        /* left: the low 28 bits of the bases before the seed = the last min(i, 14) of them (:295, :149) */
        int ll = i < MAX_EXTENSION_BP ? i : MAX_EXTENSION_BP;
        uint32_t l = 0;
        for (int k = i - ll; k < i; ++k) l = (l << 2) | base2(seq[k]);
        e.left_str = l & 0x0FFFFFFFu; e.right_str = r & 0x0FFFFFFFu; e.left_len = (uint8_t)ll; e.right_len = (uint8_t)rl;
        t->ext[t->off[seed] + counts[seed]++] = e;
    }
}

/* junction_key (:1506-1518) */
static uint32_t junction_key(uint64_t up, uint64_t down) {
    uint64_t up_half = up & ~(0xFFFFFFFFFFFFFFFFull << 10);
    uint64_t down_half = (down & ~(0xFFFFFFFFFFFFFFFFull >> 10)) >> (64 - 10);
    return ((uint32_t)up_half << 10) | (uint32_t)down_half;
}
/* mismatching_bases (:598-648) with max_mis = extension_mismatches = 0 (:1464): > 0 iff the low `len` bases differ */
static int ext_mismatch(uint32_t w1, uint32_t w2, int len) {
    uint32_t x = w1 ^ w2;
    if (!x) return 0;
    int bit = __builtin_ctz(x);
    return (bit >> 1) < len;
}
/* extendable_junction (:1520-1566) = left_extendable_junction || right_extendable_junction (:1467-1504), min_ext_len 7 */
static int extendable_junction(const mer_table* t, uint64_t up, uint64_t down) {
    uint32_t key = junction_key(up, down);
    up >>= 10; down <<= 10;
    for (int64_t k = t->off[key]; k < t->off[key + 1]; ++k) {
        const mer_ext* e = &t->ext[k];
        if (e->left_len < 7) continue;
        uint64_t u = up & ~(0xFFFFFFFFFFFFFFFFull << (e->left_len << 1));
        if (!ext_mismatch(e->left_str, (uint32_t)u, e->left_len)) return 1;
    }
    for (int64_t k = t->off[key]; k < t->off[key + 1]; ++k) {
        const mer_ext* e = &t->ext[k];
        if (e->right_len < 7) continue;
        uint64_t d = down & ~(0xFFFFFFFFFFFFFFFFull >> (e->right_len << 1));
        d >>= ((32 - e->right_len) << 1);
        if (!ext_mismatch(e->right_str, (uint32_t)d, e->right_len)) return 1;
    }
    return 0;
}
/* rc_dna_str (:650-661) */
static uint64_t rc_dna_str(uint64_t s) {
    s = ~s;
    uint64_t rc = 0;
    for (int i = 0; i < 32; ++i) { rc = (rc << 2) | (s & 3); s >>= 2; }
    return rc;
}

/* ---- sites -------------------------------------------------------------------------------------------------- */
typedef struct { int64_t pos; uint64_t fwd, rev; } site;
typedef struct { site* a; int64_t n, cap; } site_vec;
static void sv_push(site_vec* v, int64_t pos) {
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 256; v->a = (site*)realloc(v->a, (size_t)v->cap * sizeof(site)); }
    v->a[v->n].pos = pos; v->a[v->n].fwd = 0; v->a[v->n].rev = 0; v->n++;
}
static int site_cmp(const void* a, const void* b) { int64_t x = ((const site*)a)->pos, y = ((const site*)b)->pos; return x < y ? -1 : x > y; }
static void sv_unique(site_vec* v) {                          /* IntronMotifs::unique (:712-724); strings are still (0,0) */
    if (v->n == 0) return;
    qsort(v->a, (size_t)v->n, sizeof(site), site_cmp);
    int64_t w = 1;
    for (int64_t i = 1; i < v->n; ++i) if (v->a[i].pos != v->a[w - 1].pos) v->a[w++] = v->a[i];
    v->n = w;
}
static uint64_t mer32(const char* s) { uint64_t x = 0; for (int i = 0; i < 32; ++i) x = (x << 2) | base2(s[i]); return x; }
/* attach_upstream_mers / attach_downstream_mers (:741-833): sites too close to a contig end keep (0, 0) */
static void attach_upstream(const char* ref, int64_t len, site_vec* v) {
    for (int64_t i = 0; i < v->n; ++i) {
        int64_t pos = v->a[i].pos;
        if (pos <= 32 || pos >= len) continue;
        v->a[i].fwd = mer32(ref + pos - 32); v->a[i].rev = rc_dna_str(v->a[i].fwd);
    }
}
static void attach_downstream(const char* ref, int64_t len, site_vec* v) {
    for (int64_t i = 0; i < v->n; ++i) {
        int64_t pos = v->a[i].pos;
        if (pos + 2 + 32 >= len) continue;
        v->a[i].fwd = mer32(ref + pos + 2); v->a[i].rev = rc_dna_str(v->a[i].fwd);
    }
}

typedef struct { uint64_t skip; orc_junction j; } cand;
typedef struct { cand* a; int64_t n, cap; } cand_vec;
static void cv_push(cand_vec* v, uint32_t ref, int64_t left, int64_t right, int anti, uint64_t skip) {
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 1024; v->a = (cand*)realloc(v->a, (size_t)v->cap * sizeof(cand)); }
    cand* c = &v->a[v->n++];
    c->skip = skip; c->j.ref_id = ref; c->j.left = (uint32_t)left; c->j.right = (uint32_t)right; c->j.antisense = (uint32_t)anti;
}
/* RecordExtendableJuncs::record (:1568-1626) */
static void record(const mer_table* t, uint32_t ref, const site_vec* L, const site_vec* R, int antisense, int min_intron, int max_intron, cand_vec* out) {
    int64_t curr_R = 0;
    for (int64_t l = 0; l < L->n; ++l) {
        while (curr_R < R->n && R->a[curr_R].pos < L->a[l].pos + min_intron) curr_R++;
        int64_t max_right = L->a[l].pos + max_intron;
        for (int64_t r = curr_R; r < R->n && R->a[r].pos < max_right; ++r)
            if (extendable_junction(t, L->a[l].fwd, R->a[r].fwd) || extendable_junction(t, R->a[r].rev, L->a[l].rev))
                cv_push(out, ref, L->a[l].pos - 1, R->a[r].pos + 2, antisense, (uint64_t)(r - curr_R));
    }
}
static int junc_cmp(const orc_junction* a, const orc_junction* b) {     /* Junction::operator< (junctions.h:39-57) */
    if (a->ref_id != b->ref_id) return a->ref_id < b->ref_id ? -1 : 1;
    if (a->left != b->left) return a->left < b->left ? -1 : 1;
    if (a->right != b->right) return a->right < b->right ? -1 : 1;
    if (a->antisense != b->antisense) return a->antisense < b->antisense ? -1 : 1;
    return 0;
}
static int cand_cmp(const void* a, const void* b) {                      /* skip_count_lt (junctions.h:72-80) */
    const cand* x = (const cand*)a; const cand* y = (const cand*)b;
    if (x->skip != y->skip) return x->skip < y->skip ? -1 : 1;
    return junc_cmp(&x->j, &y->j);
}
static int junc_cmp_v(const void* a, const void* b) { return junc_cmp((const orc_junction*)a, (const orc_junction*)b); }

typedef struct { uint32_t ref; int64_t left, right; } window;
typedef struct { window* a; int64_t n, cap; } win_vec;
static window* wv_push(win_vec* v, uint32_t ref, int64_t left, int64_t right) {
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 256; v->a = (window*)realloc(v->a, (size_t)v->cap * sizeof(window)); }
    v->a[v->n].ref = ref; v->a[v->n].left = left; v->a[v->n].right = right;
    return &v->a[v->n++];
}

/* index_read_mers (:548-571): count, size, store */
static mer_table index_read_mers(const char* ium_bases, const int64_t* ium_off, int64_t n_ium) {
    mer_table t;
    int64_t* counts = (int64_t*)calloc(N_KEYS, sizeof(int64_t));
    for (int64_t r = 0; r < n_ium; ++r) read_extensions(ium_bases + ium_off[r], (int)(ium_off[r + 1] - ium_off[r]), counts, NULL);
    t.off = (int64_t*)malloc(((size_t)N_KEYS + 1) * sizeof(int64_t));
    t.off[0] = 0;
    for (uint32_t k = 0; k < N_KEYS; ++k) t.off[k + 1] = t.off[k] + counts[k];
    t.ext = (mer_ext*)malloc((size_t)(t.off[N_KEYS] + 1) * sizeof(mer_ext));
    memset(counts, 0, (size_t)N_KEYS * sizeof(int64_t));
    for (int64_t r = 0; r < n_ium; ++r) read_extensions(ium_bases + ium_off[r], (int)(ium_off[r + 1] - ium_off[r]), counts, &t);
    free(counts);
    return t;
}

int orc_coverage_search(const orc_genome* g, const orc_hit* hits, int64_t n_hits,
                        const char* ium_bases, const int64_t* ium_off, int64_t n_ium,
                        int min_cov_length, int min_intron, int max_intron, int64_t max_juncs,
                        orc_junction** out, int64_t* n_out) {
    enum { LOOK_LEFT = 1, LOOK_RIGHT = 2 };
    static const int extend = 45, repeat_tol = 5;              /* :4346-4347 */
    mer_table t = index_read_mers(ium_bases, ium_off, n_ium);

    /* ---- build_coverage_map (:4140-4176) + capture_island_ends (:4268-4543), contigs in increasing ref_id order */
    const int nc = g->n_contigs;
    int64_t* cov_size = (int64_t*)calloc((size_t)nc + 1, sizeof(int64_t));
    for (int64_t h = 0; h < n_hits; ++h) {
        uint32_t ref = hits[h].ref_id;
        if (ref == 0 || ref > (uint32_t)nc) continue;
        int64_t right = (uint32_t)hits[h].right;              /* size_t right_extent = bh.right() */
        if (right >= cov_size[ref]) cov_size[ref] = right + 1;
    }
    win_vec look_left = {0, 0, 0}, look_right = {0, 0, 0};
    for (int ref = 1; ref <= nc; ++ref) {
        /* a contig enters the map when a hit lands on it, even one with right() == 0 (then cov.size() == 1) */
        int has = 0;
        for (int64_t h = 0; h < n_hits && !has; ++h) has = hits[h].ref_id == (uint32_t)ref;
        if (!has) continue;
        const int64_t n = cov_size[ref] ? cov_size[ref] : 0;
        uint8_t* cov = (uint8_t*)calloc((size_t)n + 1, 1);
        for (int64_t h = 0; h < n_hits; ++h)
            if (hits[h].ref_id == (uint32_t)ref)
                for (uint32_t c = (uint32_t)hits[h].left; c < (uint32_t)hits[h].right; ++c) cov[c] = 1;
        uint8_t* long_enough = (uint8_t*)calloc((size_t)n + 1, 1);
        int64_t last_uncovered = 0;
        for (int64_t c = 1; c < n; ++c) {                      /* :4368-4394 */
            if (!cov[c] || c == n - 1) {
                int putative_exon_length = (int)c - (int)last_uncovered;
                if (cov[c - 1] && putative_exon_length >= min_cov_length)
                    for (int64_t l = c; l > last_uncovered; --l) long_enough[l] = 1;
                last_uncovered = c;
            }
        }
        uint8_t* state = (uint8_t*)calloc((size_t)n + 1, 1);
        for (int64_t c = 1; c < n; ++c) {                      /* :4426-4455 */
            if (long_enough[c]) {
                if (!long_enough[c - 1])
                    for (int64_t r = c - extend; r >= 0 && r < c + repeat_tol && r < n; ++r) state[r] |= LOOK_LEFT;
            } else if (long_enough[c - 1])
                for (int64_t l = c - repeat_tol; l >= 0 && l < c + extend && l < n; ++l) state[l] |= LOOK_RIGHT;
        }
        window* cl = NULL; window* cr = NULL;                  /* :4457-4512; indices, not pointers: the vectors may move */
        int64_t il = -1, ir = -1;
        (void)cl; (void)cr;
        for (int64_t c = 1; c < n; ++c) {
            if (state[c] & LOOK_LEFT) {
                if (!(state[c - 1] & LOOK_LEFT)) { wv_push(&look_left, (uint32_t)ref, c, c + 1); il = look_left.n - 1; }
                else if (il >= 0) look_left.a[il].right++;
            } else if (state[c - 1] & LOOK_LEFT) il = -1;
            if (state[c] & LOOK_RIGHT) {
                if (!(state[c - 1] & LOOK_RIGHT)) { wv_push(&look_right, (uint32_t)ref, c, c + 1); ir = look_right.n - 1; }
                else if (ir >= 0) look_right.a[ir].right++;
            } else if (state[c - 1] & LOOK_RIGHT) ir = -1;
        }
        free(cov); free(long_enough); free(state);
    }
    free(cov_size);

    /* ---- juncs_from_ref_segs<RecordExtendableJuncs> (:2052-2377) over look-right then look-left windows, "GT" / "AG" */
    site_vec* fd = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));    /* fwd_donors   GT  (look right) */
    site_vec* ra = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));    /* rev_acceptors CT (look right) */
    site_vec* fa = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));    /* fwd_acceptors AG (look left)  */
    site_vec* rd = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));    /* rev_donors   AC  (look left)  */
    uint8_t* in_map = (uint8_t*)calloc((size_t)nc + 1, 1);
    for (int pass = 0; pass < 2; ++pass) {
        const win_vec* wv = pass == 0 ? &look_right : &look_left;
        for (int64_t w = 0; w < wv->n; ++w) {
            const window* s = &wv->a[w];
            const char* ref = g->seq[s->ref - 1];
            if (!ref) continue;                                /* :2105-2106 */
            in_map[s->ref] = 1;                                /* ims.insert happens before the bounds test (:2140-2143) */
            const int64_t len = g->len[s->ref - 1];
            if (s->left < 0 || s->right >= len - 1) continue;  /* :2154 */
            const int64_t seg_len = s->right - s->left;
            for (int64_t i = 0; i + 2 <= seg_len; ++i) {       /* to = seg_len - 2 (:2171-2175); DnaString folds N to A */
                uint32_t b0 = base2(ref[s->left + i]), b1 = base2(ref[s->left + i + 1]);
                if (pass == 1) {
                    if (b0 == 0 && b1 == 2) sv_push(&fa[s->ref], s->left + i);          /* AG */
                    else if (b0 == 0 && b1 == 1) sv_push(&rd[s->ref], s->left + i);     /* rc(GT) = AC */
                } else {
                    if (b0 == 2 && b1 == 3) sv_push(&fd[s->ref], s->left + i);          /* GT */
                    else if (b0 == 1 && b1 == 3) sv_push(&ra[s->ref], s->left + i);     /* rc(AG) = CT */
                }
            }
        }
    }
    cand_vec cands = {0, 0, 0};
    for (int ref = 1; ref <= nc; ++ref) {
        if (!in_map[ref]) continue;
        const char* rs = g->seq[ref - 1];
        const int64_t len = g->len[ref - 1];
        sv_unique(&fd[ref]); sv_unique(&fa[ref]); sv_unique(&rd[ref]); sv_unique(&ra[ref]);      /* !all_both (:2320-2321) */
        attach_upstream(rs, len, &fd[ref]); attach_upstream(rs, len, &ra[ref]);                /* :726-732 */
        attach_downstream(rs, len, &rd[ref]); attach_downstream(rs, len, &fa[ref]);
        record(&t, (uint32_t)ref, &fd[ref], &fa[ref], 0, min_intron, max_intron, &cands);
        record(&t, (uint32_t)ref, &ra[ref], &rd[ref], 1, min_intron, max_intron, &cands);
    }
    /* the set<Junction, skip_count_lt> with its cap (:1611-1621): what survives is the max_juncs smallest distinct
     * (skip_count, junction) elements; merged into the coordinate-ordered set afterwards (:5027) */
    qsort(cands.a, (size_t)cands.n, sizeof(cand), cand_cmp);
    int64_t m = 0;
    for (int64_t i = 0; i < cands.n; ++i) if (m == 0 || cand_cmp(&cands.a[m - 1], &cands.a[i]) != 0) cands.a[m++] = cands.a[i];
    if (m > max_juncs) m = max_juncs;
    orc_junction* res = (orc_junction*)malloc((size_t)(m + 1) * sizeof(orc_junction));
    for (int64_t i = 0; i < m; ++i) res[i] = cands.a[i].j;
    qsort(res, (size_t)m, sizeof(orc_junction), junc_cmp_v);
    int64_t k = 0;
    for (int64_t i = 0; i < m; ++i) if (k == 0 || junc_cmp(&res[k - 1], &res[i]) != 0) res[k++] = res[i];
    *out = res; *n_out = k;
    for (int ref = 0; ref <= nc; ++ref) { free(fd[ref].a); free(fa[ref].a); free(rd[ref].a); free(ra[ref].a); }
    free(fd); free(fa); free(rd); free(ra); free(in_map); free(cands.a);
    free(look_left.a); free(look_right.a); free(t.off); free(t.ext);
    return 0;
}


/* ================================================================================================ butterfly search
 * segment_juncs.cpp (opt-in, --butterfly-search; tophat.py never passes it):
 *   prune_extension_table(butterfly_overhang = 6) :478-501, compact_extension_table :466-476   (driver :5002-5003)
 *   pair_covered_sites                               :4178-4249   (islands of the coverage map, each widened by 45 bases, as one
 *                                                                 POINT_DIR_LEFT and one POINT_DIR_RIGHT window)
 *   juncs_from_ref_segs<RecordButterflyJuncs>        :2052-2377   ("GT" / "AG", max / min_coverage_intron_length, max_cov_juncs)
 *   ButterflyKey, RecordButterflyJuncs::record       :1698-2049
 * A junction (donor, acceptor) is proposed when some unmapped read's 10-mer seed ends 6 bases short of the donor with the 6 bases
 * after the seed unknown to the genome there, and some read's seed starts 6 bases past the acceptor, and the 6 + 6 bases either read
 * shows across the gap agree ("butterfly": the two 12-base keys meet in the middle).  skip_count = intron length.
 * PARITY: unpinned (no reference test reaches it). */
typedef struct { uint32_t key, pos; } bkey;
typedef struct { bkey* a; int64_t n, cap; } bkey_vec;
static void bk_push(bkey_vec* v, uint32_t pos, uint32_t key) {
    if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 1024; v->a = (bkey*)realloc(v->a, (size_t)v->cap * sizeof(bkey)); }
    v->a[v->n].key = key; v->a[v->n].pos = pos; v->n++;
}
static int bk_cmp(const void* a, const void* b) {               /* ButterflyKey::operator< (:1705-1712): key, then pos */
    const bkey* x = (const bkey*)a; const bkey* y = (const bkey*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    if (x->pos != y->pos) return x->pos < y->pos ? -1 : 1;
    return 0;
}
static void bk_sort_unique(bkey_vec* v) {
    if (v->n == 0) return;
    qsort(v->a, (size_t)v->n, sizeof(bkey), bk_cmp);
    int64_t w = 1;
    for (int64_t i = 1; i < v->n; ++i) if (bk_cmp(&v->a[i], &v->a[w - 1]) != 0) v->a[w++] = v->a[i];
    v->n = w;
}
static int ext_cmp(const void* a, const void* b) {              /* MerExtension::operator< (:158-168) */
    const mer_ext* x = (const mer_ext*)a; const mer_ext* y = (const mer_ext*)b;
    if (x->left_str != y->left_str) return x->left_str < y->left_str ? -1 : 1;
    if (x->left_len != y->left_len) return x->left_len < y->left_len ? -1 : 1;
    if (x->right_str != y->right_str) return x->right_str < y->right_str ? -1 : 1;
    if (x->right_len != y->right_len) return x->right_len < y->right_len ? -1 : 1;
    return 0;
}
/* RecordButterflyJuncs::record (:1742-2049), half_splice_mer_len = 5, colour space omitted */
static void butterfly_record(const mer_table* t, const int64_t* bucket_n, uint32_t ref, const site_vec* all_left, const site_vec* all_right,
                             int antisense, int min_intron, int max_intron, cand_vec* out) {
    const int key_length = 10, ext_len = 6;
    const uint64_t bottom_bit_mask = ~(0xFFFFFFFFFFFFFFFFull << (key_length << 1));
    const uint64_t top_bit_mask = ~(0xFFFFFFFFFFFFFFFFull >> (key_length << 1));
    const uint64_t mask = ~(0xFFFFFFFFFFFFFFFFull << (ext_len << 1));
    if (all_left->n == 0 || all_right->n == 0) return;
    const int64_t last_site = all_left->a[all_left->n - 1].pos > all_right->a[all_right->n - 1].pos ? all_left->a[all_left->n - 1].pos : all_right->a[all_right->n - 1].pos;
    int64_t curr_left = 0, curr_right = 0;
    for (int64_t edge = 0; edge < last_site; edge += max_intron) {
        while (curr_left < all_left->n && all_left->a[curr_left].pos < edge) curr_left++;
        while (curr_right < all_right->n && all_right->a[curr_right].pos < edge) curr_right++;
        bkey_vec lk = {0, 0, 0}, rk = {0, 0, 0};
        for (int64_t L = curr_left; L < all_left->n; ++L) {
            if (!(all_left->a[L].pos < edge + 2 * (int64_t)max_intron)) continue;
            const site* s = &all_left->a[L];
            const uint64_t fwd_up = s->fwd & bottom_bit_mask;
            for (int64_t i = t->off[fwd_up]; i < t->off[fwd_up] + bucket_n[fwd_up]; ++i) {
                const mer_ext* e = &t->ext[i];
                if (e->right_len < ext_len) continue;
                uint64_t key = (uint64_t)e->right_str >> ((e->right_len - ext_len) << 1);
                key |= (fwd_up & mask) << (ext_len << 1);
                bk_push(&lk, (uint32_t)s->pos, (uint32_t)key);
            }
            const uint64_t rev_up = (s->rev & top_bit_mask) >> (64 - (key_length << 1));
            for (int64_t i = t->off[rev_up]; i < t->off[rev_up] + bucket_n[rev_up]; ++i) {
                const mer_ext* e = &t->ext[i];
                if (e->left_len < ext_len) continue;
                uint64_t x = rc_dna_str((uint64_t)e->left_str);
                x >>= 64 - (e->left_len << 1);
                uint64_t key = x >> ((e->left_len - ext_len) << 1);
                key |= (fwd_up & mask) << (ext_len << 1);        /* the reference takes the forward key's bases here too */
                bk_push(&lk, (uint32_t)s->pos, (uint32_t)key);
            }
        }
        bk_sort_unique(&lk);
        for (int64_t R = curr_right; R < all_right->n; ++R) {
            if (!(all_right->a[R].pos < edge + 2 * (int64_t)max_intron)) continue;
            const site* s = &all_right->a[R];
            const uint64_t fwd_down = (s->fwd & top_bit_mask) >> (64 - (key_length << 1));
            for (int64_t i = t->off[fwd_down]; i < t->off[fwd_down] + bucket_n[fwd_down]; ++i) {
                const mer_ext* e = &t->ext[i];
                if (e->left_len < ext_len) continue;
                uint64_t key = ((uint64_t)e->left_str & mask) << (ext_len << 1);
                key |= fwd_down >> ((key_length - ext_len) << 1);
                bk_push(&rk, (uint32_t)s->pos, (uint32_t)key);
            }
            const uint64_t rev_down = s->rev & bottom_bit_mask;
            for (int64_t i = t->off[rev_down]; i < t->off[rev_down] + bucket_n[rev_down]; ++i) {
                const mer_ext* e = &t->ext[i];
                if (e->right_len < ext_len) continue;
                uint64_t x = rc_dna_str((uint64_t)e->right_str);
                x >>= 64 - (e->right_len << 1);
                uint64_t key = x << (ext_len << 1);
                key |= fwd_down >> ((key_length - ext_len) << 1);
                bk_push(&rk, (uint32_t)s->pos, (uint32_t)key);
            }
        }
        bk_sort_unique(&rk);
        int64_t l = 0, r = 0;
        while (l < lk.n && r < rk.n) {
            while (l < lk.n && lk.a[l].key < rk.a[r].key) ++l;
            if (l == lk.n) break;
            while (r < rk.n && rk.a[r].key < lk.a[l].key) ++r;
            if (r == rk.n) break;
            if (rk.a[r].key == lk.a[l].key) {
                const uint32_t k = rk.a[r].key;
                int64_t le = l, re = r;
                while (re < rk.n && rk.a[re].key == k) ++re;
                while (le < lk.n && lk.a[le].key == k) ++le;
                for (int64_t a = l; a < le; ++a)
                    for (int64_t b = r; b < re; ++b) {
                        const int donor = (int)lk.a[a].pos - 1, acceptor = (int)rk.a[b].pos + 2;
                        if (acceptor - donor > min_intron && acceptor - donor < max_intron)
                            cv_push(out, ref, donor, acceptor, antisense, (uint64_t)(acceptor - donor));      /* "just prefer shorter introns" */
                    }
                l = le; r = re;
            }
        }
        free(lk.a); free(rk.a);
    }
}

int orc_butterfly_search(const orc_genome* g, const orc_hit* hits, int64_t n_hits,
                         const char* ium_bases, const int64_t* ium_off, int64_t n_ium,
                         int min_intron, int max_intron, int64_t max_juncs,
                         orc_junction** out, int64_t* n_out) {
    static const int extend = 45, overhang = 6;                /* :4190, butterfly_overhang :61 */
    mer_table t = index_read_mers(ium_bases, ium_off, n_ium);
    /* prune_extension_table(6) (:478-501), then compact_extension_table (:466-476): sort + unique inside every bucket */
    int64_t* bucket_n = (int64_t*)malloc((size_t)N_KEYS * sizeof(int64_t));
    for (uint32_t k = 0; k < N_KEYS; ++k) {
        mer_ext* e = t.ext + t.off[k];
        const int64_t n = t.off[k + 1] - t.off[k];
        const uint32_t m = ~(0xFFFFFFFFu << (overhang << 1));
        for (int64_t j = 0; j < n; ++j) {
            if (e[j].left_len > overhang) { e[j].left_len = overhang; e[j].left_str &= m; }
            if (e[j].right_len > overhang) { e[j].right_str >>= ((e[j].right_len - overhang) << 1); e[j].right_len = overhang; }
        }
        qsort(e, (size_t)n, sizeof(mer_ext), ext_cmp);
        int64_t w = n ? 1 : 0;
        for (int64_t j = 1; j < n; ++j) if (ext_cmp(&e[j], &e[w - 1]) != 0) e[w++] = e[j];
        bucket_n[k] = w;
    }
    /* build_coverage_map (:4140-4176) + the island walk of pair_covered_sites (:4195-4231), contigs in increasing ref_id order */
    const int nc = g->n_contigs;
    site_vec* fd = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));
    site_vec* ra = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));
    site_vec* fa = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));
    site_vec* rd = (site_vec*)calloc((size_t)nc + 1, sizeof(site_vec));
    uint8_t* in_map = (uint8_t*)calloc((size_t)nc + 1, 1);
    for (int ref = 1; ref <= nc; ++ref) {
        int64_t n = 0; int has = 0;
        for (int64_t h = 0; h < n_hits; ++h)
            if (hits[h].ref_id == (uint32_t)ref) { has = 1; const int64_t right = (uint32_t)hits[h].right; if (right >= n) n = right + 1; }
        if (!has) continue;
        uint8_t* cov = (uint8_t*)calloc((size_t)n + 1, 1);
        for (int64_t h = 0; h < n_hits; ++h)
            if (hits[h].ref_id == (uint32_t)ref)
                for (uint32_t c = (uint32_t)hits[h].left; c < (uint32_t)hits[h].right; ++c) cov[c] = 1;
        win_vec wins = {0, 0, 0};
        int64_t island_left_edge = 0;
        for (int64_t c = 1; c < n; ++c) {
            if (cov[c]) {
                if (!cov[c - 1]) { int64_t edge = c - extend; if (edge < 0) edge = 0; island_left_edge = edge; }
            } else if (cov[c - 1]) wv_push(&wins, (uint32_t)ref, island_left_edge, c + extend);      /* once LEFT, once RIGHT */
        }
        free(cov);
        const char* rs = g->seq[ref - 1];
        if (rs) {                                              /* :2105-2106 */
            const int64_t len = g->len[ref - 1];
            for (int64_t w = 0; w < wins.n; ++w) {
                const window* s = &wins.a[w];
                in_map[ref] = 1;
                if (s->left < 0 || s->right >= len - 1) continue;                      /* :2154 */
                const int64_t seg_len = s->right - s->left;
                for (int64_t i = 0; i + 2 <= seg_len; ++i) {
                    const uint32_t b0 = base2(rs[s->left + i]), b1 = base2(rs[s->left + i + 1]);
                    if (b0 == 0 && b1 == 2) sv_push(&fa[ref], s->left + i);            /* POINT_DIR_LEFT: AG */
                    else if (b0 == 0 && b1 == 1) sv_push(&rd[ref], s->left + i);       /*                 AC */
                    if (b0 == 2 && b1 == 3) sv_push(&fd[ref], s->left + i);            /* POINT_DIR_RIGHT: GT */
                    else if (b0 == 1 && b1 == 3) sv_push(&ra[ref], s->left + i);       /*                  CT */
                }
            }
        }
        free(wins.a);
    }
    cand_vec cands = {0, 0, 0};
    for (int ref = 1; ref <= nc; ++ref) {
        if (!in_map[ref]) continue;
        const char* rs = g->seq[ref - 1];
        const int64_t len = g->len[ref - 1];
        sv_unique(&fd[ref]); sv_unique(&fa[ref]); sv_unique(&rd[ref]); sv_unique(&ra[ref]);
        attach_upstream(rs, len, &fd[ref]); attach_upstream(rs, len, &ra[ref]);
        attach_downstream(rs, len, &rd[ref]); attach_downstream(rs, len, &fa[ref]);
        butterfly_record(&t, bucket_n, (uint32_t)ref, &fd[ref], &fa[ref], 0, min_intron, max_intron, &cands);
        butterfly_record(&t, bucket_n, (uint32_t)ref, &ra[ref], &rd[ref], 1, min_intron, max_intron, &cands);
    }
    /* set<Junction, skip_count_lt> with its cap (:2031-2041), then the coordinate-ordered set (:5031) */
    qsort(cands.a, (size_t)cands.n, sizeof(cand), cand_cmp);
    int64_t m = 0;
    for (int64_t i = 0; i < cands.n; ++i) if (m == 0 || cand_cmp(&cands.a[m - 1], &cands.a[i]) != 0) cands.a[m++] = cands.a[i];
    if (m > max_juncs) m = max_juncs;
    orc_junction* res = (orc_junction*)malloc((size_t)(m + 1) * sizeof(orc_junction));
    for (int64_t i = 0; i < m; ++i) res[i] = cands.a[i].j;
    qsort(res, (size_t)m, sizeof(orc_junction), junc_cmp_v);
    int64_t k = 0;
    for (int64_t i = 0; i < m; ++i) if (k == 0 || junc_cmp(&res[k - 1], &res[i]) != 0) res[k++] = res[i];
    *out = res; *n_out = k;
    for (int ref = 0; ref <= nc; ++ref) { free(fd[ref].a); free(fa[ref].a); free(rd[ref].a); free(ra[ref].a); }
    free(fd); free(fa); free(rd); free(ra); free(in_map); free(cands.a); free(bucket_n); free(t.off); free(t.ext);
    return 0;
}

/* ================================================================================================ microexon search
 * segment_juncs.cpp:3880-3941 (window registration inside look_for_hit_group), :3671-3735 (add_to_microexon_windows),
 * :3737-3815 (align_microexon_segs).  A read whose FIRST segment has no hit while every other segment has some may start in a
 * microexon: for each hit of its second segment a window of max_microexon_stretch (2000, :60) bases next to that hit is
 * registered with the read's first segment_length bases (reverse-complemented for antisense hits); overlapping windows merge,
 * pooling their strings.  Each window is then searched like a coverage-search island pair, with an extension table made of ITS
 * strings only: every GT / CT in it is a left site, every AG / AC a right site (juncs_from_ref_segs with a POINT_DIR_DONTCARE
 * and a POINT_DIR_LEFT copy of the window, :3788-3806, :2289-2318), pairs within [min_coverage_intron, 2000) kept when a
 * string extends across them (RecordExtendableJuncs, :1568-1626).  All windows share one skip-count-ordered junction set
 * capped at max_cov_juncs (:5021-5024). */
typedef struct { uint32_t ref; int32_t left, right; int side; char** str; int64_t n_str, cap_str; } mx_window;
typedef struct { mx_window* a; int64_t n, cap; } mx_map;                         /* std::map<RefSeg, vector<string>*>, kept sorted */

static int mx_key_cmp(uint32_t ref, int32_t l, int32_t r, const mx_window* w) {    /* RefSeg::operator< (segments.h:50-58) */
    if (ref != w->ref) return ref < w->ref ? -1 : 1;
    if (l != w->left) return l < w->left ? -1 : 1;
    if (r != w->right) return r < w->right ? -1 : 1;
    return 0;
}
static int64_t mx_lower_bound(const mx_map* m, uint32_t ref, int32_t l, int32_t r) {
    int64_t lo = 0, hi = m->n;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (mx_key_cmp(ref, l, r, &m->a[mid]) > 0) lo = mid + 1; else hi = mid; }
    return lo;
}
static void mx_push_str(mx_window* w, const char* s) {
    if (w->n_str == w->cap_str) { w->cap_str = w->cap_str ? 2 * w->cap_str : 4; w->str = (char**)realloc(w->str, (size_t)w->cap_str * sizeof(char*)); }
    w->str[w->n_str++] = strdup(s);
}
/* map::insert: nothing happens when the key is there already (the reference then leaks the vector it made) */
static void mx_insert(mx_map* m, mx_window w) {
    int64_t at = mx_lower_bound(m, w.ref, w.left, w.right);
    if (at < m->n && mx_key_cmp(w.ref, w.left, w.right, &m->a[at]) == 0) { for (int64_t i = 0; i < w.n_str; ++i) free(w.str[i]); free(w.str); return; }
    if (m->n == m->cap) { m->cap = m->cap ? 2 * m->cap : 64; m->a = (mx_window*)realloc(m->a, (size_t)m->cap * sizeof(mx_window)); }
    memmove(&m->a[at + 1], &m->a[at], (size_t)(m->n - at) * sizeof(mx_window));
    m->a[at] = w; m->n++;
}
static int overlap_in_genome(int ll, int lr, int rl, int rr) {                    /* :3662-3673 */
    if (ll >= rl && ll < rr) return 1;
    if (lr > rl && lr < rr) return 1;
    if (rl >= ll && rl < lr) return 1;
    if (rr > ll && rr < lr) return 1;
    return 0;
}
/* add_to_microexon_windows (:3675-3735), statement for statement */
static void mx_add(mx_map* m, uint32_t ref, int left_boundary, int right_boundary, const char* dna, int side) {
    mx_window nw; memset(&nw, 0, sizeof nw);
    nw.ref = ref; nw.left = left_boundary; nw.right = right_boundary; nw.side = side;          /* left_dummy */
    int64_t lb = mx_lower_bound(m, ref, left_boundary, right_boundary);
    const int64_t ub = mx_lower_bound(m, ref, right_boundary, right_boundary + 1);
    if (lb == m->n) { mx_push_str(&nw, dna); mx_insert(m, nw); return; }
    int64_t first_erased = -1, last_erased = ub;
    int have_new = 0;
    for (; lb < ub; ++lb) {
        if (overlap_in_genome(m->a[lb].left, m->a[lb].right, left_boundary, right_boundary)) {
            have_new = 1;
            if (first_erased < 0) first_erased = lb;
            nw.left = m->a[lb].left < left_boundary ? m->a[lb].left : left_boundary;
            nw.right = m->a[lb].right > right_boundary ? m->a[lb].right : right_boundary;
            for (int64_t i = 0; i < m->a[lb].n_str; ++i) { mx_push_str(&nw, m->a[lb].str[i]); free(m->a[lb].str[i]); }
            free(m->a[lb].str); m->a[lb].str = NULL; m->a[lb].n_str = 0;
        } else if (first_erased >= 0) last_erased = lb;
    }
    if (first_erased >= 0) {
        memmove(&m->a[first_erased], &m->a[last_erased], (size_t)(m->n - last_erased) * sizeof(mx_window));
        m->n -= last_erased - first_erased;
    }
    (void)have_new;
    mx_push_str(&nw, dna);                                   /* a window of its own, or the merged one with the new string last */
    mx_insert(m, nw);
}
static char mx_comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

int orc_microexon_search(const orc_params* p, int min_anchor_len, int min_intron, int64_t max_juncs, const orc_genome* g,
                         const orc_batch* const* batches, const int* sides, int n_batches, orc_junction** out, int64_t* n_out, int64_t* n_windows) {
    static const int max_microexon_stretch = 2000;            /* :60 */
    const int L = p->segment_length;
    const int seq_key_len = min_anchor_len < 6 ? min_anchor_len : 6;   /* :3848 */
    mx_map map = {0, 0, 0};
    char* fwd = (char*)malloc((size_t)L + 1); char* rev = (char*)malloc((size_t)L + 1);
    for (int bi = 0; bi < n_batches; ++bi) {
        const orc_batch* b = batches[bi];
        if (b->nseg < 2) continue;                             /* look_for_hit_group starts at the file before the last */
        for (int r = 0; r < b->n_reads; ++r) {
            const int64_t* so = b->seg_off + (int64_t)r * b->nseg;
            if (so[1] != so[0]) continue;                      /* the leftmost segment has hits */
            int empty_seg = 0;
            for (int h = 1; h < b->nseg; ++h) if (so[h + 1] == so[h]) empty_seg = h;
            if (empty_seg != 0) continue;                      /* :3902-3910 */
            const char* rs = b->bases + b->read_off[r];
            const int rl = (int)(b->read_off[r + 1] - b->read_off[r]);
            const int n = rl < L ? rl : L;                     /* substr(0, segment_length) */
            for (int i = 0; i < n; ++i) { fwd[i] = rs[i]; rev[n - 1 - i] = mx_comp(rs[i]); }
            fwd[n] = rev[n] = 0;
            for (int64_t h = so[1]; h < so[2]; ++h) {          /* hits_for_read[empty_seg + 1] */
                const orc_hit* bh = &b->hits[h];
                if (bh->ref_id == 0 || bh->ref_id > (uint32_t)g->n_contigs || !g->seq[bh->ref_id - 1]) continue;
                const int ref_len = (int)g->len[bh->ref_id - 1];
                int lb, rb;
                if (bh->flags & ORC_HIT_ANTISENSE) {
                    lb = bh->right - min_anchor_len; if (lb < 0) lb = 0;
                    rb = lb + max_microexon_stretch; if (rb > ref_len - 2) rb = ref_len - 2;
                    if (rb - lb < 2 * seq_key_len) continue;
                    mx_add(&map, bh->ref_id, lb, rb, rev, sides[bi]);
                } else {
                    rb = bh->left + min_anchor_len; if (rb > ref_len - 2) rb = ref_len - 2;
                    lb = rb - max_microexon_stretch; if (lb < 0) lb = 0;
                    if (rb - lb < 2 * seq_key_len) continue;
                    mx_add(&map, bh->ref_id, lb, rb, fwd, sides[bi]);
                }
            }
        }
    }
    free(fwd); free(rev);
    if (n_windows) *n_windows = map.n;
    /* ---- align_microexon_segs */
    cand_vec cands = {0, 0, 0};
    int64_t* counts = (int64_t*)calloc(N_KEYS, sizeof(int64_t));
    for (int64_t wi = 0; wi < map.n; ++wi) {
        const mx_window* w = &map.a[wi];
        /* the window's own extension table: store_read_extensions(extensions, 5, 5, s, false) per string (:3775-3780) */
        mer_table t;
        memset(counts, 0, (size_t)N_KEYS * sizeof(int64_t));
        for (int64_t k = 0; k < w->n_str; ++k) read_extensions(w->str[k], (int)strlen(w->str[k]), counts, NULL);
        t.off = (int64_t*)malloc(((size_t)N_KEYS + 1) * sizeof(int64_t));
        t.off[0] = 0;
        for (uint32_t k = 0; k < N_KEYS; ++k) t.off[k + 1] = t.off[k] + counts[k];
        t.ext = (mer_ext*)malloc((size_t)(t.off[N_KEYS] + 1) * sizeof(mer_ext));
        memset(counts, 0, (size_t)N_KEYS * sizeof(int64_t));
        for (int64_t k = 0; k < w->n_str; ++k) read_extensions(w->str[k], (int)strlen(w->str[k]), counts, &t);
        const char* ref = g->seq[w->ref - 1];
        const int64_t len = g->len[w->ref - 1];
        int skip_fwd = 0, skip_rev = 0;                        /* :2112-2138, seg.antisense == false */
        if (p->library_type == 2) { if (w->side == 1) skip_fwd = 1; else if (w->side == 2) skip_rev = 1; }
        if (p->library_type == 3) { if (w->side == 1) skip_rev = 1; else if (w->side == 2) skip_fwd = 1; }
        site_vec fd = {0, 0, 0}, ra = {0, 0, 0}, fa = {0, 0, 0}, rd = {0, 0, 0};
        if (ref && !(w->left < 0 || w->right >= len - 1)) {    /* :2154, for both copies of the window */
            const int64_t seg_len = w->right - w->left;
            for (int64_t i = 0; i + 2 <= seg_len; ++i) {       /* i <= to = seg_len - 2 */
                const uint32_t b0 = base2(ref[w->left + i]), b1 = base2(ref[w->left + i + 1]);
                /* the POINT_DIR_DONTCARE copy takes the "right pointing" branch (:2304-2318): GT, else CT */
                if (b0 == 2 && b1 == 3) { if (!skip_fwd) sv_push(&fd, w->left + i); }
                else if (b0 == 1 && b1 == 3) { if (!skip_rev) sv_push(&ra, w->left + i); }
                /* the POINT_DIR_LEFT copy (:2289-2303): AG, else AC */
                if (b0 == 0 && b1 == 2) { if (!skip_fwd) sv_push(&fa, w->left + i); }
                else if (b0 == 0 && b1 == 1) { if (!skip_rev) sv_push(&rd, w->left + i); }
            }
            sv_unique(&fd); sv_unique(&fa); sv_unique(&rd); sv_unique(&ra);
            attach_upstream(ref, len, &fd); attach_upstream(ref, len, &ra);
            attach_downstream(ref, len, &rd); attach_downstream(ref, len, &fa);
            record(&t, w->ref, &fd, &fa, 0, min_intron, max_microexon_stretch, &cands);
            record(&t, w->ref, &ra, &rd, 1, min_intron, max_microexon_stretch, &cands);
        }
        free(fd.a); free(fa.a); free(rd.a); free(ra.a); free(t.off); free(t.ext);
    }
    free(counts);
    qsort(cands.a, (size_t)cands.n, sizeof(cand), cand_cmp);
    int64_t m = 0;
    for (int64_t i = 0; i < cands.n; ++i) if (m == 0 || cand_cmp(&cands.a[m - 1], &cands.a[i]) != 0) cands.a[m++] = cands.a[i];
    if (m > max_juncs) m = max_juncs;
    orc_junction* res = (orc_junction*)malloc((size_t)(m + 1) * sizeof(orc_junction));
    for (int64_t i = 0; i < m; ++i) res[i] = cands.a[i].j;
    qsort(res, (size_t)m, sizeof(orc_junction), junc_cmp_v);
    int64_t k = 0;
    for (int64_t i = 0; i < m; ++i) if (k == 0 || junc_cmp(&res[k - 1], &res[i]) != 0) res[k++] = res[i];
    *out = res; *n_out = k;
    for (int64_t wi = 0; wi < map.n; ++wi) { for (int64_t i = 0; i < map.a[wi].n_str; ++i) free(map.a[wi].str[i]); free(map.a[wi].str); }
    free(map.a); free(cands.a);
    return 0;
}
