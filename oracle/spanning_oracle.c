/*
 * spanning_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see thj_oracle.h) for
 * TopHat's long_spanning_reads: stitching per-segment alignments through known
 * junctions / deletions / insertions into one spliced alignment per read.
 *
 * Plain-C restatement of DaehwanKimLab/tophat v2.1.2:
 *   JoinSegmentsWorker::operator()   long_spanning_reads.cpp:2669-2845
 *   join_segments_for_read           :2612-2667
 *   dfs_seg_hits                     :2222-2610   (fusion_search == false paths)
 *   merge_segment_chain              :2101-2220
 *   merge_chain                      :805-2038    (fusion_dir == FUSION_NOTHING paths)
 *   valid_hit                        :2045-2099
 *   BowtieHit::check_editdist_consistency  bwt_map.cpp:2349-2465
 *   bowtie_sam_extra                 bwt_map.cpp:2467-2648
 *   BowtieHit::operator< / ==        bwt_map.h:167-207
 * Colour-space and fusion branches are out of scope and omitted.
 *
 * PARITY: pinned in part by the reference's regression cases (every recorded
 * alignment's strand / POS / CIGAR / NM, tests/golden_ref/); tags and record
 * order are unpinned by the reference's own tests; see oracle/README.md.
 */
#include "thj_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXC 24
#define MAXSEQ 1024   /* bases of a read (a longer one is skipped); the device path takes up to 512 */

typedef struct {
    uint32_t insert_id;        /* 0 = the empty BowtieHit() */
    uint32_t ref_id;
    int left;
    int n;
    uint32_t cig[MAXC];
    int antisense, antisense_splice, end;
    unsigned char mm, ed;
    char seq[MAXSEQ];
    int seq_len;
} BH;

static int bh_right(const BH* h)            /* bwt_map.h:213-243 */
{
    int r = h->left;
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        uint32_t len = ORC_CIG_LEN(h->cig[i]);
        if (op == ORC_MATCH || op == ORC_REF_SKIP || op == ORC_DEL) r += (int)len;
        else if (op == ORC_mATCH || op == ORC_rEF_SKIP || op == ORC_dEL) r -= (int)len;
    }
    return r;
}
static int bh_read_len(const BH* h)         /* bwt_map.h:141-163 */
{
    int len = 0;
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        if (op == ORC_MATCH || op == ORC_mATCH || op == ORC_INS || op == ORC_iNS || op == ORC_SOFT_CLIP)
            len += (int)ORC_CIG_LEN(h->cig[i]);
    }
    return len;
}
static int bh_is_spliced(const BH* h)
{
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        if (op == ORC_REF_SKIP || op == ORC_rEF_SKIP) return 1;
    }
    return 0;
}
static int gap_length(const uint32_t* cig, int n)   /* bwt_map.cpp:32-43 */
{
    int g = 0;
    for (int i = 0; i < n; ++i) {
        int op = ORC_CIG_OP(cig[i]);
        if (op == ORC_INS || op == ORC_iNS || op == ORC_DEL || op == ORC_dEL) g += (int)ORC_CIG_LEN(cig[i]);
    }
    return g;
}
static int is_match_op(int op) { return op == ORC_MATCH || op == ORC_mATCH; }

static char comp(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; }
}
/* char -> Dna5 -> char: anything but ACGT is N */
static char d5(char c) { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N'; }

typedef struct {
    const orc_span_params* p;
    const orc_genome* g;
    const orc_junction* juncs; int64_t n_juncs;
    const orc_ins_in* ins; int64_t n_ins;
} sctx;

static const char* contig(const orc_genome* g, uint32_t ref_id, int64_t* len)
{
    if (ref_id == 0 || (int64_t)ref_id > g->n_contigs) { *len = 0; return NULL; }
    *len = g->len[ref_id - 1];
    return g->seq[ref_id - 1];
}
/* Dna5 of the reference at pos, 'N' outside (seqan::infix clamps; callers stay in range) */
static char refc(const char* ref, int64_t len, int64_t pos) { return (pos < 0 || pos >= len) ? 'N' : ref[pos]; }

/* Junction::operator< (junctions.h:39-57) */
static int jless(uint32_t r1, uint32_t l1, uint32_t rr1, uint32_t a1, const orc_junction* j)
{
    if (r1 != j->ref_id) return r1 < j->ref_id;
    if (l1 != j->left) return l1 < j->left;
    if (rr1 != j->right) return rr1 < j->right;
    return a1 < j->antisense;
}
static int jless_rev(const orc_junction* j, uint32_t r1, uint32_t l1, uint32_t rr1, uint32_t a1)
{
    if (j->ref_id != r1) return j->ref_id < r1;
    if (j->left != l1) return j->left < l1;
    if (j->right != rr1) return j->right < rr1;
    return j->antisense < a1;
}
static int64_t j_upper_bound(const sctx* c, uint32_t r, uint32_t l, uint32_t rr, uint32_t a)
{   /* first element x with key < x */
    int64_t lo = 0, hi = c->n_juncs;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (jless(r, l, rr, a, &c->juncs[m])) hi = m; else lo = m + 1; }
    return lo;
}
static int64_t j_lower_bound(const sctx* c, uint32_t r, uint32_t l, uint32_t rr, uint32_t a)
{   /* first element x with !(x < key) */
    int64_t lo = 0, hi = c->n_juncs;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (jless_rev(&c->juncs[m], r, l, rr, a)) lo = m + 1; else hi = m; }
    return lo;
}
/* Insertion::operator< (insertions.h:52-67): (refid, left, sequence.length()) */
static int64_t i_upper_bound(const sctx* c, uint32_t r, uint32_t l, size_t len)
{
    int64_t lo = 0, hi = c->n_ins;
    while (lo < hi) {
        int64_t m = (lo + hi) / 2;
        const orc_ins_in* x = &c->ins[m];
        int less;   /* key < x ? */
        if (r != x->ref_id) less = r < x->ref_id;
        else if (l != x->left) less = l < x->left;
        else less = len < strlen(x->seq);
        if (less) hi = m; else lo = m + 1;
    }
    return lo;
}

/* ------------------------------------------------ check_editdist_consistency */
static int check_editdist(const sctx* c, const BH* h)     /* bwt_map.cpp:2349-2465 */
{
    int64_t rlen;
    const char* ref = contig(c->g, h->ref_id, &rlen);
    if (!ref) return 0;
    size_t pos_seq = 0;
    int64_t pos_ref = h->left;
    size_t mismatch = 0, n_mismatch = 0;
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        uint32_t len = ORC_CIG_LEN(h->cig[i]);
        switch (op) {
        case ORC_MATCH:
            for (uint32_t j = 0; j < len; ++j) {
                char s = d5(pos_seq < (size_t)h->seq_len ? h->seq[pos_seq] : 'N');
                char r = refc(ref, rlen, pos_ref + j);
                if (s != r) ++mismatch;
                if (s == r && s == 'N') ++n_mismatch;
                ++pos_seq;
            }
            pos_ref += len;
            break;
        case ORC_INS: case ORC_iNS: pos_seq += len; break;
        case ORC_DEL: case ORC_REF_SKIP: pos_ref += len; break;
        default: break;
        }
    }
    return mismatch == h->mm || mismatch + n_mismatch == h->mm;
}

/* ------------------------------------------------------------ merge_chain */
/* chain[0..n) ordered left to right on the genome; returns 1 and fills *out, or 0 for BowtieHit(). */
static int merge_chain(const sctx* c, const char* read_seq, int read_len, BH* chain, int n, BH* out)
{
    const orc_span_params* p = c->p;
    int antisense = chain[0].antisense;
    uint32_t insert_id = chain[0].insert_id;
    const int left = chain[0].left;
    char seq[MAXSEQ * 2]; int seq_len = 0;
    int old_read_length = 0;
    for (int i = 0; i < n; ++i) {                                        /* :826-831 */
        memcpy(seq + seq_len, chain[i].seq, (size_t)chain[i].seq_len);
        seq_len += chain[i].seq_len;
        old_read_length += bh_read_len(&chain[i]);
    }
    /* :843-891 pre-check: at most one fusion-like gap */
    {
        size_t num_fusions = 0;
        for (int k = 1; k < n; ++k) {
            const BH* prev = &chain[k - 1]; const BH* curr = &chain[k];
            if (prev->ref_id != curr->ref_id) ++num_fusions;
            else {
                int gap = curr->left - bh_right(prev);
                int maxi = p->max_report_intron;   /* min(max_report_intron_length, fusion_min_dist = 10000000) */
                if (maxi > 10000000) maxi = 10000000;
                if (gap < -p->max_insertion_length ||
                    (gap > p->max_deletion_length && (gap < p->min_report_intron || gap > maxi)))
                    ++num_fusions;
            }
            if (num_fusions >= 2) return 0;
        }
    }
    int pi = 0, ci = 1;
    while (ci < n) {
        BH* prev = &chain[pi]; BH* curr = &chain[ci];
        antisense = prev->antisense;
        int pback = ORC_CIG_OP(prev->cig[prev->n - 1]), cfront = ORC_CIG_OP(curr->cig[0]);
        if (!(is_match_op(pback) || is_match_op(cfront))) return 0;            /* :924-928 */
        if (bh_is_spliced(prev) && bh_is_spliced(curr) && prev->antisense_splice != curr->antisense_splice)
            return 0;                                                           /* :936-943 */
        int found_closure = 0;
        int antisense_closure = bh_is_spliced(prev) ? prev->antisense_splice : curr->antisense_splice;
        uint32_t new_cigar[MAXC * 2]; int new_n = 0;
        int new_left = -1;
        int mismatch = 0;
        int prev_right_end_match_length = (int)ORC_CIG_LEN(prev->cig[prev->n - 1]);
        int curr_left_end_match_length = (int)ORC_CIG_LEN(curr->cig[0]);
        int check_fusion = prev->ref_id != curr->ref_id;
        if (prev->ref_id == curr->ref_id) {
            uint32_t reference_id = prev->ref_id;
            int64_t rlen;
            const char* ref = contig(c->g, reference_id, &rlen);
            int prev_right = bh_right(prev);
            int left_boundary = prev_right - 4, right_boundary = curr->left + 4;
            int dist_btw_two = curr->left - prev_right;
            if (dist_btw_two < 0 && dist_btw_two >= -p->max_insertion_length && prev->antisense == curr->antisense) {
                /* :1010-1306 insertion closure */
                if (!ref) return 0;
                int64_t lb = i_upper_bound(c, reference_id, (uint32_t)left_boundary, 0);
                int64_t ub = i_upper_bound(c, reference_id, (uint32_t)right_boundary, (size_t)p->max_insertion_length);
                for (; lb < ub && lb < c->n_ins; ++lb) {
                    const orc_ins_in* in = &c->ins[lb];
                    int ilen = (int)strlen(in->seq);
                    if (ilen != prev_right - curr->left) continue;
                    int itpr = prev_right - (int)in->left - 1;
                    int clti = (int)in->left - curr->left + 1;
                    if (itpr > prev_right_end_match_length || clti > curr_left_end_match_length) continue;
                    int trm = 0, insertion_mismatch = 0;
                    if (itpr > 0) {
                        /* reference ref[in->left+1, prev_right); old = tail of prev's sequence */
                        const char* old = prev->seq + prev->seq_len - itpr;
                        for (int ri = 0; ri < itpr; ++ri) {
                            char r = refc(ref, rlen, (int64_t)in->left + 1 + ri);
                            char o = d5(old[ri]);
                            if (r == 'N' || r != o) ++trm;
                            if (ri < ilen) {
                                char ic = d5(in->seq[ri]);
                                if (ic == 'N' || ic != o) { ++insertion_mismatch; break; }
                            } else {
                                char r2 = refc(ref, rlen, (int64_t)in->left + 1 + ri - ilen);
                                if (r2 == 'N' || r2 != o) --trm;
                            }
                        }
                    }
                    if (clti > 0) {
                        /* reference ref[curr->left, in->left+1); old = head of curr's sequence */
                        const char* old = curr->seq;
                        for (int ri = 0; ri < clti; ++ri) {
                            int sp = clti - ri - 1, ip = ilen - ri - 1;
                            char r = refc(ref, rlen, (int64_t)curr->left + sp);
                            char o = d5(old[sp]);
                            if (r == 'N' || r != o) ++trm;
                            if (ri < ilen) {
                                char ic = d5(in->seq[ip]);
                                if (ic == 'N' || ic != o) { ++insertion_mismatch; break; }
                            } else {
                                char r2 = refc(ref, rlen, (int64_t)curr->left + sp + ilen);
                                if (r2 == 'N' || r2 != o) --trm;
                            }
                        }
                    }
                    if (found_closure) return 0;                               /* :1243-1247 */
                    if (insertion_mismatch == 0) {
                        mismatch = -trm;
                        found_closure = 1;
                        new_left = prev->left;
                        new_n = prev->n;
                        memcpy(new_cigar, prev->cig, sizeof(uint32_t) * (size_t)prev->n);
                        uint32_t bl = ORC_CIG_LEN(new_cigar[new_n - 1]) - (uint32_t)itpr;     /* uint32 arithmetic */
                        bl &= 0x0FFFFFFFu;
                        if (bl == 0) --new_n; else new_cigar[new_n - 1] = ORC_CIG(ORC_CIG_OP(new_cigar[new_n - 1]), bl);
                        new_cigar[new_n++] = ORC_CIG(ORC_INS, (uint32_t)ilen);
                        uint32_t fl = (ORC_CIG_LEN(curr->cig[0]) + (uint32_t)(itpr - ilen)) & 0x0FFFFFFFu;
                        int cst = fl > 0 ? 0 : 1;
                        for (int q = cst; q < curr->n; ++q)
                            new_cigar[new_n++] = (q == 0) ? ORC_CIG(ORC_CIG_OP(curr->cig[0]), fl) : curr->cig[q];
                    }
                }
                if (!found_closure) return 0;
            } else if (dist_btw_two > 0 && dist_btw_two <= p->max_report_intron && prev->antisense == curr->antisense) {
                /* :1311-1591 junction / deletion closure */
                if (!ref) return 0;
                int64_t lb = j_upper_bound(c, reference_id, (uint32_t)left_boundary, (uint32_t)(right_boundary - 8), 1);
                int64_t ub = j_lower_bound(c, reference_id, (uint32_t)(left_boundary + 8), (uint32_t)right_boundary, 0);
                int new_diff_mismatches = 0xff;
                for (; lb < ub && lb < c->n_juncs; ++lb) {
                    const orc_junction* j = &c->juncs[lb];
                    int dtl = (int)j->left - prev_right + 1;
                    int dtr = (int)j->right - curr->left;
                    if (!(abs(dtl) <= 4 && abs(dtr) <= 4 && dtl == dtr)) continue;
                    if (dtl > curr_left_end_match_length || -dtl > prev_right_end_match_length) continue;
                    int new_mismatch = 0, old_mismatch = 0;
                    if (dtl > 0) {
                        /* new_cmp = ref[prev_right, j->left+1), old_cmp = ref[curr->left, j->right) */
                        for (int i = 0; i < dtl; ++i) {
                            char s = i < curr->seq_len ? curr->seq[i] : 0;
                            if (s != refc(ref, rlen, (int64_t)prev_right + i)) ++new_mismatch;
                            if (s != refc(ref, rlen, (int64_t)curr->left + i)) ++old_mismatch;
                        }
                    } else if (dtl < 0) {
                        /* new_cmp = ref[j->right, curr->left), old_cmp = ref[j->left+1, prev_right) */
                        int ad = -dtl;
                        for (int i = 0; i < ad; ++i) {
                            char s = prev->seq[prev->seq_len - (ad - i)];
                            if (s != refc(ref, rlen, (int64_t)j->right + i)) ++new_mismatch;
                            if (s != refc(ref, rlen, (int64_t)j->left + 1 + i)) ++old_mismatch;
                        }
                    }
                    int temp_diff = new_mismatch - old_mismatch;
                    if (temp_diff >= new_diff_mismatches || new_mismatch >= 2) continue;
                    new_diff_mismatches = temp_diff;
                    new_left = prev->left;
                    new_n = prev->n;
                    memcpy(new_cigar, prev->cig, sizeof(uint32_t) * (size_t)prev->n);
                    int nlb = (int)ORC_CIG_LEN(new_cigar[new_n - 1]) + dtl;
                    int nrf = (int)ORC_CIG_LEN(curr->cig[0]) - dtr;
                    if (nlb > 0) new_cigar[new_n - 1] = ORC_CIG(ORC_CIG_OP(new_cigar[new_n - 1]), (uint32_t)nlb);
                    else --new_n;
                    uint32_t skip = j->right - j->left - 1;
                    if (skip <= (uint32_t)p->max_deletion_length) {
                        new_cigar[new_n++] = ORC_CIG(ORC_DEL, skip);
                        antisense_closure = bh_is_spliced(prev) ? prev->antisense_splice : curr->antisense_splice;
                    } else {
                        new_cigar[new_n++] = ORC_CIG(ORC_REF_SKIP, skip);
                        antisense_closure = (int)j->antisense;
                    }
                    int cst = nrf > 0 ? 0 : 1;
                    for (int q = cst; q < curr->n; ++q)
                        new_cigar[new_n++] = (q == 0) ? ORC_CIG(ORC_CIG_OP(curr->cig[0]), (uint32_t)nrf) : curr->cig[q];
                    mismatch = new_diff_mismatches;
                    found_closure = 1;
                }
                if (!found_closure) return 0;
            } else if (!(dist_btw_two == 0 && prev->antisense == curr->antisense))
                check_fusion = 1;
        }
        if (check_fusion) return 0;      /* possible_fusions is empty without --fusion-search (:1596-1818) */

        if (found_closure) {                                                    /* :1822-1870 */
            if (new_n > MAXC) return 0;  /* oracle capacity guard */
            BH m;
            memset(&m, 0, sizeof m);
            int mismatches = (int)prev->mm + (int)curr->mm + mismatch;
            m.insert_id = insert_id; m.ref_id = prev->ref_id; m.left = new_left;
            m.n = new_n; memcpy(m.cig, new_cigar, sizeof(uint32_t) * (size_t)new_n);
            m.antisense = antisense; m.antisense_splice = antisense_closure;
            m.mm = (unsigned char)mismatches;
            m.ed = (unsigned char)(mismatches + gap_length(new_cigar, new_n));
            m.end = 0;
            memcpy(m.seq, prev->seq, (size_t)prev->seq_len);
            memcpy(m.seq + prev->seq_len, curr->seq, (size_t)curr->seq_len);
            m.seq_len = prev->seq_len + curr->seq_len;
            chain[pi] = m;
            for (int q = ci; q + 1 < n; ++q) chain[q] = chain[q + 1];
            --n;
            ci = pi + 1;
            continue;
        }
        ++pi; ++ci;
    }

    /* :1888-1944 concatenate */
    int saw_as = 0, saw_s = 0;
    uint32_t lc[MAXC * 4]; int ln = 0;
    int num_mm = 0;
    for (int s = 0; s < n; ++s) {
        num_mm += chain[s].mm;
        if (bh_is_spliced(&chain[s])) {
            if (chain[s].antisense_splice) { if (saw_s) return 0; saw_as = 1; }
            else { if (saw_as) return 0; saw_s = 1; }
        }
        if (ln == 0) { memcpy(lc, chain[s].cig, sizeof(uint32_t) * (size_t)chain[s].n); ln = chain[s].n; }
        else {
            int b0 = 0;
            if (ORC_CIG_OP(lc[ln - 1]) == ORC_CIG_OP(chain[s].cig[0])) {
                lc[ln - 1] = ORC_CIG(ORC_CIG_OP(lc[ln - 1]), ORC_CIG_LEN(lc[ln - 1]) + ORC_CIG_LEN(chain[s].cig[0]));
                b0 = 1;
            }
            for (int b = b0; b < chain[s].n; ++b) { if (ln >= MAXC * 4) return 0; lc[ln++] = chain[s].cig[b]; }
        }
    }
    if (ln > MAXC) return 0;   /* oracle capacity guard (reads of <= 8 segments never get here) */
    memset(out, 0, sizeof *out);
    out->insert_id = insert_id; out->ref_id = chain[0].ref_id; out->left = left;
    out->n = ln; memcpy(out->cig, lc, sizeof(uint32_t) * (size_t)ln);
    out->antisense = antisense; out->antisense_splice = saw_as;
    out->mm = (unsigned char)num_mm;
    out->ed = (unsigned char)(num_mm + gap_length(lc, ln));
    out->end = 0;
    if (seq_len > MAXSEQ) return 0;
    memcpy(out->seq, seq, (size_t)seq_len); out->seq_len = seq_len;
    (void)read_seq; (void)read_len;
    /* :2014-2033 */
    if (bh_read_len(out) != old_read_length || !check_editdist(c, out)) return 0;
    return 1;
}

/* valid_hit, long_spanning_reads.cpp:2045-2099 */
static int valid_hit(const sctx* c, const BH* bh)
{
    const orc_span_params* p = c->p;
    if (!bh->insert_id) return 0;
    for (int i = 1; i < bh->n; ++i) {
        int cop = ORC_CIG_OP(bh->cig[i]), pop = ORC_CIG_OP(bh->cig[i - 1]);
        uint32_t clen = ORC_CIG_LEN(bh->cig[i]);
        if (!is_match_op(cop) && !is_match_op(pop)) return 0;
        if ((cop == ORC_INS || cop == ORC_iNS) && clen > (uint32_t)p->max_insertion_length) return 0;
        if ((cop == ORC_DEL || cop == ORC_dEL) && clen > (uint32_t)p->max_deletion_length) return 0;
        if ((cop == ORC_REF_SKIP || cop == ORC_rEF_SKIP) && (uint64_t)clen < (uint64_t)p->min_report_intron) return 0;
    }
    if (!is_match_op(ORC_CIG_OP(bh->cig[0])) || !is_match_op(ORC_CIG_OP(bh->cig[bh->n - 1]))) return 0;
    return 1;
}

typedef struct { BH* v; int n, cap; } bhvec;
static void bhpush(bhvec* a, const BH* h)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 8; a->v = (BH*)realloc(a->v, sizeof(BH) * (size_t)a->cap); }
    a->v[a->n++] = *h;
}

/* merge_segment_chain, :2101-2220 (fusion_dir == FUSION_NOTHING) */
static void merge_segment_chain(const sctx* c, const char* read_seq, int read_len, const BH* stack, int n, bhvec* joined)
{
    if (n == 0) return;
    BH bh;
    if (n > 1) {
        BH chain[16];
        if (stack[0].antisense) for (int i = 0; i < n; ++i) chain[i] = stack[n - 1 - i];   /* :2118-2123 */
        else for (int i = 0; i < n; ++i) chain[i] = stack[i];
        if (!merge_chain(c, read_seq, read_len, chain, n, &bh)) memset(&bh, 0, sizeof bh);
    } else bh = stack[0];
    if (valid_hit(c, &bh)) bhpush(joined, &bh);
}

typedef struct { const BH* v; int n; } seglist;

/* dfs_seg_hits, :2222-2610 with fusion_search == false */
static int dfs(const sctx* c, const char* read_seq, int read_len, const seglist* segs, int nsegs, int curr,
               BH* stack, int depth, bhvec* joined, int* num_try)
{
    const orc_span_params* p = c->p;
    if (*num_try <= 0) return 0;
    int join_success = 0;
    if (curr < nsegs) {
        for (int i = 0; i < segs[curr].n; ++i) {
            const BH* bh = &segs[curr].v[i];
            const BH* prev = &stack[depth - 1];
            const BH* ph = prev; const BH* ch = bh;
            /* no fusion ops in either hit */
            int dir_set = 0;
            if (ph->antisense && ch->antisense && ph->ref_id == ch->ref_id) {      /* :2352-2359 swap */
                const BH* t = ph; ph = ch; ch = t;
            } else {
                if (ph->ref_id == ch->ref_id && ph->antisense == ch->antisense) {   /* :2360-2378 */
                    int dist = ph->antisense ? ph->left - bh_right(ch) : ch->left - bh_right(ph);
                    if (dist > p->max_report_intron || dist < -p->max_insertion_length) dir_set = 1;
                } else dir_set = 1;                                                 /* :2379-2399 */
            }
            if (dir_set) continue;                                                  /* :2402 */
            if (ph->ref_id != ch->ref_id) continue;                                 /* :2518-2522 */
            int dist = ch->left - bh_right(ph);                                     /* :2531-2543 */
            if (dist <= p->max_report_intron && dist >= -p->max_insertion_length) { /* :2553-2556 */
                stack[depth] = *bh;
                if (dfs(c, read_seq, read_len, segs, nsegs, curr + 1, stack, depth + 1, joined, num_try))
                    join_success = 1;
                if (*num_try <= 0) return join_success;
            }
        }
    } else {
        --*num_try;
        merge_segment_chain(c, read_seq, read_len, stack, depth, joined);
        return 1;
    }
    return join_success;
}

/* BowtieHit::operator< (bwt_map.h:180-207) */
static int bh_less(const BH* a, const BH* b)
{
    if (a->insert_id != b->insert_id) return a->insert_id < b->insert_id;
    if (a->ref_id != b->ref_id) return a->ref_id < b->ref_id;
    if (a->left != b->left) return a->left < b->left;
    if (a->antisense != b->antisense) return a->antisense < b->antisense;
    if (a->mm != b->mm) return a->mm < b->mm;
    if (a->ed != b->ed) return a->ed < b->ed;
    if (a->n != b->n) return a->n < b->n;
    for (int i = 0; i < a->n; ++i)
        if (a->cig[i] != b->cig[i]) {
            int oa = ORC_CIG_OP(a->cig[i]), ob = ORC_CIG_OP(b->cig[i]);
            return oa < ob || (oa == ob && ORC_CIG_LEN(a->cig[i]) < ORC_CIG_LEN(b->cig[i]));
        }
    return 0;
}
/* BowtieHit::operator== (bwt_map.h:167-178) */
static int bh_eq(const BH* a, const BH* b)
{
    if (a->insert_id != b->insert_id || a->ref_id != b->ref_id || a->antisense != b->antisense || a->left != b->left ||
        a->antisense_splice != b->antisense_splice || a->ed != b->ed || a->n != b->n) return 0;
    return memcmp(a->cig, b->cig, sizeof(uint32_t) * (size_t)a->n) == 0;
}

/* bowtie_sam_extra, bwt_map.cpp:2467-2648 */
static void sam_extra(const sctx* c, const BH* bh, const char* qual, int qual_len, orc_aln* o)
{
    const orc_span_params* p = c->p;
    int64_t rlen;
    const char* ref = contig(c->g, bh->ref_id, &rlen);
    o->AS = o->XM = o->XO = o->XG = 0; o->md[0] = 0;
    if (!ref) return;
    size_t pos_seq = 0, pos_mismatch = 0, mismatch = 0, opens = 0, conts = 0;
    int64_t pos_ref = bh->left;
    int AS = 0;
    char md[2048]; int ml = 0;
    for (int i = 0; i < bh->n; ++i) {
        int op = ORC_CIG_OP(bh->cig[i]);
        uint32_t len = ORC_CIG_LEN(bh->cig[i]);
        if (op == ORC_MATCH) {
            for (uint32_t j = 0; j < len; ++j) {
                char r = refc(ref, rlen, pos_ref + j);
                char s = pos_seq < (size_t)bh->seq_len ? bh->seq[pos_seq] : 'N';
                if (d5(s) != r) {
                    ++mismatch;
                    if (pos_seq < (size_t)qual_len) {
                        if (d5(s) == 'N' || r == 'N') AS -= p->bowtie2_penalty_for_N;
                        else {
                            int q = qual[pos_seq] - '!'; if (q > 40) q = 40;
                            float penalty = p->bowtie2_min_penalty + (p->bowtie2_max_penalty - p->bowtie2_min_penalty) * q / 40.0;
                            AS -= (int)penalty;
                        }
                    }
                    ml += sprintf(md + ml, "%d%c", (int)pos_mismatch, r);
                    pos_mismatch = 0;
                } else {
                    if (r == 'N') AS -= p->bowtie2_penalty_for_N;
                    ++pos_mismatch;
                }
                ++pos_seq;
            }
            pos_ref += len;
        } else if (op == ORC_INS) {
            pos_seq += len;
            AS -= p->bowtie2_read_gap_open; AS -= (int)(p->bowtie2_read_gap_cont * len);
            opens += 1; conts += len;
        } else if (op == ORC_DEL) {
            AS -= p->bowtie2_ref_gap_open; AS -= (int)(p->bowtie2_ref_gap_cont * len);
            opens += 1; conts += len;
            ml += sprintf(md + ml, "%d^", (int)pos_mismatch);
            for (uint32_t k = 0; k < len; ++k) md[ml++] = refc(ref, rlen, pos_ref + k);
            pos_ref += len;
            pos_mismatch = 0;
        } else if (op == ORC_REF_SKIP) pos_ref += len;
        if (ml > 1900) break;
    }
    ml += sprintf(md + ml, "%d", (int)pos_mismatch);
    md[ml] = 0;
    o->AS = AS; o->XM = (int)mismatch; o->XO = (int)opens; o->XG = (int)conts;
    if (ml < (int)sizeof o->md) strcpy(o->md, md);
    else snprintf(o->md, sizeof o->md, "\x01%lld", (long long)orc_long_md_put(md));
}

int orc_spanning_batch(const orc_span_params* p, const orc_genome* g, const orc_span_batch* b,
                       const orc_junction* juncs, int64_t n_juncs, const orc_ins_in* ins, int64_t n_ins,
                       orc_aln** out, int64_t* n_out)
{
    sctx c; c.p = p; c.g = g; c.juncs = juncs; c.n_juncs = n_juncs; c.ins = ins; c.n_ins = n_ins;
    orc_aln* res = NULL; int64_t nres = 0, cap = 0;
    const int L = p->segment_length;
    for (int r = 0; r < b->n_reads; ++r) {
        const int64_t* so = b->seg_off + (int64_t)r * b->nseg;
        const char* rseq = b->bases + b->read_off[r];
        const char* rqual = b->quals + b->read_off[r];
        int rl = (int)(b->read_off[r + 1] - b->read_off[r]);
        if (so[1] == so[0]) continue;                    /* the worker iterates over segment-1 groups only (:2706-2765) */
        /* look_right_for_hit_group stops at the first segment without hits (:151-152) */
        int nsegs = 0;
        while (nsegs < b->nseg && so[nsegs + 1] > so[nsegs]) ++nsegs;
        /* :2777-2785 */
        const orc_span_hit* lastfirst = &b->hits[so[nsegs - 1]];
        if (!(lastfirst->flags & ORC_HIT_END)) continue;
        if (rl > MAXSEQ) continue;
        /* build BH lists with their segment sequences */
        seglist segs[16]; BH* store[16];
        if (nsegs > 16) continue;
        int multihit_drop = 0;
        for (int s = 0; s < nsegs; ++s) {
            int n = (int)(so[s + 1] - so[s]);
            if (p->bowtie2 && n > p->max_seg_multihits) multihit_drop = 1;      /* :2625-2632 */
            store[s] = (BH*)calloc((size_t)n, sizeof(BH));
            for (int k = 0; k < n; ++k) {
                const orc_span_hit* h = &b->hits[so[s] + k];
                BH* x = &store[s][k];
                x->insert_id = (uint32_t)r + 1;          /* any non-zero id; one read at a time */
                x->ref_id = h->ref_id; x->left = h->left; x->n = h->n_cigar;
                memcpy(x->cig, h->cigar, sizeof(uint32_t) * (size_t)h->n_cigar);
                x->antisense = (h->flags & ORC_HIT_ANTISENSE) != 0;
                x->antisense_splice = (h->flags & ORC_HIT_ANTISENSE_SPLICE) != 0;
                x->end = (h->flags & ORC_HIT_END) != 0;
                x->mm = h->mismatches; x->ed = h->edit_dist;
                /* the segment record's SEQ: the read piece, reverse-complemented when the hit is antisense */
                int st = s * L; if (st > rl) st = rl;
                int ln = x->end ? rl - st : L; if (ln > rl - st) ln = rl - st;
                for (int q = 0; q < ln; ++q)
                    x->seq[q] = x->antisense ? comp(rseq[st + ln - 1 - q]) : rseq[st + q];
                x->seq_len = ln;
            }
            segs[s].v = store[s]; segs[s].n = n;
        }
        bhvec joined = {0, 0, 0};
        if (!multihit_drop) {
            BH stack[17];
            for (int i = 0; i < segs[0].n; ++i) {                                /* :2634-2664 */
                stack[0] = segs[0].v[i];
                int num_try = 10000;
                dfs(&c, rseq, rl, segs, nsegs, 1, stack, 1, &joined, &num_try);
            }
        }
        /* sort + unique (:2805-2807); insertion sort = what libstdc++ does below 16 elements, and stable */
        for (int i = 1; i < joined.n; ++i) {
            BH t = joined.v[i]; int k = i;
            while (k > 0 && bh_less(&t, &joined.v[k - 1])) { joined.v[k] = joined.v[k - 1]; --k; }
            joined.v[k] = t;
        }
        int w = 0;
        for (int i = 0; i < joined.n; ++i)
            if (w == 0 || !bh_eq(&joined.v[w - 1], &joined.v[i])) joined.v[w++] = joined.v[i];
        joined.n = w;
        for (int i = 0; i < joined.n; ++i) {
            const BH* h = &joined.v[i];
            int gapl = (unsigned char)(h->ed - h->mm);
            if (h->mm > p->read_mismatches || gapl > p->read_gap_length || h->ed > p->read_edit_dist) continue;  /* :2810-2813 */
            if (nres == cap) { cap = cap ? cap * 2 : 1024; res = (orc_aln*)realloc(res, sizeof(orc_aln) * (size_t)cap); }
            orc_aln* o = &res[nres++];
            memset(o, 0, sizeof *o);
            o->read_idx = r; o->ref_id = h->ref_id; o->left = h->left;
            o->antisense = (uint8_t)h->antisense; o->antisense_splice = (uint8_t)h->antisense_splice;
            o->mismatches = h->mm; o->edit_dist = h->ed;
            o->n_cigar = h->n; memcpy(o->cigar, h->cig, sizeof(uint32_t) * (size_t)h->n);
            /* merge_chain :1966-1978 qual: read quals, reversed when the joined seq differs from the read */
            char q[MAXSEQ];
            int same = (h->seq_len == rl) && memcmp(h->seq, rseq, (size_t)rl) == 0;
            if (nsegs == 1) same = 1;        /* single-segment hits keep the record's own qual: SEQ/QUAL of the BAM record */
            for (int k = 0; k < rl; ++k) q[k] = same ? rqual[k] : rqual[rl - 1 - k];
            if (nsegs == 1 && h->antisense) for (int k = 0; k < rl; ++k) q[k] = rqual[rl - 1 - k];
            sam_extra(&c, h, q, rl, o);
        }
        free(joined.v);
        for (int s = 0; s < nsegs; ++s) free(store[s]);
    }
    *out = res; *n_out = nres;
    return 0;
}

void orc_free(void* p) { free(p); }

/* per-thread pool for the MD strings that do not fit a record (see thj_oracle.h) */
static __thread char* lmd_pool = NULL;
static __thread size_t lmd_n = 0, lmd_cap = 0;
void orc_long_md_reset(void) { lmd_n = 0; }
int64_t orc_long_md_put(const char* s)
{
    size_t l = strlen(s) + 1;
    if (lmd_n + l > lmd_cap) { lmd_cap = (lmd_n + l) * 2 + 4096; lmd_pool = (char*)realloc(lmd_pool, lmd_cap); }
    memcpy(lmd_pool + lmd_n, s, l);
    int64_t off = (int64_t)lmd_n;
    lmd_n += l;
    return off;
}
const char* orc_long_md(int64_t off) { return (off >= 0 && (size_t)off < lmd_n) ? lmd_pool + off : ""; }
