/*
 * spanning_fusion_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see thj_oracle.h) for
 * TopHat's long_spanning_reads WITH its fusion branches: the whole of dfs_seg_hits /
 * merge_segment_chain / merge_chain including reversed (lower-case) CIGAR pieces, fused
 * segment hits from the fusion junction database, the fusion closure search and
 * BowtieHit::reverse.  With fusion_search == 0 it must give what spanning_oracle.c gives
 * (tests check that), so the two files are two restatements of the shared part.
 *
 * Plain-C restatement of DaehwanKimLab/tophat v2.1.2:
 *   JoinSegmentsWorker::operator()   long_spanning_reads.cpp:2669-2845
 *   join_segments_for_read           :2612-2667
 *   dfs_seg_hits                     :2222-2610
 *   merge_segment_chain              :2101-2220
 *   merge_chain                      :805-2038
 *   valid_hit                        :2045-2099
 *   BowtieHit::right / read_len / fusion_opcode / is_forwarding_* / antisense_align2 / reverse
 *                                    bwt_map.h:141-163, :213-243, :254-442
 *   BowtieHit::operator< / ==        bwt_map.h:167-207
 *   BowtieHit::check_editdist_consistency  bwt_map.cpp:2349-2465
 *   bowtie_sam_extra                 bwt_map.cpp:2467-2648
 *   fusions_from_spliced_hit         fusions.cpp:441-495
 *   Fusion::operator<                fusions.h:44-71
 * Colour space is out of scope and omitted.
 *
 * PARITY: the fusion branches are unpinned by the reference's own tests (it ships none for
 * --fusion-search); checked differentially against the survey-stage scratch build only
 * (tests/golden/pe100_fusion_span, oracle/README.md).
 */
#include "thj_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXC 32
#define MAXSEQ 320

enum { FUS_NONE = 0, FUS_FF = ORC_FUSION_FF, FUS_FR = ORC_FUSION_FR, FUS_RF = ORC_FUSION_RF, FUS_RR = ORC_FUSION_RR };

typedef struct {
    uint32_t insert_id;        /* 0 = the empty BowtieHit() */
    uint32_t ref_id, ref_id2;
    int left;
    int n;
    uint32_t cig[MAXC];
    int antisense, antisense_splice, end;
    unsigned char mm, ed;
    char seq[MAXSEQ], qual[MAXSEQ];
    int seq_len;
} BH;

typedef struct {
    const orc_span_params* p;
    int fusion_search, fusion_min_dist;
    const orc_genome* g;
    const orc_junction* juncs; int64_t n_juncs;
    const orc_ins_in* ins; int64_t n_ins;
    const orc_fusion_in* fus; int64_t n_fus;
} fctx;

static int is_fusion_op(int op) { return op == FUS_FF || op == FUS_FR || op == FUS_RF || op == FUS_RR; }
static int is_match_op(int op) { return op == ORC_MATCH || op == ORC_mATCH; }

static int bh_right(const BH* h)            /* bwt_map.h:213-243 */
{
    int r = h->left;
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        int len = (int)ORC_CIG_LEN(h->cig[i]);
        if (op == ORC_MATCH || op == ORC_REF_SKIP || op == ORC_DEL) r += len;
        else if (op == ORC_mATCH || op == ORC_rEF_SKIP || op == ORC_dEL) r -= len;
        else if (is_fusion_op(op)) r = len;
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
static int bh_is_spliced(const BH* h)       /* bwt_map.h:245-254 */
{
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        if (op == ORC_REF_SKIP || op == ORC_rEF_SKIP) return 1;
    }
    return 0;
}
static int bh_fusion_opcode(const BH* h)    /* bwt_map.h:256-265 */
{
    for (int i = 0; i < h->n; ++i)
        if (is_fusion_op(ORC_CIG_OP(h->cig[i]))) return ORC_CIG_OP(h->cig[i]);
    return FUS_NONE;
}
static int fwd_op(int op) { return op == ORC_MATCH || op == ORC_REF_SKIP || op == ORC_INS || op == ORC_DEL; }
static int rev_op(int op) { return op == ORC_mATCH || op == ORC_rEF_SKIP || op == ORC_iNS || op == ORC_dEL; }
static int bh_forwarding_left(const BH* h)  /* bwt_map.h:271-289 */
{
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        if (fwd_op(op)) return 1;
        if (rev_op(op)) return 0;
        if (is_fusion_op(op)) break;
    }
    return 1;
}
static int bh_forwarding_right(const BH* h) /* bwt_map.h:295-313 */
{
    for (int i = h->n - 1; i >= 0; --i) {
        int op = ORC_CIG_OP(h->cig[i]);
        if (fwd_op(op)) return 1;
        if (rev_op(op)) return 0;
        if (is_fusion_op(op)) break;
    }
    return 1;
}
static int bh_antisense2(const BH* h)       /* bwt_map.h:319-329 */
{
    int f = bh_fusion_opcode(h);
    if (f == FUS_NONE || f == FUS_FF || f == FUS_RR) return h->antisense;
    return !h->antisense;
}
static char comp(char c)
{
    switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
                 case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a'; default: return 'N'; }
}
static char d5(char c) { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N'; }
static char comp5(char c) { return comp(d5(c)); }

/* BowtieHit::reverse, bwt_map.h:331-442 */
static void bh_reverse(const BH* h, BH* out)
{
    BH r;
    memset(&r, 0, sizeof r);
    r.ref_id = h->ref_id2; r.ref_id2 = h->ref_id; r.insert_id = h->insert_id;
    uint32_t right, fusion_pos;
    right = fusion_pos = (uint32_t)h->left;
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]); uint32_t len = ORC_CIG_LEN(h->cig[i]);
        if (op == ORC_MATCH || op == ORC_REF_SKIP || op == ORC_DEL) right += len;
        else if (op == ORC_mATCH || op == ORC_rEF_SKIP || op == ORC_dEL) right -= len;
        else if (is_fusion_op(op)) { fusion_pos = right; right = len; }
    }
    if (bh_forwarding_left(h)) fusion_pos -= 1; else fusion_pos += 1;
    int f = bh_fusion_opcode(h);
    if (f == FUS_NONE || f == FUS_FF || f == FUS_RR) {
        if (bh_forwarding_left(h)) r.left = (int)(right - 1); else r.left = (int)(right + 1);
    } else {
        if (f == FUS_FR) r.left = (int)(right + 1); else r.left = (int)(right - 1);
    }
    r.n = h->n;
    for (int i = h->n - 1, k = 0; i >= 0; --i, ++k) {
        int op = ORC_CIG_OP(h->cig[i]); uint32_t len = ORC_CIG_LEN(h->cig[i]);
        switch (op) {
        case ORC_MATCH: op = ORC_mATCH; break;
        case ORC_mATCH: op = ORC_MATCH; break;
        case ORC_INS: op = ORC_iNS; break;
        case ORC_iNS: op = ORC_INS; break;
        case ORC_DEL: op = ORC_dEL; break;
        case ORC_dEL: op = ORC_DEL; break;
        case ORC_REF_SKIP: op = ORC_rEF_SKIP; break;
        case ORC_rEF_SKIP: op = ORC_REF_SKIP; break;
        default: if (is_fusion_op(op)) len = fusion_pos; break;
        }
        r.cig[k] = ORC_CIG(op, len);
    }
    r.antisense = (f == FUS_FR || f == FUS_RF) ? !h->antisense : h->antisense;
    r.antisense_splice = h->antisense_splice;
    r.mm = h->mm; r.ed = h->ed; r.end = h->end;
    r.seq_len = h->seq_len;
    for (int i = 0; i < h->seq_len; ++i) {           /* reverse_complement(string) keeps non-ACGT as N; quals reversed */
        r.seq[i] = comp(h->seq[h->seq_len - 1 - i]);
        r.qual[i] = h->qual[h->seq_len - 1 - i];
    }
    *out = r;
}

static int gap_length(const uint32_t* cig, int n)   /* bwt_map.cpp:32-43 */
{
    int e = 0;
    for (int i = 0; i < n; ++i) {
        int op = ORC_CIG_OP(cig[i]);
        if (op == ORC_INS || op == ORC_iNS || op == ORC_DEL || op == ORC_dEL) e += (int)ORC_CIG_LEN(cig[i]);
    }
    return e;
}

static const char* contig(const orc_genome* g, uint32_t ref_id, int64_t* len)
{
    if (ref_id == 0 || (int64_t)ref_id > g->n_contigs) { *len = 0; return NULL; }
    *len = g->len[ref_id - 1];
    return g->seq[ref_id - 1];
}
static char refc(const char* ref, int64_t len, int64_t pos) { return (pos < 0 || pos >= len) ? 'N' : ref[pos]; }
static char refrc(const char* ref, int64_t len, int64_t pos) { return comp(refc(ref, len, pos)); }

/* ---- ordered-set bounds ---- */
static int jkey_less(uint32_t r1, uint32_t l1, uint32_t rr1, uint32_t a1, const orc_junction* j)   /* junctions.h:39-57 */
{
    if (r1 != j->ref_id) return r1 < j->ref_id;
    if (l1 != j->left) return l1 < j->left;
    if (rr1 != j->right) return rr1 < j->right;
    return a1 < j->antisense;
}
static int j_less_key(const orc_junction* j, uint32_t r1, uint32_t l1, uint32_t rr1, uint32_t a1)
{
    if (j->ref_id != r1) return j->ref_id < r1;
    if (j->left != l1) return j->left < l1;
    if (j->right != rr1) return j->right < rr1;
    return j->antisense < a1;
}
static int64_t j_upper_bound(const fctx* c, uint32_t r, uint32_t l, uint32_t rr, uint32_t a)
{
    int64_t lo = 0, hi = c->n_juncs;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (jkey_less(r, l, rr, a, &c->juncs[m])) hi = m; else lo = m + 1; }
    return lo;
}
static int64_t j_lower_bound(const fctx* c, uint32_t r, uint32_t l, uint32_t rr, uint32_t a)
{
    int64_t lo = 0, hi = c->n_juncs;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (j_less_key(&c->juncs[m], r, l, rr, a)) lo = m + 1; else hi = m; }
    return lo;
}
static int64_t i_upper_bound(const fctx* c, uint32_t r, uint32_t l, size_t len)   /* insertions.h:52-67 */
{
    int64_t lo = 0, hi = c->n_ins;
    while (lo < hi) {
        int64_t m = (lo + hi) / 2;
        const orc_ins_in* x = &c->ins[m];
        int less;
        if (r != x->ref_id) less = r < x->ref_id;
        else if (l != x->left) less = l < x->left;
        else less = len < strlen(x->seq);
        if (less) hi = m; else lo = m + 1;
    }
    return lo;
}
static int fus_cmp(const orc_fusion_in* a, const orc_fusion_in* b)    /* fusions.h:44-71 */
{
    if (a->ref1 != b->ref1) return a->ref1 < b->ref1 ? -1 : 1;
    if (a->ref2 != b->ref2) return a->ref2 < b->ref2 ? -1 : 1;
    if (a->left != b->left) return a->left < b->left ? -1 : 1;
    if (a->right != b->right) return a->right < b->right ? -1 : 1;
    if (a->dir != b->dir) return a->dir < b->dir ? -1 : 1;
    return 0;
}
static int64_t f_upper_bound(const fctx* c, const orc_fusion_in* k)
{
    int64_t lo = 0, hi = c->n_fus;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (fus_cmp(k, &c->fus[m]) < 0) hi = m; else lo = m + 1; }
    return lo;
}
static int64_t f_lower_bound(const fctx* c, const orc_fusion_in* k)
{
    int64_t lo = 0, hi = c->n_fus;
    while (lo < hi) { int64_t m = (lo + hi) / 2; if (fus_cmp(&c->fus[m], k) < 0) lo = m + 1; else hi = m; }
    return lo;
}

/* ------------------------------------------------ check_editdist_consistency, bwt_map.cpp:2349-2465 */
static int check_editdist(const fctx* c, const BH* h)
{
    int64_t rlen1, rlen2;
    const char* ref1 = contig(c->g, h->ref_id, &rlen1);
    const char* ref2 = contig(c->g, h->ref_id2, &rlen2);
    if (!ref1 || !ref2) return 0;
    const char* ref = ref1; int64_t rlen = rlen1;
    size_t pos_seq = 0;
    int64_t pos_ref = h->left;
    size_t mismatch = 0, n_mismatch = 0;
    int saw_fusion = 0;
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]);
        int64_t len = ORC_CIG_LEN(h->cig[i]);
        switch (op) {
        case ORC_MATCH:
            for (int64_t j = 0; j < len; ++j) {
                char s = d5(pos_seq < (size_t)h->seq_len ? h->seq[pos_seq] : 'N');
                char r = refc(ref, rlen, pos_ref + j);
                if (s != r) ++mismatch;
                if (s == r && s == 'N') ++n_mismatch;
                ++pos_seq;
            }
            pos_ref += len;
            break;
        case ORC_mATCH:
            /* infix(pos_ref - len + 1, pos_ref + 1) reverse-complemented: element j = comp(ref[pos_ref - j]) */
            for (int64_t j = 0; j < len; ++j) {
                char s = d5(pos_seq < (size_t)h->seq_len ? h->seq[pos_seq] : 'N');
                char r = refrc(ref, rlen, pos_ref - j);
                if (s != r) ++mismatch;
                if (s == r && s == 'N') ++n_mismatch;
                ++pos_seq;
            }
            pos_ref -= len;
            break;
        case ORC_INS: case ORC_iNS: pos_seq += (size_t)len; break;
        case ORC_DEL: case ORC_REF_SKIP: pos_ref += len; break;
        case ORC_dEL: case ORC_rEF_SKIP: pos_ref -= len; break;
        default:
            if (is_fusion_op(op)) {
                if (saw_fusion) return 0;
                ref = ref2; rlen = rlen2;
                pos_ref = len;
                saw_fusion = 1;
            }
            break;
        }
    }
    return mismatch == h->mm || mismatch + n_mismatch == h->mm;
}

/* fusions_from_spliced_hit(bh, fusions, auto_sort = false)[0] -> (left, right); 0 when the hit has no fusion (fusions.cpp:441-495) */
static int first_fusion(const BH* h, uint32_t* fl, uint32_t* fr)
{
    uint32_t pos = (uint32_t)h->left;
    for (int i = 0; i < h->n; ++i) {
        int op = ORC_CIG_OP(h->cig[i]); uint32_t len = ORC_CIG_LEN(h->cig[i]);
        if (op == ORC_REF_SKIP || op == ORC_MATCH || op == ORC_DEL) pos += len;
        else if (op == ORC_rEF_SKIP || op == ORC_mATCH || op == ORC_dEL) pos -= len;
        else if (is_fusion_op(op)) {
            if (op == FUS_RF || op == FUS_RR) pos = pos + 1; else pos = pos - 1;
            *fl = pos; *fr = len;
            return 1;
        }
    }
    return 0;
}

static char rs_at(const char* read_seq, int read_len, int pos) { return (pos >= 0 && pos < read_len) ? read_seq[pos] : 0; }

/* ------------------------------------------------------------ merge_chain, :805-2038 */
static int merge_chain(const fctx* c, const char* read_seq, const char* read_qual, int read_len, BH* chain, int n, int fusion_dir, BH* out)
{
    const orc_span_params* p = c->p;
    const int L = p->segment_length;
    int antisense = chain[0].antisense;
    uint32_t insert_id = chain[0].insert_id;
    const int left = chain[0].left;
    char seq[MAXSEQ * 2], qual[MAXSEQ * 2]; int seq_len = 0;
    int old_read_length = 0;
    for (int i = 0; i < n; ++i) {                                        /* :826-831 */
        if (seq_len + chain[i].seq_len > MAXSEQ) return 0;
        memcpy(seq + seq_len, chain[i].seq, (size_t)chain[i].seq_len);
        memcpy(qual + seq_len, chain[i].qual, (size_t)chain[i].seq_len);
        seq_len += chain[i].seq_len;
        old_read_length += bh_read_len(&chain[i]);
    }
    /* :843-897 */
    {
        size_t num_fusions = bh_fusion_opcode(&chain[0]) == FUS_NONE ? 0 : 1;
        int fusion_passed = 0;
        for (int k = 1; k < n; ++k) {
            const BH* prev = &chain[k - 1]; const BH* curr = &chain[k];
            if (prev->ref_id != prev->ref_id2 || prev->ref_id2 != curr->ref_id) fusion_passed = 1;
            if (prev->ref_id2 != curr->ref_id) ++num_fusions;
            if (bh_fusion_opcode(curr) != FUS_NONE) ++num_fusions;
            if (prev->ref_id2 == curr->ref_id) {
                int reversed = (fusion_dir == FUS_FR && fusion_passed) || (fusion_dir == FUS_RF && !fusion_passed);
                int gap = reversed ? bh_right(prev) - curr->left : curr->left - bh_right(prev);
                int maxi = p->max_report_intron < c->fusion_min_dist ? p->max_report_intron : c->fusion_min_dist;
                if (gap < -p->max_insertion_length ||
                    (gap > p->max_deletion_length && (gap < p->min_report_intron || gap > maxi))) {
                    fusion_passed = 1;
                    ++num_fusions;
                }
            }
            if (num_fusions >= 2) return 0;
        }
    }
    int pi = 0, ci = 1;
    int curr_seg_index = 1;
    int fusion_passed = 0;
    while (ci < n) {
        BH* prev = &chain[pi]; BH* curr = &chain[ci];
        antisense = prev->antisense;
        if (bh_fusion_opcode(prev) != FUS_NONE || prev->ref_id2 != curr->ref_id) fusion_passed = 1;     /* :925-926 */
        int pback = ORC_CIG_OP(prev->cig[prev->n - 1]), cfront = ORC_CIG_OP(curr->cig[0]);
        if (!(is_match_op(pback) || is_match_op(cfront))) return 0;            /* :934-938 */
        if (bh_is_spliced(prev) && bh_is_spliced(curr) && prev->antisense_splice != curr->antisense_splice)
            return 0;                                                           /* :946-953 */
        int found_closure = 0;
        int antisense_closure = bh_is_spliced(prev) ? prev->antisense_splice : curr->antisense_splice;
        uint32_t new_cigar[MAXC * 2]; int new_n = 0;
        int new_left = -1;
        int mismatch = 0;
        int prev_right_end_match_length = (int)ORC_CIG_LEN(prev->cig[prev->n - 1]);
        int curr_left_end_match_length = (int)ORC_CIG_LEN(curr->cig[0]);
        int check_fusion = prev->ref_id2 != curr->ref_id;
        const int prev_right = bh_right(prev);

        if (prev->ref_id2 == curr->ref_id) {
            int reversed = (fusion_dir == FUS_FR && fusion_passed) || (fusion_dir == FUS_RF && !fusion_passed);   /* :985-987 */
            uint32_t reference_id = prev->ref_id2;
            int64_t rlen;
            const char* ref = contig(c->g, reference_id, &rlen);
            int left_boundary, right_boundary;
            if (reversed) { left_boundary = curr->left - 4; right_boundary = prev_right + 4; }
            else { left_boundary = prev_right - 4; right_boundary = curr->left + 4; }
            int dist_btw_two = reversed ? prev_right - curr->left : curr->left - prev_right;

            if (dist_btw_two < 0 && dist_btw_two >= -p->max_insertion_length && bh_antisense2(prev) == curr->antisense) {
                /* :1010-1306 insertion closure */
                if (!ref) return 0;
                int64_t lb = i_upper_bound(c, reference_id, (uint32_t)left_boundary, 0);
                int64_t ub = i_upper_bound(c, reference_id, (uint32_t)right_boundary, (size_t)p->max_insertion_length);
                for (; lb < ub && lb < c->n_ins; ++lb) {
                    const orc_ins_in* in = &c->ins[lb];
                    int ilen = (int)strlen(in->seq);
                    if (ilen != (reversed ? curr->left - prev_right : prev_right - curr->left)) continue;
                    int itpr, clti;
                    if (reversed) { itpr = (int)in->left - prev_right; clti = curr->left - (int)in->left; }
                    else { itpr = prev_right - (int)in->left - 1; clti = (int)in->left - curr->left + 1; }
                    if (itpr > prev_right_end_match_length || clti > curr_left_end_match_length) continue;
                    int trm = 0, insertion_mismatch = 0;
                    /* the inserted sequence as the read sees it: reverse-complemented in the reversed case */
                    char iseq[32];
                    for (int q = 0; q < ilen && q < 31; ++q) iseq[q] = reversed ? comp5(in->seq[ilen - 1 - q]) : d5(in->seq[q]);
                    if (itpr > 0) {
                        for (int ri = 0; ri < itpr; ++ri) {
                            char r, o, r2;
                            if (reversed) {
                                /* reference = rc(ref[prev_right+1, in->left+1)): element k = comp(ref[in->left - k]);
                                   old = read_seq.substr(curr_seg_index * L - itpr, itpr) */
                                r = refrc(ref, rlen, (int64_t)in->left - ri);
                                o = d5(rs_at(read_seq, read_len, curr_seg_index * L - itpr + ri));
                                r2 = refrc(ref, rlen, (int64_t)in->left - (ri - ilen));
                            } else {
                                /* reference = ref[in->left+1, prev_right); old = tail of prev's sequence */
                                r = refc(ref, rlen, (int64_t)in->left + 1 + ri);
                                o = d5(prev->seq[prev->seq_len - itpr + ri]);
                                r2 = refc(ref, rlen, (int64_t)in->left + 1 + ri - ilen);
                            }
                            if (r == 'N' || r != o) ++trm;
                            if (ri < ilen) {
                                if (iseq[ri] == 'N' || iseq[ri] != o) { ++insertion_mismatch; break; }
                            } else {
                                if (r2 == 'N' || r2 != o) --trm;
                            }
                        }
                    }
                    if (clti > 0) {
                        for (int ri = 0; ri < clti; ++ri) {
                            int sp = clti - ri - 1, ip = ilen - ri - 1;
                            char r, o, r2;
                            if (reversed) {
                                /* reference = rc(ref[in->left+1, curr->left+1)): element k = comp(ref[curr->left - k]);
                                   old = read_seq.substr(curr_seg_index * L, clti) */
                                r = refrc(ref, rlen, (int64_t)curr->left - sp);
                                o = d5(rs_at(read_seq, read_len, curr_seg_index * L + sp));
                                r2 = refrc(ref, rlen, (int64_t)curr->left - (sp + ilen));
                            } else {
                                /* reference = ref[curr->left, in->left+1); old = head of curr's sequence */
                                r = refc(ref, rlen, (int64_t)curr->left + sp);
                                o = d5(curr->seq[sp]);
                                r2 = refc(ref, rlen, (int64_t)curr->left + sp + ilen);
                            }
                            if (r == 'N' || r != o) ++trm;
                            if (ri < ilen) {
                                if (iseq[ip] == 'N' || iseq[ip] != o) { ++insertion_mismatch; break; }
                            } else {
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
                        uint32_t bl = (ORC_CIG_LEN(new_cigar[new_n - 1]) - (uint32_t)itpr) & 0x0FFFFFFFu;    /* uint32 arithmetic */
                        if (bl == 0) --new_n; else new_cigar[new_n - 1] = ORC_CIG(ORC_CIG_OP(new_cigar[new_n - 1]), bl);
                        new_cigar[new_n++] = ORC_CIG(reversed ? ORC_iNS : ORC_INS, (uint32_t)ilen);
                        uint32_t fl = (ORC_CIG_LEN(curr->cig[0]) + (uint32_t)(itpr - ilen)) & 0x0FFFFFFFu;
                        int cst = fl > 0 ? 0 : 1;
                        for (int q = cst; q < curr->n; ++q)
                            new_cigar[new_n++] = (q == 0) ? ORC_CIG(ORC_CIG_OP(curr->cig[0]), fl) : curr->cig[q];
                    }
                }
                if (!found_closure) return 0;
            } else if (dist_btw_two > 0 && dist_btw_two <= p->max_report_intron && bh_antisense2(prev) == curr->antisense) {
                /* :1311-1591 junction / deletion closure */
                if (!ref) return 0;
                int64_t lb = j_upper_bound(c, reference_id, (uint32_t)left_boundary, (uint32_t)(right_boundary - 8), 1);
                int64_t ub = j_lower_bound(c, reference_id, (uint32_t)(left_boundary + 8), (uint32_t)right_boundary, 0);
                int new_diff_mismatches = 0xff;
                for (; lb < ub && lb < c->n_juncs; ++lb) {
                    const orc_junction* j = &c->juncs[lb];
                    int dtl, dtr;
                    if (reversed) { dtl = (int)j->left - curr->left; dtr = (int)j->right - prev_right - 1; }
                    else { dtl = (int)j->left - prev_right + 1; dtr = (int)j->right - curr->left; }
                    if (!(abs(dtl) <= 4 && abs(dtr) <= 4 && dtl == dtr)) continue;
                    if ((reversed && (dtl > prev_right_end_match_length || -dtl > curr_left_end_match_length)) ||
                        (!reversed && (dtl > curr_left_end_match_length || -dtl > prev_right_end_match_length))) continue;
                    int new_mismatch = 0, old_mismatch = 0;
                    if (dtl > 0) {
                        for (int i = 0; i < dtl; ++i) {
                            char s, nc, oc;
                            if (reversed) {
                                /* new = rc(ref[curr->left+1, j->left+1)), old = rc(ref[prev_right+1, j->right));
                                   seq = read_seq.substr(curr_seg_index * L - dtl, dtl) */
                                s = rs_at(read_seq, read_len, curr_seg_index * L - dtl + i);
                                nc = refrc(ref, rlen, (int64_t)j->left - i);
                                oc = refrc(ref, rlen, (int64_t)j->right - 1 - i);
                            } else {
                                /* new = ref[prev_right, j->left+1), old = ref[curr->left, j->right) */
                                s = i < curr->seq_len ? curr->seq[i] : 0;
                                nc = refc(ref, rlen, (int64_t)prev_right + i);
                                oc = refc(ref, rlen, (int64_t)curr->left + i);
                            }
                            if (s != nc) ++new_mismatch;
                            if (s != oc) ++old_mismatch;
                        }
                    } else if (dtl < 0) {
                        int ad = -dtl;
                        for (int i = 0; i < ad; ++i) {
                            char s, nc, oc;
                            if (reversed) {
                                /* new = rc(ref[j->right, prev_right+1)), old = rc(ref[j->left+1, curr->left+1));
                                   seq = read_seq.substr(curr_seg_index * L, ad), indexed len - (ad - i) */
                                int avail = read_len - curr_seg_index * L; if (avail > ad) avail = ad; if (avail < 0) avail = 0;
                                s = rs_at(read_seq, read_len, curr_seg_index * L + avail - (ad - i));
                                nc = refrc(ref, rlen, (int64_t)prev_right - i);
                                oc = refrc(ref, rlen, (int64_t)curr->left - i);
                            } else {
                                /* new = ref[j->right, curr->left), old = ref[j->left+1, prev_right) */
                                s = prev->seq[prev->seq_len - (ad - i)];
                                nc = refc(ref, rlen, (int64_t)j->right + i);
                                oc = refc(ref, rlen, (int64_t)j->left + 1 + i);
                            }
                            if (s != nc) ++new_mismatch;
                            if (s != oc) ++old_mismatch;
                        }
                    }
                    int temp_diff = new_mismatch - old_mismatch;
                    if (temp_diff >= new_diff_mismatches || new_mismatch >= 2) continue;
                    new_diff_mismatches = temp_diff;
                    new_left = prev->left;
                    new_n = prev->n;
                    memcpy(new_cigar, prev->cig, sizeof(uint32_t) * (size_t)prev->n);
                    int nlb = (int)ORC_CIG_LEN(new_cigar[new_n - 1]);
                    int nrf = (int)ORC_CIG_LEN(curr->cig[0]);
                    if (reversed) { nlb -= dtl; nrf += dtr; } else { nlb += dtl; nrf -= dtr; }
                    if (nlb > 0) new_cigar[new_n - 1] = ORC_CIG(ORC_CIG_OP(new_cigar[new_n - 1]), (uint32_t)nlb);
                    else --new_n;
                    uint32_t skip = j->right - j->left - 1;
                    if (skip <= (uint32_t)p->max_deletion_length) {
                        new_cigar[new_n++] = ORC_CIG(reversed ? ORC_dEL : ORC_DEL, skip);
                        antisense_closure = bh_is_spliced(prev) ? prev->antisense_splice : curr->antisense_splice;
                    } else {
                        new_cigar[new_n++] = ORC_CIG(reversed ? ORC_rEF_SKIP : ORC_REF_SKIP, skip);
                        antisense_closure = (int)j->antisense;
                    }
                    int cst = nrf > 0 ? 0 : 1;
                    for (int q = cst; q < curr->n; ++q)
                        new_cigar[new_n++] = (q == 0) ? ORC_CIG(ORC_CIG_OP(curr->cig[0]), (uint32_t)nrf) : curr->cig[q];
                    mismatch = new_diff_mismatches;
                    found_closure = 1;
                }
                if (!found_closure) return 0;
            } else if (!(dist_btw_two == 0 && bh_antisense2(prev) == curr->antisense))
                check_fusion = 1;
        }

        if (check_fusion) {                                                     /* :1596-1818 */
            orc_fusion_in k1, k2;
            uint32_t ref_id1 = prev->ref_id2, ref_id2 = curr->ref_id;
            uint32_t fleft = (uint32_t)prev_right - 4u, fright = (uint32_t)curr->left - 4u;
            int reversed = 0;
            if (fusion_dir != FUS_FF && (ref_id2 < ref_id1 || (ref_id1 == ref_id2 && fleft > fright))) {
                reversed = 1;
                uint32_t t = ref_id1; ref_id1 = ref_id2; ref_id2 = t;
                t = fleft; fleft = fright; fright = t;
            }
            k1.ref1 = ref_id1; k1.ref2 = ref_id2; k1.left = fleft; k1.right = fright; k1.dir = FUS_FF;
            k2 = k1; k2.left = fleft + 8u; k2.right = fright + 8u;
            int64_t lb = f_upper_bound(c, &k1), ub = f_lower_bound(c, &k2);
            int64_t rlen1, rlen2;
            const char* ref1 = contig(c->g, prev->ref_id2, &rlen1);
            const char* ref2 = contig(c->g, curr->ref_id, &rlen2);
            int new_diff_mismatches = 0xff;
            for (; lb < ub && lb < c->n_fus; ++lb) {
                int lb_left = (int)c->fus[lb].left, lb_right = (int)c->fus[lb].right;
                if (reversed) { lb_left = (int)c->fus[lb].right; lb_right = (int)c->fus[lb].left; }
                int dtl, dtr;
                if (fusion_dir == FUS_RF) dtl = prev_right - lb_left + 1; else dtl = lb_left - prev_right + 1;
                if (fusion_dir == FUS_FR) dtr = curr->left - lb_right; else dtr = lb_right - curr->left;
                if (!(abs(dtl) <= 4 && abs(dtr) <= 4 && dtl == dtr)) continue;
                if (dtl > curr_left_end_match_length || -dtl > prev_right_end_match_length) continue;
                if (!ref1 || !ref2) return 0;
                int new_mismatch = 0, old_mismatch = 0;
                if (dtl > 0) {
                    for (int i = 0; i < dtl; ++i) {
                        char nc, oc, s;
                        if (fusion_dir == FUS_RF) nc = refrc(ref1, rlen1, (int64_t)prev_right - i);      /* rc(ref1[lb_left, prev_right+1)) */
                        else nc = refc(ref1, rlen1, (int64_t)prev_right + i);                            /* ref1[prev_right, lb_left+1) */
                        if (fusion_dir == FUS_FR) oc = refrc(ref2, rlen2, (int64_t)curr->left - i);      /* rc(ref2[lb_right+1, curr->left+1)) */
                        else oc = refc(ref2, rlen2, (int64_t)curr->left + i);                            /* ref2[curr->left, lb_right) */
                        if (fusion_dir == FUS_FF || fusion_dir == FUS_RR) s = i < curr->seq_len ? curr->seq[i] : 0;
                        else s = (i < L) ? rs_at(read_seq, read_len, curr_seg_index * L + i) : 0;
                        if (s != nc) ++new_mismatch;
                        if (s != oc) ++old_mismatch;
                    }
                } else if (dtl < 0) {
                    int ad = -dtl;
                    for (int i = 0; i < ad; ++i) {
                        char nc, oc, s;
                        if (fusion_dir == FUS_FR) nc = refrc(ref2, rlen2, (int64_t)lb_right - i);        /* rc(ref2[curr->left+1, lb_right+1)) */
                        else nc = refc(ref2, rlen2, (int64_t)lb_right + i);                              /* ref2[lb_right, curr->left) */
                        if (fusion_dir == FUS_RF) oc = refrc(ref1, rlen1, (int64_t)lb_left - 1 - i);     /* rc(ref1[prev_right+1, lb_left)) */
                        else oc = refc(ref1, rlen1, (int64_t)lb_left + 1 + i);                           /* ref1[lb_left+1, prev_right) */
                        if (fusion_dir == FUS_FF || fusion_dir == FUS_RR) s = prev->seq[prev->seq_len - (ad - i)];
                        else {
                            int st = (curr_seg_index - 1) * L;
                            int plen = read_len - st; if (plen > L) plen = L; if (plen < 0) plen = 0;
                            s = rs_at(read_seq, read_len, st + plen - (ad - i));
                        }
                        if (s != nc) ++new_mismatch;
                        if (s != oc) ++old_mismatch;
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
                new_cigar[new_n++] = ORC_CIG(fusion_dir, (uint32_t)lb_right);
                antisense_closure = bh_is_spliced(prev) ? prev->antisense_splice : curr->antisense_splice;
                int cst = nrf > 0 ? 0 : 1;
                for (int q = cst; q < curr->n; ++q)
                    new_cigar[new_n++] = (q == 0) ? ORC_CIG(ORC_CIG_OP(curr->cig[0]), (uint32_t)nrf) : curr->cig[q];
                mismatch = new_diff_mismatches;
                found_closure = 1;
            }
            if (!found_closure) return 0;
        }

        if (found_closure) {                                                    /* :1822-1870 */
            if (new_n > MAXC) return 0;  /* oracle capacity guard */
            BH m;
            memset(&m, 0, sizeof m);
            int mismatches = (int)prev->mm + (int)curr->mm + mismatch;
            m.insert_id = insert_id; m.ref_id = prev->ref_id; m.ref_id2 = curr->ref_id2; m.left = new_left;
            m.n = new_n; memcpy(m.cig, new_cigar, sizeof(uint32_t) * (size_t)new_n);
            m.antisense = antisense; m.antisense_splice = antisense_closure;
            m.mm = (unsigned char)mismatches;
            m.ed = (unsigned char)(mismatches + gap_length(new_cigar, new_n));
            m.end = 0;
            if (prev->seq_len + curr->seq_len > MAXSEQ) return 0;
            memcpy(m.seq, prev->seq, (size_t)prev->seq_len);
            memcpy(m.seq + prev->seq_len, curr->seq, (size_t)curr->seq_len);
            m.seq_len = prev->seq_len + curr->seq_len;
            chain[pi] = m;
            for (int q = ci; q + 1 < n; ++q) chain[q] = chain[q + 1];
            --n;
            ci = pi + 1;
            ++curr_seg_index;
            continue;
        }
        ++pi; ++ci; ++curr_seg_index;
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
    if (ln > MAXC) return 0;   /* oracle capacity guard */
    BH nh;
    memset(&nh, 0, sizeof nh);
    nh.insert_id = insert_id; nh.ref_id = chain[0].ref_id; nh.ref_id2 = chain[n - 1].ref_id2; nh.left = left;
    nh.n = ln; memcpy(nh.cig, lc, sizeof(uint32_t) * (size_t)ln);
    nh.antisense = antisense; nh.antisense_splice = saw_as;
    nh.mm = (unsigned char)num_mm;
    nh.ed = (unsigned char)(num_mm + gap_length(lc, ln));
    nh.end = 0;
    if (fusion_dir == FUS_NONE || fusion_dir == FUS_FF || fusion_dir == FUS_RR) {        /* :1959-1978 */
        memcpy(nh.seq, seq, (size_t)seq_len); nh.seq_len = seq_len;
        if (p->bowtie2) {
            int same = seq_len == read_len && memcmp(seq, read_seq, (size_t)read_len) == 0;
            for (int k = 0; k < read_len; ++k) nh.qual[k] = same ? read_qual[k] : read_qual[read_len - 1 - k];
        } else memcpy(nh.qual, qual, (size_t)seq_len);
    } else {                                                                             /* :1979-1983 */
        memcpy(nh.seq, read_seq, (size_t)read_len); nh.seq_len = read_len;
        memcpy(nh.qual, read_qual, (size_t)read_len);
    }
    int do_reverse = nh.ref_id > nh.ref_id2;                                             /* :1985-1999 */
    if (nh.ref_id == nh.ref_id2) {
        uint32_t fl, fr;
        if (first_fusion(&nh, &fl, &fr)) do_reverse = fl > fr;
    }
    if (do_reverse) { BH t; bh_reverse(&nh, &t); nh = t; }
    if (fusion_dir != FUS_NONE)                                                          /* :2007-2013 */
        nh.antisense = !(nh.seq_len == read_len && memcmp(nh.seq, read_seq, (size_t)read_len) == 0);
    if (bh_read_len(&nh) != old_read_length || !check_editdist(c, &nh)) return 0;        /* :2022-2034 */
    *out = nh;
    return 1;
}

/* valid_hit, long_spanning_reads.cpp:2045-2099 */
static int valid_hit(const fctx* c, const BH* bh)
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

/* merge_segment_chain, :2101-2220 */
static void merge_segment_chain(const fctx* c, const char* read_seq, const char* read_qual, int read_len, const BH* hits, int n, bhvec* joined,
                                int fusion_dir)
{
    const orc_span_params* p = c->p;
    if (n == 0) return;
    BH bh;
    if (n > 1) {
        BH chain[17];
        if (fusion_dir == FUS_NONE || fusion_dir == FUS_FF || fusion_dir == FUS_RR) {
            if (hits[0].antisense) for (int i = 0; i < n; ++i) chain[i] = hits[n - 1 - i];
            else for (int i = 0; i < n; ++i) chain[i] = hits[i];
        } else {
            int saw = 0, m = 0;
            for (int i = 0; i < n; ++i) {
                int pushed = 0;
                if (!saw && i > 0) {
                    if (hits[i - 1].ref_id != hits[i].ref_id) saw = 1;
                    else if (hits[i - 1].antisense != hits[i].antisense) saw = 1;
                    else {
                        int dist = hits[i].antisense ? hits[i - 1].left - bh_right(&hits[i]) : hits[i].left - bh_right(&hits[i - 1]);
                        if (dist >= p->max_report_intron || dist < -p->max_insertion_length) saw = 1;
                    }
                }
                if (bh_fusion_opcode(&hits[i]) == FUS_NONE &&
                    ((fusion_dir == FUS_FR && saw) || (fusion_dir == FUS_RF && !saw)) &&
                    hits[i].left < bh_right(&hits[i])) {
                    bh_reverse(&hits[i], &chain[m++]);
                    pushed = 1;
                }
                if (i > 0 && bh_fusion_opcode(&hits[i]) != FUS_NONE && hits[i].ref_id != hits[i - 1].ref_id) {
                    if (m >= 17) return;
                    bh_reverse(&hits[i], &chain[m++]);
                    pushed = 1;
                }
                if (!saw && bh_fusion_opcode(&hits[i]) != FUS_NONE) saw = 1;
                if (!pushed) { if (m >= 17) return; chain[m++] = hits[i]; }
            }
            n = m;
        }
        if (!merge_chain(c, read_seq, read_qual, read_len, chain, n, fusion_dir, &bh)) memset(&bh, 0, sizeof bh);
    } else {
        bh = hits[0];
        int do_reverse = bh.ref_id > bh.ref_id2;
        if (bh.ref_id == bh.ref_id2) {
            uint32_t fl, fr;
            if (first_fusion(&bh, &fl, &fr)) do_reverse = fl > fr;
        }
        if (do_reverse) { BH t; bh_reverse(&bh, &t); bh = t; }
    }
    if (valid_hit(c, &bh)) bhpush(joined, &bh);
}

typedef struct { const BH* v; int n; } seglist;

/* dfs_seg_hits, :2222-2610 */
static int dfs(const fctx* c, const char* read_seq, const char* read_qual, int read_len, const seglist* segs, int nsegs, int curr,
               BH* stack, int depth, bhvec* joined, int* num_try, int fusion_dir)
{
    const orc_span_params* p = c->p;
    const int fs = c->fusion_search;
    if (*num_try <= 0) return 0;
    int join_success = 0;
    if (curr < nsegs) {
        for (int i = 0; i < segs[curr].n; ++i) {
            BH bh = segs[curr].v[i];
            BH bh_prev = stack[depth - 1];
            BH* prevHit = &bh_prev;
            BH* currHit = &bh;
            int prev_fused = bh_fusion_opcode(prevHit) != FUS_NONE;
            int curr_fused = bh_fusion_opcode(currHit) != FUS_NONE;
            int num_fusions = (prev_fused ? 1 : 0) + (curr_fused ? 1 : 0);
            int dir = prev_fused ? bh_fusion_opcode(prevHit) : bh_fusion_opcode(currHit);
            if (!fs && num_fusions > 0) continue;
            if (num_fusions >= 2) continue;
            if (fusion_dir != FUS_NONE && curr_fused) continue;
            if (fusion_dir == FUS_FF || fusion_dir == FUS_RR) {
                if ((currHit->antisense && currHit->ref_id != prevHit->ref_id) ||
                    (!currHit->antisense && currHit->ref_id != prevHit->ref_id2)) continue;
            }
            if ((fusion_dir == FUS_FR || fusion_dir == FUS_RF) && prevHit->ref_id2 != currHit->ref_id) continue;
            if ((fusion_dir == FUS_FR && !currHit->antisense) || (fusion_dir == FUS_RF && currHit->antisense)) continue;
            BH t;
            if (curr_fused && dir == FUS_RR) { bh_reverse(currHit, &t); *currHit = t; }
            if (fusion_dir == FUS_FR || fusion_dir == FUS_RF ||
                (curr_fused && currHit->ref_id == currHit->ref_id2 && (dir == FUS_FR || dir == FUS_RF))) {
                if (curr_fused) {
                    if ((dir == FUS_FR && currHit->antisense) || (dir == FUS_RF && !currHit->antisense)) { bh_reverse(currHit, &t); *currHit = t; }
                } else {
                    if (fusion_dir == FUS_FR && currHit->antisense) { bh_reverse(currHit, &t); *currHit = t; }
                }
            } else if ((num_fusions == 0 && prevHit->antisense && currHit->antisense && prevHit->ref_id == currHit->ref_id &&
                        (!fs || (prevHit->left <= bh_right(currHit) + p->max_report_intron &&
                                 prevHit->left + p->max_insertion_length >= bh_right(currHit)))) ||
                       (num_fusions == 1 && (dir == FUS_FF || dir == FUS_RR) &&
                        ((!prev_fused && prevHit->antisense) || (!curr_fused && currHit->antisense)))) {
                BH* tmp = prevHit; prevHit = currHit; currHit = tmp;
            } else if (num_fusions == 0) {
                if (prevHit->ref_id2 == currHit->ref_id && prevHit->antisense == currHit->antisense) {
                    int dist = prevHit->antisense ? prevHit->left - bh_right(currHit) : currHit->left - bh_right(prevHit);
                    if (dist > p->max_report_intron || dist < -p->max_insertion_length) {
                        if ((prevHit->antisense && prevHit->left > currHit->left) || (!prevHit->antisense && prevHit->left < currHit->left))
                            dir = FUS_FF;
                        else
                            dir = FUS_RR;
                    }
                } else {
                    if (prevHit->antisense == currHit->antisense) {
                        if ((prevHit->antisense && prevHit->ref_id > currHit->ref_id) || (!prevHit->antisense && prevHit->ref_id < currHit->ref_id))
                            dir = FUS_FF;
                        else
                            dir = FUS_RR;
                    } else if (!prevHit->antisense) dir = FUS_FR;
                    else dir = FUS_RF;
                    if (dir == FUS_FR) { bh_reverse(currHit, &t); *currHit = t; }
                    else if (dir == FUS_RF) { bh_reverse(prevHit, &t); *prevHit = t; }
                }
            }
            if (!fs && dir != FUS_NONE) continue;
            if (num_fusions == 1) {                                              /* :2442-2514 */
                if (dir != FUS_FF && dir != FUS_RR) {
                    int prev_rep = 0, curr_rep = 0;
                    if (prev_fused) {
                        if ((dir == FUS_FR && !currHit->antisense) || (dir == FUS_RF && currHit->antisense)) continue;
                        if (prevHit->ref_id2 != currHit->ref_id) prev_rep = 1;
                        else if ((dir == FUS_FR && prevHit->antisense) || (dir == FUS_RF && !prevHit->antisense)) prev_rep = 1;
                    }
                    if (curr_fused) {
                        if ((dir == FUS_FR && prevHit->antisense) || (dir == FUS_RF && !prevHit->antisense)) continue;
                        if (currHit->ref_id != prevHit->ref_id2) curr_rep = 1;
                    }
                    if (prev_rep) { bh_reverse(prevHit, &t); *prevHit = t; }
                    if (curr_rep) { bh_reverse(currHit, &t); *currHit = t; }
                    prev_rep = 0; curr_rep = 0;
                    if (prev_fused) { if (bh_forwarding_right(prevHit) != bh_forwarding_left(currHit)) curr_rep = 1; }
                    else { if (bh_forwarding_right(prevHit) != bh_forwarding_left(currHit)) prev_rep = 1; }
                    if (prev_rep) { bh_reverse(prevHit, &t); *prevHit = t; }
                    if (curr_rep) { bh_reverse(currHit, &t); *currHit = t; }
                }
            }
            int same_contig = prevHit->ref_id2 == currHit->ref_id;
            if (!same_contig && num_fusions > 0) continue;
            if (!fs && (!same_contig || num_fusions > 0)) continue;
            if (same_contig && num_fusions >= 1 && bh_antisense2(prevHit) != currHit->antisense) continue;
            int bh_l = 0, back_right = 0, dist = 0;
            if (same_contig) {
                if ((fusion_dir == FUS_FR || fusion_dir == FUS_RF || dir == FUS_FR || dir == FUS_RF) && bh_antisense2(prevHit)) {
                    bh_l = bh_right(prevHit) + 1;
                    back_right = currHit->left + 1;
                } else {
                    bh_l = currHit->left;
                    back_right = bh_right(prevHit);
                }
                dist = bh_l - back_right;
            }
            if (!same_contig ||
                (same_contig && num_fusions == 0 && dir != FUS_NONE && fusion_dir == FUS_NONE) ||
                (same_contig && dist <= p->max_report_intron && dist >= -p->max_insertion_length &&
                 bh_forwarding_right(prevHit) == bh_forwarding_left(currHit))) {
                BH saved = stack[depth - 1];
                stack[depth - 1] = bh_prev;
                stack[depth] = bh;
                if (dfs(c, read_seq, read_qual, read_len, segs, nsegs, curr + 1, stack, depth + 1, joined, num_try,
                        dir == FUS_NONE ? fusion_dir : dir))
                    join_success = 1;
                stack[depth - 1] = saved;
                if (*num_try <= 0) return join_success;
            }
        }
    } else {
        --*num_try;
        merge_segment_chain(c, read_seq, read_qual, read_len, stack, depth, joined, fusion_dir);
        return 1;
    }
    return join_success;
}

/* BowtieHit::operator< (bwt_map.h:180-207) */
static int bh_less(const BH* a, const BH* b)
{
    if (a->insert_id != b->insert_id) return a->insert_id < b->insert_id;
    if (a->ref_id != b->ref_id) return a->ref_id < b->ref_id;
    if (a->ref_id2 != b->ref_id2) return a->ref_id2 < b->ref_id2;
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
    if (a->insert_id != b->insert_id || a->ref_id != b->ref_id || a->ref_id2 != b->ref_id2 || a->antisense != b->antisense ||
        a->left != b->left || a->antisense_splice != b->antisense_splice || a->ed != b->ed || a->n != b->n) return 0;
    return memcmp(a->cig, b->cig, sizeof(uint32_t) * (size_t)a->n) == 0;
}

/* bowtie_sam_extra, bwt_map.cpp:2467-2648 */
static void sam_extra(const fctx* c, const BH* bh, orc_faln* o)
{
    const orc_span_params* p = c->p;
    int64_t rlen1, rlen2;
    const char* ref1 = contig(c->g, bh->ref_id, &rlen1);
    const char* ref2 = contig(c->g, bh->ref_id2, &rlen2);
    o->AS = o->XM = o->XO = o->XG = 0; o->md[0] = 0;
    if (!ref1 || !ref2) return;
    const char* ref = ref1; int64_t rlen = rlen1;
    size_t pos_seq = 0, pos_mismatch = 0, mismatch = 0, opens = 0, conts = 0;
    int64_t pos_ref = bh->left;
    int AS = 0, saw_fusion = 0;
    char md[2048]; int ml = 0;
    const int qual_len = bh->seq_len;
    for (int i = 0; i < bh->n; ++i) {
        int op = ORC_CIG_OP(bh->cig[i]);
        int64_t len = ORC_CIG_LEN(bh->cig[i]);
        if (op == ORC_MATCH || op == ORC_mATCH) {
            for (int64_t j = 0; j < len; ++j) {
                char r = op == ORC_MATCH ? refc(ref, rlen, pos_ref + j) : refrc(ref, rlen, pos_ref - j);
                char s = pos_seq < (size_t)bh->seq_len ? bh->seq[pos_seq] : 'N';
                if (d5(s) != r) {
                    ++mismatch;
                    if (pos_seq < (size_t)qual_len) {
                        if (d5(s) == 'N' || r == 'N') AS -= p->bowtie2_penalty_for_N;
                        else {
                            int q = bh->qual[pos_seq] - '!'; if (q > 40) q = 40;
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
                if (ml > 1900) break;
            }
            if (op == ORC_MATCH) pos_ref += len; else pos_ref -= len;
        } else if (op == ORC_INS || op == ORC_iNS) {
            pos_seq += (size_t)len;
            AS -= p->bowtie2_read_gap_open; AS -= (int)(p->bowtie2_read_gap_cont * len);
            opens += 1; conts += (size_t)len;
        } else if (op == ORC_DEL || op == ORC_dEL) {
            AS -= p->bowtie2_ref_gap_open; AS -= (int)(p->bowtie2_ref_gap_cont * len);
            opens += 1; conts += (size_t)len;
            ml += sprintf(md + ml, "%d^", (int)pos_mismatch);
            for (int64_t k = 0; k < len && ml < 600; ++k)
                md[ml++] = op == ORC_DEL ? refc(ref, rlen, pos_ref + k) : refrc(ref, rlen, pos_ref - k);
            if (op == ORC_DEL) pos_ref += len; else pos_ref -= len;
            pos_mismatch = 0;
        } else if (op == ORC_REF_SKIP) pos_ref += len;
        else if (op == ORC_rEF_SKIP) pos_ref -= len;
        else if (is_fusion_op(op)) {
            if (saw_fusion) return;
            ref = ref2; rlen = rlen2;
            pos_ref = len;
            saw_fusion = 1;
        }
        if (ml > 1900) break;
    }
    ml += sprintf(md + ml, "%d", (int)pos_mismatch);
    md[ml] = 0;
    o->AS = AS; o->XM = (int)mismatch; o->XO = (int)opens; o->XG = (int)conts;
    if (ml < (int)sizeof o->md) strcpy(o->md, md);
    else snprintf(o->md, sizeof o->md, "\x01%lld", (long long)orc_long_md_put(md));
}

int orc_spanning_batch_fusion(const orc_span_params* p, int fusion_search, int fusion_min_dist, const orc_genome* g, const orc_span_batch* b,
                              const orc_junction* juncs, int64_t n_juncs, const orc_ins_in* ins, int64_t n_ins,
                              const orc_fusion_in* fusions, int64_t n_fusions, orc_faln** out, int64_t* n_out)
{
    fctx c; c.p = p; c.fusion_search = fusion_search; c.fusion_min_dist = fusion_min_dist; c.g = g;
    c.juncs = juncs; c.n_juncs = n_juncs; c.ins = ins; c.n_ins = n_ins; c.fus = fusions; c.n_fus = n_fusions;
    orc_faln* res = NULL; int64_t nres = 0, cap = 0;
    const int L = p->segment_length;
    for (int r = 0; r < b->n_reads; ++r) {
        const int64_t* so = b->seg_off + (int64_t)r * b->nseg;
        const char* rseq = b->bases + b->read_off[r];
        const char* rqual = b->quals + b->read_off[r];
        int rl = (int)(b->read_off[r + 1] - b->read_off[r]);
        if (so[1] == so[0]) continue;                    /* the worker iterates over segment-1 groups only (:2706-2765) */
        int nsegs = 0;                                   /* look_right_for_hit_group stops at the first segment without hits */
        while (nsegs < b->nseg && so[nsegs + 1] > so[nsegs]) ++nsegs;
        const orc_span_hit* lastfirst = &b->hits[so[nsegs - 1]];
        if (!(lastfirst->flags & ORC_HIT_END)) continue; /* :2777-2785 */
        if (rl > MAXSEQ || nsegs > 16) continue;
        seglist segs[16]; BH* store[16];
        int multihit_drop = 0;
        for (int s = 0; s < nsegs; ++s) {
            int n = (int)(so[s + 1] - so[s]);
            if (p->bowtie2 && n > p->max_seg_multihits) multihit_drop = 1;      /* :2625-2632 */
            store[s] = (BH*)calloc((size_t)n, sizeof(BH));
            for (int k = 0; k < n; ++k) {
                const orc_span_hit* h = &b->hits[so[s] + k];
                BH* x = &store[s][k];
                x->insert_id = (uint32_t)r + 1;
                x->ref_id = h->ref_id; x->ref_id2 = h->ref_id; x->left = h->left; x->n = h->n_cigar;
                memcpy(x->cig, h->cigar, sizeof(uint32_t) * (size_t)h->n_cigar);
                for (int q = 0; q < h->n_cigar; ++q)
                    if (is_fusion_op(ORC_CIG_OP(h->cigar[q]))) x->ref_id2 = h->cigar[4];   /* fused hits: <= 4 ops, cigar[4] = ref_id2 */
                x->antisense = (h->flags & ORC_HIT_ANTISENSE) != 0;
                x->antisense_splice = (h->flags & ORC_HIT_ANTISENSE_SPLICE) != 0;
                x->end = (h->flags & ORC_HIT_END) != 0;
                x->mm = h->mismatches; x->ed = h->edit_dist;
                /* the segment record's SEQ / QUAL: the read piece, reversed when the RECORD is on the reverse strand -- which is
                   antisense_align except for hits on rf / rr fusion contigs, whose orientation was flipped (bwt_map.cpp:1744-1745) */
                int rec_rev = x->antisense ^ ((h->flags & ORC_HIT_STRAND_FLIPPED) != 0);
                int st = s * L; if (st > rl) st = rl;
                int ln = x->end ? rl - st : L; if (ln > rl - st) ln = rl - st;
                for (int q = 0; q < ln; ++q) {
                    x->seq[q] = rec_rev ? comp(rseq[st + ln - 1 - q]) : rseq[st + q];
                    x->qual[q] = rec_rev ? rqual[st + ln - 1 - q] : rqual[st + q];
                }
                x->seq_len = ln;
            }
            segs[s].v = store[s]; segs[s].n = n;
        }
        bhvec joined = {0, 0, 0};
        if (!multihit_drop) {
            BH stack[18];
            for (int i = 0; i < segs[0].n; ++i) {                                /* :2634-2664 */
                if (bh_fusion_opcode(&segs[0].v[i]) == FUS_RR) bh_reverse(&segs[0].v[i], &stack[0]);
                else stack[0] = segs[0].v[i];
                int num_try = 10000;
                dfs(&c, rseq, rqual, rl, segs, nsegs, 1, stack, 1, &joined, &num_try, FUS_NONE);
            }
        }
        for (int i = 1; i < joined.n; ++i) {             /* sort + unique (:2805-2807); insertion sort is stable like libstdc++ below 16 */
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
            if (nres == cap) { cap = cap ? cap * 2 : 1024; res = (orc_faln*)realloc(res, sizeof(orc_faln) * (size_t)cap); }
            orc_faln* o = &res[nres++];
            memset(o, 0, sizeof *o);
            o->read_idx = r; o->ref_id = h->ref_id; o->ref_id2 = h->ref_id2; o->left = h->left;
            o->antisense = (uint8_t)h->antisense; o->antisense_splice = (uint8_t)h->antisense_splice;
            o->mismatches = h->mm; o->edit_dist = h->ed;
            o->n_cigar = h->n; memcpy(o->cigar, h->cig, sizeof(uint32_t) * (size_t)h->n);
            sam_extra(&c, h, o);
        }
        free(joined.v);
        for (int s = 0; s < nsegs; ++s) free(store[s]);
    }
    *out = res; *n_out = nres;
    return 0;
}
