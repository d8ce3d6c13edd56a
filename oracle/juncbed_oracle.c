/* juncbed_oracle.c -- TEST INFRASTRUCTURE (see thj_oracle.h): plain-C restatement of the junction consensus of
 * tophat_reports, the step that turns spliced alignments into junctions.bed.
 *
 *   junctions_from_spliced_hit + junctions_from_alignment   junctions.cpp:19-142   (per REF_SKIP: junction key, extents)
 *   JunctionStats::merge_with                               junctions.h:87-101     (max extents, summed support)
 *   accept_if_valid, knockout_shadow_junctions, filter_junctions   junctions.cpp:192-330
 *   the two passes of the driver                            tophat_reports.cpp:2845 (filter after the first pass),
 *       :1194-1230 exclude_hits_on_filtered_junctions (second pass), :2974-2984 (drop support 0 / extents < 8), :2991 print
 *   print_junction                                          junctions.cpp:100-120
 *
 * Not restated (tophat_reports' alignment selection, outside the hot path): read_best_alignments, realign_reads, pair
 * grading -- every record handed in counts as a reported alignment.  splice_mms is 0 for records parsed from BAM
 * (bwt_map.cpp:1176), so max_splice_mismatches never rejects.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "thj_oracle.h"

typedef struct { uint32_t ref, left, right, anti; int le, re, support, accepted; } jent;

static int jcmp(const void* a, const void* b) {                 /* Junction::operator<, junctions.h:39-57 */
    const jent* x = (const jent*)a; const jent* y = (const jent*)b;
    if (x->ref != y->ref) return x->ref < y->ref ? -1 : 1;
    if (x->left != y->left) return x->left < y->left ? -1 : 1;
    if (x->right != y->right) return x->right < y->right ? -1 : 1;
    if (x->anti != y->anti) return x->anti < y->anti ? -1 : 1;
    return 0;
}

/* junctions of one record appended to out[] (<= 8); returns their number.  junctions_from_spliced_hit (junctions.cpp:19-92) with its
 * fusion branches: pieces that run down the genome (lower-case ops mATCH 2, dEL 6, rEF_SKIP 12) walk backwards and swap the extents;
 * a fusion op (FF 7, FR 8, RF 9) jumps to its length = the position on the second contig and the junctions behind it belong to that
 * contig (the record keeps it in cigar[15], as thj_aln does); FUSION_RR (10) has no case there -- the walk neither jumps nor switches
 * contigs -- and none here. */
static int rec_juncs(const orc_jrec* r, jent* out) {
    int n = 0;
    int64_t j = r->left;
    int saw_fusion = 0;
    for (int c = 0; c < r->n_cigar; ++c) {
        const uint32_t op = r->cigar[c] >> 28, len = r->cigar[c] & 0x0FFFFFFFu;
        if (op == 11 || op == 12) {                             /* REF_SKIP, rEF_SKIP */
            jent e; memset(&e, 0, sizeof e);
            const int prev = c > 0 ? (int)(r->cigar[c - 1] & 0x0FFFFFFFu) : 0, next = c + 1 < r->n_cigar ? (int)(r->cigar[c + 1] & 0x0FFFFFFFu) : 0;
            e.ref = saw_fusion ? r->cigar[15] : r->ref_id; e.anti = r->antisense_splice ? 1u : 0u;
            if (op == 11) { e.left = (uint32_t)(j - 1); e.right = (uint32_t)(j + len); e.le = prev; e.re = next; j += len; }
            else { e.right = (uint32_t)(j + 1); e.left = (uint32_t)(j - len); e.re = prev; e.le = next; j -= len; }
            e.support = 1;
            if (n < 8) out[n++] = e;
        } else if (op == 1 || op == 5) j += len;                /* MATCH, DEL */
        else if (op == 2 || op == 6) j -= len;                  /* mATCH, dEL */
        else if (op == 7 || op == 8 || op == 9) { j = len; saw_fusion = 1; }
    }
    return n;
}

/* reduce: sort all (junction, stats) pairs, merge equal keys */
static int64_t reduce(jent* v, int64_t n) {
    qsort(v, (size_t)n, sizeof *v, jcmp);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (m && jcmp(&v[m - 1], &v[i]) == 0) {
            if (v[i].le > v[m - 1].le) v[m - 1].le = v[i].le;
            if (v[i].re > v[m - 1].re) v[m - 1].re = v[i].re;
            v[m - 1].support += v[i].support;
        } else v[m++] = v[i];
    }
    return m;
}

static jent* find(jent* v, int64_t n, const jent* key) { return (jent*)bsearch(key, v, (size_t)n, sizeof *v, jcmp); }

int orc_junction_consensus(const orc_jrec* recs, int64_t n_recs, int min_anchor_len, orc_jstat** out, int64_t* n_out) {
    jent* a = (jent*)malloc(sizeof(jent) * (size_t)(n_recs * 8 + 1));
    jent* b = (jent*)malloc(sizeof(jent) * (size_t)(n_recs * 8 + 1));
    if (!a || !b) return -1;
    /* ---- first pass: the junction set of all alignments, then filter_junctions */
    int64_t na = 0;
    for (int64_t i = 0; i < n_recs; ++i) na += rec_juncs(&recs[i], a + na);
    na = reduce(a, na);
    for (int64_t i = 0; i < na; ++i) {                           /* accept_if_valid */
        const int mn = a[i].le < a[i].re ? a[i].le : a[i].re;
        if (mn < min_anchor_len) a[i].accepted = 0;
        else if ((int)a[i].right - (int)a[i].left > 50000) a[i].accepted = a[i].support >= 2 && mn > 12;
        else a[i].accepted = 1;
    }
    for (int64_t i = 0; i < na; ++i) {                           /* knockout_shadow_junctions */
        if (!a[i].accepted) continue;
        /* candidates: junctions of the other strand between (left - anchor, .., !anti) and (.., right + anchor, !anti) in set order */
        for (int64_t k = 0; k < na; ++k) {
            if (k == i || a[k].ref != a[i].ref || a[k].anti == a[i].anti) continue;
            jent lo = a[i], hi = a[i];
            lo.left -= (uint32_t)min_anchor_len; lo.anti = !a[i].anti;
            hi.right += (uint32_t)min_anchor_len; hi.anti = !a[i].anti;
            if (jcmp(&a[k], &lo) < 0 || jcmp(&a[k], &hi) > 0) continue;      /* [lower_bound(fuzzy_left), upper_bound(fuzzy_right)) */
            const int left_diff = (int)a[i].left - (int)a[k].left, right_diff = (int)a[i].right - (int)a[k].right;
            if ((left_diff < min_anchor_len || right_diff < min_anchor_len) && a[i].support < a[k].support) a[i].accepted = 0;
        }
    }
    /* NB the reference's inner loop reads itr->second.accepted of junctions it may already have knocked out; it only ever
       clears the flag of the junction being visited (itr), and the test does not look at the neighbour's flag, so the order
       of visits does not matter */
    /* ---- second pass: alignments on a filtered junction are excluded; the rest make the final set */
    int64_t nb = 0;
    for (int64_t i = 0; i < n_recs; ++i) {
        jent t[8];
        const int n = rec_juncs(&recs[i], t);
        int ok = 1;
        for (int k = 0; k < n; ++k) { const jent* f = find(a, na, &t[k]); if (!f || !f->accepted) ok = 0; }
        if (!ok) continue;
        for (int k = 0; k < n; ++k) b[nb++] = t[k];
    }
    nb = reduce(b, nb);
    orc_jstat* o = (orc_jstat*)malloc(sizeof(orc_jstat) * (size_t)(nb + 1));
    if (!o) return -1;
    int64_t m = 0;
    for (int64_t i = 0; i < nb; ++i) {
        if (b[i].support == 0 || b[i].le < 8 || b[i].re < 8) continue;       /* tophat_reports.cpp:2974-2984 */
        o[m].ref_id = b[i].ref; o[m].left = b[i].left; o[m].right = b[i].right; o[m].antisense = b[i].anti;
        o[m].left_extent = (uint32_t)b[i].le; o[m].right_extent = (uint32_t)b[i].re; o[m].support = (uint32_t)b[i].support; o[m].reserved = 0;
        ++m;
    }
    free(a); free(b);
    *out = o; *n_out = m;
    return 0;
}

/* print_junctions (junctions.cpp:100-120, :330-350) -> malloc'd text */
char* orc_junctions_bed(const orc_jstat* j, int64_t n, const char* const* names) {
    size_t cap = 256 + (size_t)n * 256, len = 0;
    char* s = (char*)malloc(cap);
    if (!s) return NULL;
    len += (size_t)sprintf(s + len, "track name=junctions description=\"TopHat junctions\"\n");
    for (int64_t i = 0; i < n; ++i) {
        const int lp1 = (int)j[i].left + 1, start = lp1 - (int)j[i].left_extent, end = (int)j[i].right + (int)j[i].right_extent;
        len += (size_t)snprintf(s + len, cap - len, "%s\t%d\t%d\tJUNC%08d\t%d\t%c\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n", names[j[i].ref_id - 1], start, end,
                                (int)(i + 1), (int)j[i].support, j[i].antisense ? '-' : '+', start, end, (int)j[i].left_extent, (int)j[i].right_extent,
                                (int)j[i].right - start);
    }
    return s;
}
