/*
 * segjuncs_oracle.c -- CPU oracle (TEST INFRASTRUCTURE, see thj_oracle.h) for
 * the split-segment junction / small-indel search of TopHat's segment_juncs.
 *
 * Plain-C restatement of DaehwanKimLab/tophat v2.1.2 src/segment_juncs.cpp:
 *   find_gaps                         :3293-3650
 *   juncs_from_ref_segs<RecordSegmentJuncs> (POINT_DIR_BOTH)  :2052-2377, :1669-1696
 *   find_insertions_and_deletions     :2807-2942
 *   detect_small_insertion/deletion   :2470-2627
 *   simpleSplitAlignment              :2390-2456
 *   map_read_to_contig                :2946-2973
 * Colour-space (`if (color)`) branches are out of scope and omitted.
 *
 * PARITY: the junction search is pinned by the reference's regression case
 * test_SimpleSplicing (tests/golden_ref/); the indel search, the rescue and
 * the fusion search are unpinned by the reference's own tests; see
 * oracle/README.md for what they were checked against.
 */
#include "thj_oracle.h"
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

/* ---------------------------------------------------------------- utils */

typedef struct { orc_junction* v; int64_t n, cap; } jvec;
typedef struct { orc_insertion* v; int64_t n, cap; } ivec;

static int jpush(jvec* a, orc_junction j)
{
    if (a->n == a->cap) {
        int64_t nc = a->cap ? a->cap * 2 : 1024;
        orc_junction* nv = (orc_junction*)realloc(a->v, (size_t)nc * sizeof *nv);
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = j;
    return 0;
}

static int ipush(ivec* a, orc_insertion j)
{
    if (a->n == a->cap) {
        int64_t nc = a->cap ? a->cap * 2 : 1024;
        orc_insertion* nv = (orc_insertion*)realloc(a->v, (size_t)nc * sizeof *nv);
        if (!nv) return -1;
        a->v = nv; a->cap = nc;
    }
    a->v[a->n++] = j;
    return 0;
}

/* Junction::operator< (junctions.h:39-57): refid, left, right as uint32, then
 * antisense (false before true). skip_count is always 0 for RecordSegmentJuncs
 * so skip_count_lt (junctions.h:70-78) reduces to this. */
static int jcmp(const void* pa, const void* pb)
{
    const orc_junction* a = (const orc_junction*)pa;
    const orc_junction* b = (const orc_junction*)pb;
    if (a->ref_id != b->ref_id) return a->ref_id < b->ref_id ? -1 : 1;
    if (a->left != b->left) return a->left < b->left ? -1 : 1;
    if (a->right != b->right) return a->right < b->right ? -1 : 1;
    if (a->antisense != b->antisense) return a->antisense < b->antisense ? -1 : 1;
    return 0;
}

static int64_t junique(orc_junction* v, int64_t n)
{
    if (n == 0) return 0;
    qsort(v, (size_t)n, sizeof *v, jcmp);
    int64_t w = 1;
    for (int64_t i = 1; i < n; ++i)
        if (jcmp(&v[i], &v[w - 1]) != 0) v[w++] = v[i];
    return w;
}

/* Insertion::operator< (insertions.h:52-67) compares refid, left and only the
 * LENGTH of the sequence, so std::set keeps the first one inserted among
 * equal-length insertions at one place.  `prio` is the insertion order. */
static int icmp(const void* pa, const void* pb)
{
    const orc_insertion* a = (const orc_insertion*)pa;
    const orc_insertion* b = (const orc_insertion*)pb;
    if (a->ref_id != b->ref_id) return a->ref_id < b->ref_id ? -1 : 1;
    if (a->left != b->left) return a->left < b->left ? -1 : 1;
    size_t la = strlen(a->seq), lb = strlen(b->seq);
    if (la != lb) return la < lb ? -1 : 1;
    if (a->prio != b->prio) return a->prio < b->prio ? -1 : 1;
    return 0;
}

static int64_t iunique(orc_insertion* v, int64_t n)
{
    if (n == 0) return 0;
    qsort(v, (size_t)n, sizeof *v, icmp);
    int64_t w = 1;
    for (int64_t i = 1; i < n; ++i) {
        const orc_insertion* p = &v[w - 1];
        if (v[i].ref_id == p->ref_id && v[i].left == p->left && strlen(v[i].seq) == strlen(p->seq))
            continue; /* an earlier (lower prio) one is already kept */
        v[w++] = v[i];
    }
    return w;
}

static const char* contig(const orc_genome* g, uint32_t ref_id, int64_t* len)
{
    if (ref_id == 0 || (int64_t)ref_id > g->n_contigs) { *len = 0; return NULL; }
    *len = g->len[ref_id - 1];
    return g->seq[ref_id - 1];
}

/* reads.cpp:189-207 reverse_complement(string&): anything not ACGT becomes N. */
static void revcomp(const char* in, int n, char* out)
{
    for (int i = 0; i < n; ++i) {
        char c = in[n - 1 - i], o;
        switch (c) {
        case 'A': o = 'T'; break;
        case 'T': o = 'A'; break;
        case 'C': o = 'G'; break;
        case 'G': o = 'C'; break;
        default:  o = 'N'; break;
        }
        out[i] = o;
    }
}

/* SeqAn Dna5 -> Dna conversion used by the window copies: value & 3, so N (4)
 * becomes A (SeqAn-1.4.2/seqan/basic/alphabet_residue.h:873-876, used at
 * segment_juncs.cpp:2157 and :2499). */
static char n_to_a(char c) { return c == 'N' ? 'A' : c; }

/* ------------------------------------------- juncs_from_ref_segs (BOTH) */

static const char* const DONORS[3]    = { "GT", "GC", "AT" };  /* segment_juncs.cpp:3618-3649 */
static const char* const ACCEPTORS[3] = { "AG", "AG", "AC" };

static void rc2(const char* d, char* o) { revcomp(d, 2, o); }

int orc_window_scan(const orc_params* p, const orc_genome* g, uint32_t ref_id,
                    int32_t seg_left, int32_t seg_right, int antisense,
                    const char* support, int support_len,
                    orc_junction* out, int cap)
{
    int n_out = 0;
    int64_t ref_len;
    const char* ref = contig(g, ref_id, &ref_len);
    if (!ref) return 0;                                   /* :2105-2108 */

    /* library-type strand rules, :2110-2138 */
    int skip_fwd = 0, skip_rev = 0;
    if (p->library_type == 2) {            /* FR_FIRSTSTRAND */
        if (p->read_side == 1) { if (antisense) skip_rev = 1; else skip_fwd = 1; }
        else if (p->read_side == 2) { if (antisense) skip_fwd = 1; else skip_rev = 1; }
    }
    if (p->library_type == 3) {            /* FR_SECONDSTRAND */
        if (p->read_side == 1) { if (antisense) skip_fwd = 1; else skip_rev = 1; }
        else if (p->read_side == 2) { if (antisense) skip_rev = 1; else skip_fwd = 1; }
    }

    /* :2154 */
    if (seg_left < 0 || seg_right >= (int)ref_len - 1) return 0;

    int seg_len = seg_right - seg_left;
    int read_len = support_len;
    /* The reference indexes seg_str[read_len-2] and seg_str[seg_len-read_len+i]
     * unguarded; with the distances find_gaps admits seg_len >= read_len + min
     * intron always holds.  Guarded here so the oracle has no UB. */
    if (read_len < 2 || read_len > 128 || seg_len < read_len) return 0;

    for (int m = 0; m < 3; ++m) {
        const char* donor = DONORS[m];
        const char* acceptor = ACCEPTORS[m];
        char rev_donor[2], rev_acceptor[2];
        rc2(donor, rev_donor);                 /* :2073-2076 */
        rc2(acceptor, rev_acceptor);

        int to = read_len - 2;                 /* :2167-2172 */
        uint8_t left_mm[128], right_mm[128];
        memset(left_mm, 0, sizeof left_mm);
        memset(right_mm, 0, sizeof right_mm);

        /* :2187-2203 prefix mismatches of the support read against the window start */
        {
            int num = 0;
            for (int i = 0; i < read_len - 1; ++i) {
                if (n_to_a(ref[seg_left + i]) != support[i]) ++num;
                left_mm[i] = (uint8_t)num;
                if (num > 2) { to = i; break; }
            }
        }
        /* :2205-2218 suffix mismatches against the window end; entries below the
         * break index stay 0 */
        {
            int num = 0;
            for (int i = read_len - 1; i >= 0; --i) {
                if (n_to_a(ref[seg_left + i + (seg_len - read_len)]) != support[i]) ++num;
                right_mm[i] = (uint8_t)num;
                if (num > 2) break;
            }
        }

        /* :2240-2289 */
        for (int i = 0; i <= to; ++i) {
            char c0 = n_to_a(ref[seg_left + i]), c1 = n_to_a(ref[seg_left + i + 1]);
            int is_don = (c0 == donor[0] && c1 == donor[1]);
            int is_racc = (c0 == rev_acceptor[0] && c1 == rev_acceptor[1]);
            if ((!skip_fwd && is_don) || (!skip_rev && is_racc)) {
                const char* partner = is_don ? acceptor : rev_donor;
                int lm = i > 0 ? left_mm[i - 1] : 0;
                if (lm + right_mm[i] <= 2) {
                    int pos = seg_len - (read_len - i) - 2;
                    char p0 = n_to_a(ref[seg_left + pos]), p1 = n_to_a(ref[seg_left + pos + 1]);
                    if (p0 == partner[0] && p1 == partner[1]) {
                        /* RecordSegmentJuncs::record :1681-1695: Junction(ref,
                         * donor-1, acceptor+2, antisense); fwd lists -> '+',
                         * rev lists -> '-'.  In an all-BOTH call the two lists
                         * grow together, so the size check at :1681 never
                         * drops anything and motifs.unique() is skipped
                         * (:2339-2340): each accepted i is one junction. */
                        if (n_out < cap) {
                            orc_junction j;
                            j.ref_id = ref_id;
                            j.left = (uint32_t)(seg_left + i - 1);
                            j.right = (uint32_t)(seg_left + pos + 2);
                            j.antisense = is_don ? 0u : 1u;
                            out[n_out++] = j;
                        }
                    }
                }
            }
        }
    }
    return n_out;
}

/* -------------------------------------------------- simpleSplitAlignment */

int orc_simple_split(const char* shorter, const char* left_ref, const char* right_ref,
                     int len, int* min_err)
{
    /* segment_juncs.cpp:2390-2456 */
    unsigned short before[512], after[512];
    if (len > 512) len = 512;
    for (int idx = len - 1; idx >= 0; --idx) {
        unsigned short prev = idx < len - 1 ? before[idx + 1] : 0;
        unsigned short cur = (right_ref[idx] == 'N' || shorter[idx] == 'N' || right_ref[idx] != shorter[idx]) ? 1 : 0;
        before[idx] = (unsigned short)(prev + cur);
    }
    for (int idx = 0; idx < len; ++idx) {
        unsigned short prev = idx > 0 ? after[idx - 1] : 0;
        unsigned short cur = (left_ref[idx] == 'N' || shorter[idx] == 'N' || left_ref[idx] != shorter[idx]) ? 1 : 0;
        after[idx] = (unsigned short)(prev + cur);
    }
    int best = len + 1, best_pos = -1;
    for (int pos = 1; pos < len; ++pos) {
        int e = before[pos] + after[pos - 1];
        if (e < best) { best = e; best_pos = pos; }   /* ties keep the first (bestInsertPositions[0]) */
    }
    *min_err = best;
    return best_pos;
}

/* ----------------------------------------------------- map_read_to_contig */

int orc_map_read_to_contig(const char* ctg, int contig_len, const char* read, int read_len)
{
    /* segment_juncs.cpp:2946-2973 */
    int pos = -1, mismatch = 3;
    for (int i = 0; i < contig_len - read_len; ++i) {
        int t = 0;
        for (int j = 0; j < read_len; ++j) {
            if (ctg[i + j] != read[j]) ++t;
            if (t >= mismatch) break;
        }
        if (t < mismatch) { pos = i; mismatch = t; }
    }
    return pos;
}

/* --------------------------------------------- small insertions/deletions */

typedef struct {
    const orc_params* p;
    const orc_genome* g;
    jvec juncs, dels;
    ivec ins;
    int64_t n_windows, n_indel_pairs, n_rescue_pairs;
    uint64_t prio;
} ctx_t;

static int is_anti(const orc_hit* h) { return (h->flags & ORC_HIT_ANTISENSE) != 0; }
static int is_end(const orc_hit* h) { return (h->flags & ORC_HIT_END) != 0; }

/* detect_small_deletion, segment_juncs.cpp:2557-2627 */
static void detect_small_deletion(ctx_t* c, const char* rd, int read_length,
                                  const orc_hit* lh, const orc_hit* rh)
{
    int64_t ref_len;
    const char* ref = contig(c->g, lh->ref_id, &ref_len);
    if (!ref) return;
    if (lh->left < 0) return;
    if (rh->right < read_length) return;
    int discrepancy = (rh->right - lh->left) - read_length;
    /* seqan::infix clamps to the string end; shorter than read_length -> return (:2589) */
    if ((int64_t)lh->left + read_length > ref_len) return;
    if ((int64_t)rh->right > ref_len) return;
    const char* lg = ref + lh->left;                 /* Dna5: N stays N */
    const char* rg = ref + rh->right - read_length;
    int min_err = -1;
    int pos = orc_simple_split(rd, lg, rg, read_length, &min_err);
    if (pos < 0) return;  /* reference asserts size>0; read_length>=2 always here */
    int adjustment = 0;
    if ((int)lh->read_len + (int)rh->read_len >= read_length) adjustment = -1;
    if (min_err <= (int)lh->edit_dist + (int)rh->edit_dist + adjustment) {
        orc_junction d;
        d.ref_id = lh->ref_id;
        d.left = (uint32_t)(lh->left + pos - 1);
        d.right = (uint32_t)(lh->left + pos + discrepancy);
        d.antisense = 0;
        jpush(&c->dels, d);
    }
}

/* detect_small_insertion, segment_juncs.cpp:2470-2543 */
static void detect_small_insertion(ctx_t* c, const char* rd, int read_length,
                                   const orc_hit* lh, const orc_hit* rh)
{
    int64_t ref_len;
    const char* ref = contig(c->g, lh->ref_id, &ref_len);
    if (!ref) return;
    if (lh->left < 0) return;
    int discrepancy = read_length - (rh->right - lh->left);
    int64_t gb = lh->left, ge = rh->right;
    if (ge > ref_len) ge = ref_len;              /* infix clamps */
    int glen = (int)(ge - gb);
    if (glen < 0) glen = 0;
    if (glen > read_length) return;              /* cannot happen (discrepancy>0); guard */
    char genomic[512];
    if (glen > 512) return;
    for (int i = 0; i < glen; ++i) genomic[i] = n_to_a(ref[gb + i]);   /* DnaString: N->A (:2499) */
    const char* left_read = rd;                          /* infix(read, 0, glen) */
    const char* right_read = rd + read_length - glen;    /* infix(read, read_length-glen, read_length) */
    int min_err = -1;
    int pos = orc_simple_split(genomic, left_read, right_read, glen, &min_err);
    if (pos < 0) return;                                 /* :2514-2515 */
    int adjustment = 0;
    if ((int)lh->read_len + (int)rh->read_len >= read_length) adjustment = -1;
    if (min_err <= (int)lh->edit_dist + (int)rh->edit_dist + adjustment &&
        pos + discrepancy <= glen) {
        orc_insertion ins;
        memset(&ins, 0, sizeof ins);
        ins.ref_id = lh->ref_id;
        ins.left = (uint32_t)(lh->left + pos - 1);
        int n = discrepancy < 15 ? discrepancy : 15;
        memcpy(ins.seq, left_read + pos, (size_t)n);
        ins.seq[n] = 0;
        ins.prio = c->prio++;
        ipush(&c->ins, ins);
    }
}

/* find_insertions_and_deletions, segment_juncs.cpp:2807-2942 */
static void find_indels(ctx_t* c, const orc_batch* b, int r)
{
    int nseg = b->nseg;
    if (nseg < 2) return;                                   /* :2815-2818 */
    const int64_t* so = b->seg_off + (int64_t)r * nseg;
    const char* seq = b->bases + b->read_off[r];
    int seq_len = (int)(b->read_off[r + 1] - b->read_off[r]);
    int L = c->p->segment_length;

    for (int i = 0; i + 2 < nseg; ++i) {                    /* :2856: i < size-2 */
        int64_t lb = so[i], le = so[i + 1], rb = so[i + 1], re = so[i + 2];
        if (lb == le || rb == re) return;                   /* :2869-2870 (return, not continue) */

        /* read.seq.substr(i*L, 2L), :2881 */
        char full[512], rc[512];
        int start = i * L;
        if (start > seq_len) return;  /* std::substr would throw; unreachable for valid input */
        int plen = seq_len - start < 2 * L ? seq_len - start : 2 * L;
        if (plen > 512) plen = 512;
        memcpy(full, seq + start, (size_t)plen);
        revcomp(full, plen, rc);      /* seqan::reverseComplement on String<char>, :2883 */

        for (int64_t li = lb; li < le; ++li)
            for (int64_t ri = rb; ri < re; ++ri) {
                const orc_hit* lh = &b->hits[li];
                const orc_hit* rh = &b->hits[ri];
                if (lh->ref_id != rh->ref_id) continue;          /* :2901 */
                if (is_anti(lh) != is_anti(rh)) continue;        /* :2904 */
                const char* mod = full;
                if (is_anti(lh)) { const orc_hit* t = lh; lh = rh; rh = t; mod = rc; }  /* :2914-2920 */
                int apparent = rh->right - lh->left;
                int disc = apparent - plen;
                if (disc > 0 && disc <= c->p->max_deletion_length) {
                    c->n_indel_pairs++;
                    detect_small_deletion(c, mod, plen, lh, rh);
                }
                if (disc < 0 && disc >= -c->p->max_insertion_length) {
                    c->n_indel_pairs++;
                    detect_small_insertion(c, mod, plen, lh, rh);
                }
            }
    }
}

/* ---------------------------------------------------------------- find_gaps */

#define MAXH 4096

typedef struct { orc_hit* v; int n, cap; } hlist;

static void hl_push(hlist* l, orc_hit h)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 16;
        l->v = (orc_hit*)realloc(l->v, (size_t)l->cap * sizeof(orc_hit));
    }
    l->v[l->n++] = h;
}

/* find_gaps, segment_juncs.cpp:3293-3650 */
static void find_gaps(ctx_t* c, const orc_batch* b, int r)
{
    const orc_params* p = c->p;
    int nseg = b->nseg;
    if (nseg == 0) return;
    const int64_t* so = b->seg_off + (int64_t)r * nseg;
    const char* seq = b->bases + b->read_off[r];
    int seq_len = (int)(b->read_off[r + 1] - b->read_off[r]);
    int L = p->segment_length;

    /* local, mutable copy of hits_for_read */
    hlist* segs = (hlist*)calloc((size_t)nseg, sizeof(hlist));
    for (int s = 0; s < nseg; ++s)
        for (int64_t k = so[s]; k < so[s + 1]; ++k) hl_push(&segs[s], b->hits[k]);

    int last_segment = nseg - 1;                       /* :3304-3313 */
    while (last_segment > 0) {
        if (segs[last_segment].n) break;
        --last_segment;
    }
    int size = last_segment + 1;                       /* hits_for_read.resize */
    int first_segment = 0;
    if (last_segment == first_segment &&
        (segs[0].n == 0 || is_end(&segs[0].v[0])))     /* :3316-3318 */
        goto done;

    /* :3320-3348 partner lookup: the mate's full-read hits if present, else the
     * mate's last-segment hits; both streams are consumed in id order, so this
     * is an id join -- the batch carries the result as mate_hits. */
    {
        int has_partner = 0;
        const orc_hit* mate = NULL; int n_mate = 0;
        if (b->mate_off) {
            n_mate = (int)(b->mate_off[r + 1] - b->mate_off[r]);
            mate = b->mate_hits + b->mate_off[r];
            has_partner = n_mate > 0;
        }

        /* :3359-3393 is there an intra-read partner between first and last segment? */
        int check_partner = 1;
        if (first_segment != last_segment) {
            for (int i = 0; i < segs[0].n && check_partner; ++i) {
                const orc_hit* lh = &segs[0].v[i];
                for (int j = 0; j < segs[last_segment].n; ++j) {
                    const orc_hit* rh = &segs[last_segment].v[j];
                    if (lh->ref_id == rh->ref_id && is_anti(lh) == is_anti(rh)) {
                        int dist = is_anti(lh) ? lh->left - rh->right : rh->left - lh->right;
                        if (dist >= p->min_segment_intron && dist < p->max_segment_intron) {
                            check_partner = 0;
                            break;
                        }
                    }
                }
            }
        }

        if (check_partner && has_partner) {                 /* :3395-3492 mate-anchored rescue */
            for (int s = first_segment + 1; s < size; ++s) segs[s].n = 0;   /* :3398-3401 */
            char rcread[1024];
            int rl = seq_len < 1024 ? seq_len : 1024;
            revcomp(seq, rl, rcread);
            /* NB the reference pushes into hits_for_read[last_segment] while
             * iterating left_segment_hits = hits_for_read[0]; when
             * last_segment == 0 those are the same vector, but then s == size-1
             * for every hit below and no window is produced, so the pushes are
             * unobservable.  We iterate over the original seg-0 count only. */
            int n_left = segs[0].n;
            int check_read_len = 15 < L - p->segment_mismatches - 3 ? 15 : L - p->segment_mismatches - 3; /* :3451 */
            for (int l = 0; l < n_left; ++l) {
                orc_hit lh = segs[0].v[l];
                for (int m = 0; m < n_mate; ++m) {
                    const orc_hit* rh = &mate[m];
                    if (lh.ref_id != rh->ref_id || is_anti(&lh) == is_anti(rh)) continue;   /* :3414 */
                    /* :3423 `dist < min && dist >= max` is never true: no filter */
                    int64_t ref_len;
                    const char* ref = contig(c->g, rh->ref_id, &ref_len);
                    if (!ref) continue; /* reference would dereference NULL; unreachable for consistent input */
                    int part_seq_len = p->inner_dist_std_dev > p->inner_dist_mean ? p->inner_dist_std_dev - p->inner_dist_mean : 0;
                    int flanking_seq_len = p->inner_dist_mean + p->inner_dist_std_dev;
                    int64_t left;
                    if (is_anti(rh)) {                                            /* :3431-3440 */
                        if (flanking_seq_len <= rh->left) left = rh->left - flanking_seq_len;
                        else break;
                    } else {                                                       /* :3441-3450 */
                        if (part_seq_len <= rh->right) left = rh->right - part_seq_len;
                        else break;
                    }
                    int64_t fe = left + flanking_seq_len + part_seq_len;
                    if (fe > ref_len) fe = ref_len;                               /* infix clamps */
                    int flen = (int)(fe - left);
                    if (flen < 0) flen = 0;
                    if (check_read_len < 1 || check_read_len > rl) continue;     /* guard: infix would be invalid */
                    const char* fwd_read = seq + rl - check_read_len;             /* :3452 */
                    const char* rev_read = rcread;                                /* :3453 */
                    c->n_rescue_pairs++;
                    /* Dna5String -> String<char>: N stays 'N' and equals a read 'N' (:2958) */
                    int fwd_pos = orc_map_read_to_contig(ref + left, flen, fwd_read, check_read_len);
                    if (fwd_pos >= 0) {                                            /* :3456-3462 */
                        orc_hit h; memset(&h, 0, sizeof h);
                        h.ref_id = rh->ref_id; h.left = (int32_t)(left + fwd_pos);
                        h.right = h.left + check_read_len; h.flags = ORC_HIT_END;
                        h.read_len = (uint8_t)check_read_len;
                        hl_push(&segs[last_segment], h);
                    }
                    int rev_pos = orc_map_read_to_contig(ref + left, flen, rev_read, check_read_len);
                    if (rev_pos >= 0) {                                            /* :3464-3472 */
                        orc_hit h; memset(&h, 0, sizeof h);
                        h.ref_id = rh->ref_id; h.left = (int32_t)(left + rev_pos);
                        h.right = h.left + check_read_len; h.flags = ORC_HIT_END | ORC_HIT_ANTISENSE;
                        h.read_len = (uint8_t)check_read_len;
                        hl_push(&segs[last_segment], h);
                    }
                }
            }
        }
    }

    /* :3499-3506 multihit cap (bowtie2 only) */
    if (p->bowtie2)
        for (int s = 0; s < size; ++s)
            if (segs[s].n > p->max_seg_multihits) goto done;

    /* :3508-3617 */
    for (int s = 0; s < size; ++s) {
        for (int h = 0; h < segs[s].n; ++h) {
            int found_right_seg_partner = (s == size - 1);
            const orc_hit* bh = &segs[s].v[h];
            const orc_hit* drs[MAXH]; int n_drs = 0;
            const orc_hit* rrs[MAXH]; int n_rrs = 0;

            if (s < size - 1) {
                for (int k = 0; k < segs[s + 1].n; ++k) {
                    const orc_hit* rh = &segs[s + 1].v[k];
                    if (is_anti(bh) != is_anti(rh) || bh->ref_id != rh->ref_id) continue;
                    if ((is_anti(bh) && rh->right == bh->left) ||
                        (!is_anti(bh) && bh->right == rh->left)) {
                        found_right_seg_partner = 1;
                        break;
                    }
                    int dist = is_anti(bh) ? bh->left - rh->right : rh->left - bh->right;
                    if (dist >= p->min_segment_intron && dist < p->max_segment_intron && n_drs < MAXH)
                        drs[n_drs++] = rh;
                }
            }
            if (!found_right_seg_partner && s < size - 2) {
                for (int k = 0; k < segs[s + 2].n; ++k) {
                    const orc_hit* rrh = &segs[s + 2].v[k];
                    if (is_anti(bh) != is_anti(rrh) || bh->ref_id != rrh->ref_id) continue;
                    int dist = is_anti(bh) ? bh->left - rrh->right : rrh->left - bh->right;
                    if (dist >= p->min_segment_intron + L && dist < p->max_segment_intron + L && n_rrs < MAXH)
                        rrs[n_rrs++] = rrh;
                }
            }
            if (!found_right_seg_partner && (n_drs > 0 || n_rrs > 0)) {
                const int look_bp = 8;
                const orc_hit** d = n_rrs > 0 ? rrs : drs;
                int nd = n_rrs > 0 ? n_rrs : n_drs;
                for (int k = 0; k < nd; ++k) {
                    /* seq.substr((s+1)*L - 8, 16 [+L]) :3583-3586 */
                    int start = (s + 1) * L - look_bp;
                    int want = n_rrs <= 0 ? look_bp * 2 : L + look_bp * 2;
                    if (start < 0 || start > seq_len) continue;   /* std::substr would throw */
                    int slen = seq_len - start < want ? seq_len - start : want;
                    char support[256], tmp[256];
                    if (slen > 255) slen = 255;
                    memcpy(support, seq + start, (size_t)slen);
                    int32_t wl, wr;
                    if (!is_anti(bh)) {                          /* :3589-3594 */
                        wl = bh->right - look_bp; if (wl < 0) wl = 0;
                        wr = d[k]->left + look_bp;
                    } else {                                      /* :3596-3604 */
                        revcomp(support, slen, tmp);
                        memcpy(support, tmp, (size_t)slen);
                        wl = d[k]->right - look_bp;
                        wr = bh->left + look_bp;
                    }
                    c->n_windows++;
                    orc_junction tmpj[3 * 128];
                    int n = orc_window_scan(p, c->g, bh->ref_id, wl, wr, is_anti(bh), support, slen, tmpj, 3 * 128);
                    for (int q = 0; q < n; ++q) jpush(&c->juncs, tmpj[q]);
                }
            }
        }
    }

done:
    for (int s = 0; s < nseg; ++s) free(segs[s].v);
    free(segs);
}

/* ------------------------------------------------------------------ batch */

int orc_segjuncs_batch(const orc_params* p, const orc_genome* g, const orc_batch* b, orc_events* out)
{
    ctx_t c;
    memset(&c, 0, sizeof c);
    c.p = p; c.g = g;
    memset(out, 0, sizeof *out);
    for (int r = 0; r < b->n_reads; ++r) {
        /* process_next_hit_group :4094-4118 (and the orphan branch :3996-4011):
         * indels first, then gaps, on the same hits_for_read */
        find_indels(&c, b, r);
        find_gaps(&c, b, r);
    }
    out->n_juncs = junique(c.juncs.v, c.juncs.n);       out->juncs = c.juncs.v;
    out->n_deletions = junique(c.dels.v, c.dels.n);     out->deletions = c.dels.v;
    out->n_insertions = iunique(c.ins.v, c.ins.n);      out->insertions = c.ins.v;
    out->n_windows = c.n_windows;
    out->n_indel_pairs = c.n_indel_pairs;
    out->n_rescue_pairs = c.n_rescue_pairs;
    return 0;
}

void orc_events_free(orc_events* e)
{
    free(e->juncs); free(e->deletions); free(e->insertions);
    memset(e, 0, sizeof *e);
}

/* ================================================================== fusion search */

typedef struct { orc_fusion* v; int64_t n, cap; } fvec;
static void fpush(fvec* a, orc_fusion f)
{
    if (a->n == a->cap) { a->cap = a->cap ? a->cap * 2 : 256; a->v = (orc_fusion*)realloc(a->v, (size_t)a->cap * sizeof(orc_fusion)); }
    a->v[a->n++] = f;
}
/* Fusion::operator< (fusions.h:38-69) */
static int fcmp(const void* pa, const void* pb)
{
    const orc_fusion* a = (const orc_fusion*)pa; const orc_fusion* b = (const orc_fusion*)pb;
    if (a->ref_id1 != b->ref_id1) return a->ref_id1 < b->ref_id1 ? -1 : 1;
    if (a->ref_id2 != b->ref_id2) return a->ref_id2 < b->ref_id2 ? -1 : 1;
    if (a->left != b->left) return a->left < b->left ? -1 : 1;
    if (a->right != b->right) return a->right < b->right ? -1 : 1;
    if (a->dir != b->dir) return a->dir < b->dir ? -1 : 1;
    return 0;
}

/* detect_fusion, segment_juncs.cpp:2629-2805.  rd = the whole read (reverse-complemented by the caller when
 * both hits are antisense). */
static void detect_fusion(const orc_genome* g, int fusion_anchor_length, const char* rd, int read_length,
                          const orc_hit* lh, const orc_hit* rh, uint32_t dir, fvec* out)
{
    int64_t llen, rlen;
    const char* lref = contig(g, lh->ref_id, &llen);
    const char* rref = contig(g, rh->ref_id, &rlen);
    if (!lref || !rref || read_length > 1024) return;
    char lg[1024], rg[1024], tmp[1024];
    if (dir == ORC_FUSION_FF || dir == ORC_FUSION_FR) {
        if (lh->left + read_length > (int)llen) return;
        if (lh->left < 0) return;
        memcpy(lg, lref + lh->left, (size_t)read_length);
    } else {
        if (lh->right < read_length) return;
        if (lh->right > llen) return;
        memcpy(tmp, lref + lh->right - read_length, (size_t)read_length);
        revcomp(tmp, read_length, lg);                   /* Dna5: N stays N */
    }
    if (dir == ORC_FUSION_FF || dir == ORC_FUSION_RF) {
        if (rh->right < read_length) return;
        if (rh->right > rlen) return;
        memcpy(rg, rref + rh->right - read_length, (size_t)read_length);
    } else {
        if (rh->left + read_length > (int)rlen) return;
        if (rh->left < 0) return;
        memcpy(tmp, rref + rh->left, (size_t)read_length);
        revcomp(tmp, read_length, rg);
    }
    /* simpleSplitAlignment, all tied best positions (:2390-2456) */
    unsigned short before[1024], after[1024];
    for (int idx = read_length - 1; idx >= 0; --idx) {
        unsigned short prev = idx < read_length - 1 ? before[idx + 1] : 0;
        before[idx] = (unsigned short)(prev + ((rg[idx] == 'N' || rd[idx] == 'N' || rg[idx] != rd[idx]) ? 1 : 0));
    }
    for (int idx = 0; idx < read_length; ++idx) {
        unsigned short prev = idx > 0 ? after[idx - 1] : 0;
        after[idx] = (unsigned short)(prev + ((lg[idx] == 'N' || rd[idx] == 'N' || lg[idx] != rd[idx]) ? 1 : 0));
    }
    int min_err = read_length + 1;
    for (int pos = 1; pos < read_length; ++pos) { int e = before[pos] + after[pos - 1]; if (e < min_err) min_err = e; }
    uint32_t total_ed = (uint32_t)lh->edit_dist + (uint32_t)rh->edit_dist;
    if (min_err > (int)total_ed) return;                                         /* :2697-2699 */
    if (min_err > 2) return;
    for (int pos = 1; pos < read_length; ++pos) {                                /* :2704-2713: any tied position too close to an end */
        if (before[pos] + after[pos - 1] != min_err) continue;
        if (pos < fusion_anchor_length) return;
        if (read_length - pos < fusion_anchor_length) return;
    }
    for (int pos = 1; pos < read_length; ++pos) {
        if (before[pos] + after[pos - 1] != min_err) continue;
        uint32_t left, right;
        if (dir == ORC_FUSION_FF || dir == ORC_FUSION_FR) left = (uint32_t)(lh->left + pos - 1);
        else left = (uint32_t)(lh->right - pos);
        if (dir == ORC_FUSION_FF || dir == ORC_FUSION_RF) right = (uint32_t)(rh->right - (read_length - pos));
        else right = (uint32_t)(rh->left + (read_length - pos) - 1);
        uint32_t r1 = lh->ref_id, r2 = rh->ref_id, tdir = dir;
        if (r2 < r1 || (r1 == r2 && left > right)) {                             /* :2776-2789 */
            uint32_t t = r1; r1 = r2; r2 = t;
            t = left; left = right; right = t;
            if (dir == ORC_FUSION_FF) tdir = ORC_FUSION_RR;
        }
        orc_fusion f; f.ref_id1 = r1; f.ref_id2 = r2; f.left = left; f.right = right; f.dir = tdir;
        f.count = 1; f.edit_dist = total_ed; f.skip = 0;
        fpush(out, f);
    }
}

/* find_fusions, segment_juncs.cpp:2976-3291 */
static int ref_ignored(const uint32_t* ignore, int n_ignore, uint32_t ref)
{
    for (int i = 0; i < n_ignore; ++i) if (ignore[i] == ref) return 1;
    return 0;
}

static void find_fusions(const orc_params* p, int fusion_anchor_length, int fusion_min_dist, const orc_genome* g,
                         const orc_batch* b, int r, const uint32_t* ignore, int n_ignore, fvec* out)
{
    int nseg = b->nseg;
    if (nseg == 0) return;
    const int64_t* so = b->seg_off + (int64_t)r * nseg;
    const char* seq = b->bases + b->read_off[r];
    int read_length = (int)(b->read_off[r + 1] - b->read_off[r]);
    int last = nseg - 1;
    while (last > 0 && so[last + 1] == so[last]) --last;                        /* :2986-2993 */
    int n0 = (int)(so[1] - so[0]);
    if (last == 0 && (n0 == 0 || is_end(&b->hits[so[0]]))) return;               /* :3035-3037 */
    const orc_hit* mate = NULL; int n_mate = 0;
    if (b->mate_off) { n_mate = (int)(b->mate_off[r + 1] - b->mate_off[r]); mate = b->mate_hits + b->mate_off[r]; }
    int has_partner = n_mate > 0;
    if (read_length > 1024) return;
    char rc[1024];
    revcomp(seq, read_length, rc);
    /* right_segment_hits = copy of the last segment's hits when first != last, else empty (:3075-3080) */
    hlist right = {0, 0, 0};
    if (last != 0) for (int64_t k = so[last]; k < so[last + 1]; ++k) hl_push(&right, b->hits[k]);
    int check_partner = 1;
    if (last != 0) {
        for (int64_t i = so[0]; i < so[1] && check_partner; ++i)
            for (int j = 0; j < right.n; ++j) {
                const orc_hit* lh = &b->hits[i]; const orc_hit* rh = &right.v[j];
                if (lh->ref_id == rh->ref_id && is_anti(lh) == is_anti(rh)) {
                    int dist = is_anti(lh) ? lh->left - rh->right : rh->left - lh->right;
                    if (dist > -p->max_insertion_length && dist <= fusion_min_dist) { check_partner = 0; break; }
                }
            }
    }
    const int minus_dist = -p->max_insertion_length * 2;
    if (check_partner && has_partner) {                                          /* :3117-3202 */
        int check_read_len = 15 < p->segment_length - p->segment_mismatches - 3 ? 15 : p->segment_length - p->segment_mismatches - 3;
        for (int64_t l = so[0]; l < so[1]; ++l) {
            const orc_hit* lh = &b->hits[l];
            for (int m = 0; m < n_mate; ++m) {
                const orc_hit* rh = &mate[m];
                if (lh->ref_id == rh->ref_id && is_anti(lh) != is_anti(rh)) {
                    int dist = is_anti(lh) ? lh->left - rh->right : rh->left - lh->right;
                    if (dist > minus_dist && dist <= fusion_min_dist) continue;
                }
                int64_t ref_len;
                const char* ref = contig(g, rh->ref_id, &ref_len);
                if (!ref) continue;
                int part = p->inner_dist_std_dev > p->inner_dist_mean ? p->inner_dist_std_dev - p->inner_dist_mean : 0;
                int flank = p->inner_dist_mean + p->inner_dist_std_dev;
                int64_t left;
                if (is_anti(rh)) { if (flank <= rh->left) left = rh->left - flank; else break; }
                else { if (part <= rh->right) left = rh->right - part; else break; }
                int64_t fe = left + flank + part; if (fe > ref_len) fe = ref_len;
                int flen = (int)(fe - left); if (flen < 0) flen = 0;
                if (check_read_len < 1 || check_read_len > read_length) continue;
                int fwd_pos = orc_map_read_to_contig(ref + left, flen, seq + read_length - check_read_len, check_read_len);
                if (fwd_pos >= 0) {
                    orc_hit h; memset(&h, 0, sizeof h);
                    h.ref_id = rh->ref_id; h.left = (int32_t)(left + fwd_pos); h.right = h.left + check_read_len;
                    h.flags = ORC_HIT_END; h.read_len = (uint8_t)check_read_len;
                    hl_push(&right, h);
                }
                int rev_pos = orc_map_read_to_contig(ref + left, flen, rc, check_read_len);
                if (rev_pos >= 0) {
                    orc_hit h; memset(&h, 0, sizeof h);
                    h.ref_id = rh->ref_id; h.left = (int32_t)(left + rev_pos); h.right = h.left + check_read_len;
                    h.flags = ORC_HIT_END | ORC_HIT_ANTISENSE; h.read_len = (uint8_t)check_read_len;
                    hl_push(&right, h);
                }
            }
        }
    }
    for (int64_t li = so[0]; li < so[1]; ++li)                                   /* :3211-3290 */
        for (int ri = 0; ri < right.n; ++ri) {
            const orc_hit* lh = &b->hits[li]; const orc_hit* rh = &right.v[ri];
            /* --fusion-ignore-chromosomes (:3214-3231) */
            if (ref_ignored(ignore, n_ignore, lh->ref_id) || ref_ignored(ignore, n_ignore, rh->ref_id)) continue;
            if (p->bowtie2 && (int)lh->edit_dist + (int)rh->edit_dist > (p->segment_mismatches << 1)) continue;
            if (lh->ref_id == rh->ref_id && is_anti(lh) == is_anti(rh)) {
                int dist = is_anti(lh) ? lh->left - rh->right : rh->left - lh->right;
                if (dist > minus_dist && dist <= fusion_min_dist) continue;
            }
            uint32_t dir = ORC_FUSION_FF;
            const char* mod = seq;
            if (is_anti(lh) == is_anti(rh)) {
                if (is_anti(lh)) { const orc_hit* t = lh; lh = rh; rh = t; mod = rc; }
            } else if (!is_anti(lh) && is_anti(rh)) dir = ORC_FUSION_FR;
            else dir = ORC_FUSION_RF;
            detect_fusion(g, fusion_anchor_length, mod, read_length, lh, rh, dir, out);
        }
    free(right.v);
}

int orc_fusions_batch(const orc_params* p, int fusion_anchor_length, int fusion_min_dist,
                      const orc_genome* g, const orc_batch* b, const uint32_t* ignore_ref_ids, int n_ignore,
                      orc_fusion** out, int64_t* n_out)
{
    fvec ev = {0, 0, 0};
    for (int r = 0; r < b->n_reads; ++r) find_fusions(p, fusion_anchor_length, fusion_min_dist, g, b, r, ignore_ref_ids, n_ignore, &ev);
    /* FusionSimpleSet: count occurrences, keep the smallest edit distance (:2791-2803) */
    if (ev.n) qsort(ev.v, (size_t)ev.n, sizeof(orc_fusion), fcmp);
    int64_t w = 0;
    for (int64_t i = 0; i < ev.n; ++i) {
        if (w && fcmp(&ev.v[w - 1], &ev.v[i]) == 0) {
            ev.v[w - 1].count += 1;
            if (ev.v[i].edit_dist < ev.v[w - 1].edit_dist) ev.v[w - 1].edit_dist = ev.v[i].edit_dist;
        } else ev.v[w++] = ev.v[i];
    }
    *out = ev.v; *n_out = w;
    return 0;
}

static int coord_found(const orc_junction* j, int64_t n, uint32_t ref, uint32_t coord)
{
    /* binary_search over the sorted (refid, coord) list built from every junction's left and right (:5048-5054, :5063) */
    for (int64_t i = 0; i < n; ++i)
        if (j[i].ref_id == ref && ((int)j[i].left == (int)coord || (int)j[i].right == (int)coord)) return 1;
    return 0;
}

void orc_fusion_filter(orc_fusion* f, int64_t n, const orc_junction* juncs, int64_t n_juncs)
{
    /* segment_juncs.cpp:5096-5159 */
    uint8_t* lc = (uint8_t*)calloc((size_t)n + 1, 1); uint8_t* rc_ = (uint8_t*)calloc((size_t)n + 1, 1);
    for (int64_t i = 0; i < n; ++i) {
        lc[i] = (uint8_t)coord_found(juncs, n_juncs, f[i].ref_id1, f[i].left);
        rc_[i] = (uint8_t)coord_found(juncs, n_juncs, f[i].ref_id2, f[i].right);
    }
    for (int64_t i = 0; i < n; ++i) {
        for (int64_t k = i + 1; k < n; ++k) {
            int left_diff = abs((int)f[i].left - (int)f[k].left);
            if (!(f[i].ref_id1 == f[k].ref_id1 && f[i].ref_id2 == f[k].ref_id2 && left_diff < 10)) break;
            if (f[i].dir == f[k].dir && left_diff == abs((int)f[i].right - (int)f[k].right)) {
                if (f[k].count > f[i].count) f[i].skip = 1;
                else if (f[k].count == f[i].count) {
                    int cc = lc[i] + rc_[i], nc = lc[k] + rc_[k];
                    if (cc < nc) f[i].skip = 1; else f[k].skip = 1;
                } else f[k].skip = 1;
            }
        }
    }
    free(lc); free(rc_);
}

/* ===================== juncs_db (juncs_db.cpp:73-233, :298-525) =====================
 * FASTA text of the junction database: junctions (std::set order: refid, left, right, '+' before '-'), deletions,
 * insertions (first of equal (refid, left, length) wins), fusions.  Test infrastructure like the rest of this file. */
typedef struct { char* p; size_t n, cap; } sbuf;
static void sb_put(sbuf* b, const char* s, size_t n)
{
    if (b->n + n + 1 > b->cap) { b->cap = (b->n + n + 1) * 2 + 256; b->p = (char*)realloc(b->p, b->cap); }
    memcpy(b->p + b->n, s, n); b->n += n; b->p[b->n] = 0;
}
static void sb_printf_ll(sbuf* b, long long v) { char t[32]; int n = snprintf(t, sizeof t, "%lld", v); sb_put(b, t, (size_t)n); }
static void sb_printf_ull(sbuf* b, unsigned long long v) { char t[32]; int n = snprintf(t, sizeof t, "%llu", v); sb_put(b, t, (size_t)n); }
static char d5c(char c) { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 'N'; }
static void sb_ref(sbuf* b, const char* ref, int64_t s, int64_t e, int rc)
{
    if (!rc) { for (int64_t i = s; i < e; ++i) { char c = d5c(ref[i]); sb_put(b, &c, 1); } }
    else for (int64_t i = e - 1; i >= s; --i) {
        char c = d5c(ref[i]);
        c = c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
        sb_put(b, &c, 1);
    }
}

/* print_splice :120-164 */
static void jd_splice(sbuf* b, const orc_genome* g, const char* const* names, uint32_t ref_id, uint32_t left, uint32_t right,
                      int read_len, const char* tag)
{
    int64_t ref_len; const char* ref = contig(g, ref_id, &ref_len);
    if (!ref) return;
    int half = read_len;
    if (!((int64_t)left <= ref_len && (int64_t)right <= ref_len)) return;     /* the >= 0 tests are vacuous on unsigned fields */
    int64_t left_start = (int)left - half + 1 >= 0 ? (int)left - half + 1 : 0;
    int64_t left_end = left_start + half;
    int64_t right_start = right;
    int64_t right_end = right_start + half < ref_len ? right_start + half : ref_len;
    if (!(left_start < left_end && left_end <= ref_len && right_start < right_end && right_end <= ref_len)) return;
    sb_put(b, ">", 1); sb_put(b, names[ref_id - 1], strlen(names[ref_id - 1])); sb_put(b, "|", 1); sb_printf_ll(b, left_start);
    sb_put(b, "|", 1); sb_printf_ll(b, left); sb_put(b, "-", 1); sb_printf_ll(b, right); sb_put(b, "|", 1); sb_printf_ll(b, right_end);
    sb_put(b, "|", 1); sb_put(b, tag, strlen(tag)); sb_put(b, "\n", 1);
    sb_ref(b, ref, left_start, left_end, 0); sb_ref(b, ref, right_start, right_end, 0); sb_put(b, "\n", 1);
}

/* juncs[] / dels[] / ins[] / fus[] must already be in their std::set orders with duplicates removed (the caller's
 * parser does what the reference's fgets loops + set inserts do); dels carry left = file_left - 1 (:385). */
char* orc_juncs_db(const orc_genome* g, const char* const* names, int read_len, int min_anchor_len,
                   const orc_junction* juncs, int64_t n_juncs, const orc_junction* dels, int64_t n_dels,
                   const uint32_t* ins_ref, const uint32_t* ins_left, const char* const* ins_seq, int64_t n_ins,
                   const orc_fusion* fus, int64_t n_fus)
{
    sbuf b = {0, 0, 0};
    sb_put(&b, "", 0);
    for (int64_t i = 0; i < n_juncs; ++i)
        jd_splice(&b, g, names, juncs[i].ref_id, juncs[i].left, juncs[i].right, read_len, juncs[i].antisense ? "GTAG|rev" : "GTAG|fwd");
    for (int64_t i = 0; i < n_dels; ++i)
        jd_splice(&b, g, names, dels[i].ref_id, dels[i].left, dels[i].right, read_len, dels[i].antisense ? "del|rev" : "del|fwd");
    for (int64_t i = 0; i < n_ins; ++i) {                                     /* print_insertion :73-108 */
        int64_t ref_len; const char* ref = contig(g, ins_ref[i], &ref_len);
        if (!ref) continue;
        int half = read_len - min_anchor_len;
        uint32_t left = ins_left[i];
        if (!((int64_t)left <= ref_len)) continue;
        int64_t left_start = (int)left - half + 1 >= 0 ? (int)left - half + 1 : 0;
        int64_t left_end = left_start + half;
        int64_t right_start = left_end;
        int64_t right_end = right_start + half < ref_len ? right_start + half : ref_len;
        if (!(left_start < left_end && left_end <= ref_len && right_start < right_end && right_end <= ref_len)) continue;
        const char* nm = names[ins_ref[i] - 1];
        sb_put(&b, ">", 1); sb_put(&b, nm, strlen(nm)); sb_put(&b, "|", 1); sb_printf_ll(&b, left_start); sb_put(&b, "|", 1);
        sb_printf_ll(&b, left); sb_put(&b, "-", 1); sb_put(&b, ins_seq[i], strlen(ins_seq[i])); sb_put(&b, "|", 1); sb_printf_ll(&b, right_end);
        sb_put(&b, "|ins|fwd\n", 9);
        sb_ref(&b, ref, left_start, left_end, 0); sb_put(&b, ins_seq[i], strlen(ins_seq[i])); sb_ref(&b, ref, right_start, right_end, 0);
        sb_put(&b, "\n", 1);
    }
    for (int64_t i = 0; i < n_fus; ++i) {                                     /* print_fusion :166-233 */
        int64_t llen, rlen; const char* lref = contig(g, fus[i].ref_id1, &llen); const char* rref = contig(g, fus[i].ref_id2, &rlen);
        if (!lref || !rref) continue;
        int half = read_len - min_anchor_len;
        int64_t fl = fus[i].left, fr = fus[i].right; uint32_t dir = fus[i].dir;
        if (!(fl < llen && fr < rlen)) continue;
        int64_t left_start, left_end, right_start, right_end;
        if (dir == ORC_FUSION_FF || dir == ORC_FUSION_FR) { left_start = fl + 1 >= half ? fl - half + 1 : 0; left_end = left_start + half; }
        else { left_start = fl; left_end = left_start + half < llen ? left_start + half : llen; }
        if (dir == ORC_FUSION_FF || dir == ORC_FUSION_RF) { right_start = fr; right_end = right_start + half < rlen ? right_start + half : rlen; }
        else { right_end = fr + 1; right_start = right_end >= half ? right_end - half : 0; }
        if (!(left_start < left_end && left_end <= llen && right_start < right_end && right_end <= rlen)) continue;
        int lrc = dir == ORC_FUSION_RF || dir == ORC_FUSION_RR, rrc = dir == ORC_FUSION_FR || dir == ORC_FUSION_RR;
        int64_t ls_print = lrc ? left_end - 1 : left_start, re_print = rrc ? right_start - 1 : right_end;
        const char* d = dir == ORC_FUSION_FR ? "fr" : dir == ORC_FUSION_RF ? "rf" : dir == ORC_FUSION_RR ? "rr" : "ff";
        const char* ln = names[fus[i].ref_id1 - 1]; const char* rn = names[fus[i].ref_id2 - 1];
        sb_put(&b, ">", 1); sb_put(&b, ln, strlen(ln)); sb_put(&b, "-", 1); sb_put(&b, rn, strlen(rn)); sb_put(&b, "|", 1);
        sb_printf_ll(&b, ls_print); sb_put(&b, "|", 1); sb_printf_ll(&b, fl); sb_put(&b, "-", 1); sb_printf_ll(&b, fr); sb_put(&b, "|", 1);
        sb_printf_ull(&b, (unsigned long long)re_print); sb_put(&b, "|fus|", 5);     /* size_t in the reference: 0 - 1 wraps (:214) */ sb_put(&b, d, 2); sb_put(&b, "\n", 1);
        sb_ref(&b, lref, left_start, left_end, lrc); sb_ref(&b, rref, right_start, right_end, rrc); sb_put(&b, "\n", 1);
    }
    return b.p;
}
