// long_spanning_reads --fusion-search: the fusion branches of dfs_seg_hits / merge_segment_chain / merge_chain
// (long_spanning_reads.cpp:2222-2610, :2101-2220, :805-2038), BowtieHit::reverse (bwt_map.h:331-442) and the
// two-contig forms of check_editdist_consistency / bowtie_sam_extra (bwt_map.cpp:2349-2648).
//
// One thread per read, general arrays: this is the tier the stitch kernels hand a read to when fusion search is on and
// the read is not a plain run of abutting single hits (those come out the same with or without fusion search and stay in
// tier 0).  What differs from the other tiers' data model:
//   * a hit has two contigs (ref_id, ref_id2) and its cigar may run down the genome (lower-case ops) or jump (a fusion op
//     whose length is the position on the second contig);
//   * hits are reversed one by one (BowtieHit::reverse), so "the chain's sequence" is no longer the read or its reverse
//     complement: a hit's sequence is a list of whole read segments, each forward or reverse-complemented -- 4 bits per
//     segment in one 64-bit word (FHit::sq), read through f_seq_code().
// Genome bases are read one at a time with the contig bounds checked (outside = N, as seqan's infix gives the oracle).
#pragma once
#include "thj_span_core.h"

namespace thj {
static constexpr int FUS_MAXSEG = 8;          // segments of a read the fusion branches take (thj_span_run_async refuses --fusion-search beyond)

enum { OP_FUS_FF = 7, OP_FUS_FR = 8, OP_FUS_RF = 9, OP_FUS_RR = 10 };
enum { SH_FLIPPED = 8 };                      // == THJ_HIT_STRAND_FLIPPED (SH_FUSED = 16 == THJ_HIT_FUSED: thj_span_core.h)
static constexpr int FUS_MAXC = 16;           // cigar ops of a (joined) hit; cigar[15] of an output record carries ref_id2
static constexpr int FUS_MAXJOIN = 24;        // joined alignments kept per read before sort + unique

struct FusKey { uint32_t ref1, ref2, left, right, dir; };      // == thj_span_fusion; Fusion::operator< order (fusions.h:44-71)
struct FusionSet { const FusKey* keys; int64_t n; };

struct FHit {
    uint32_t ref_id, ref_id2;
    int32_t left;
    int32_t n;                 // 0 = BowtieHit()
    uint32_t c[FUS_MAXC];
    uint8_t anti, asplice, mm, ed;
    uint8_t end, nsq, pad0, pad1;
    u64 sq;                    // nsq pieces, 4 bits each, first piece in the low nibble: segment index | rc << 3
};

struct FRead {                 // the read a thread works on
    const u64* rp; int W, rl, L, nsegs;
    const uint8_t* qual;
};

THJ_HD bool f_is_fusion_op(int op) { return op >= OP_FUS_FF && op <= OP_FUS_RR; }
THJ_HD int f_comp(int c) { return c >= 4 ? 4 : 3 - c; }
THJ_HD int f_read_code(const FRead& r, int j) {           // base j of the read: 0..3, 4 = N, 5 = out of range
    if (j < 0 || j >= r.rl) return 5;
    const int w = j >> 6, b = j & 63;
    if ((r.rp[2 * r.W + w] >> b) & 1ull) return 4;
    return (int)(((r.rp[w] >> b) & 1ull) | (((r.rp[r.W + w] >> b) & 1ull) << 1));
}
THJ_HD int f_piece_len(const FRead& r, int seg) { return seg == r.nsegs - 1 ? r.rl - seg * r.L : r.L; }
THJ_HD int f_seq_len(const FRead& r, const FHit& h) {
    int l = 0;
    for (int k = 0; k < h.nsq; ++k) l += f_piece_len(r, (int)((h.sq >> (4 * k)) & 7));
    return l;
}
// base i of the hit's sequence; 5 when i is outside it
THJ_HD int f_seq_code(const FRead& r, const FHit& h, int i) {
    if (i < 0) return 5;
    for (int k = 0; k < h.nsq; ++k) {
        const int e = (int)((h.sq >> (4 * k)) & 15), seg = e & 7, pl = f_piece_len(r, seg);
        if (i < pl) {
            if (e & 8) return f_comp(f_read_code(r, seg * r.L + pl - 1 - i));
            return f_read_code(r, seg * r.L + i);
        }
        i -= pl;
    }
    return 5;
}
// position in the read (for the quality string) of base i of the hit's sequence, -1 outside
THJ_HD void f_seq_reverse(FHit& h) {
    u64 o = 0;
    for (int k = 0; k < h.nsq; ++k) o |= (((h.sq >> (4 * k)) & 15) ^ 8ull) << (4 * (h.nsq - 1 - k));
    h.sq = o;
}
THJ_HD bool f_seq_is_read(const FRead& r, const FHit& h) {          // new_hit.seq() == read_seq
    if (f_seq_len(r, h) != r.rl) return false;
    bool ident = h.nsq == r.nsegs;
    for (int k = 0; ident && k < h.nsq; ++k) ident = ((h.sq >> (4 * k)) & 15) == (u64)k;
    if (ident) return true;
    for (int i = 0; i < r.rl; ++i) if (f_seq_code(r, h, i) != f_read_code(r, i)) return false;
    return true;
}
THJ_HD int g_code(const Genome& g, uint32_t ref_id, int64_t pos) {   // Dna5 of the contig at pos, N outside
    if (pos < 0 || pos >= (int64_t)g_len(g, ref_id)) return 4;
    return plane_code(g_fetch(g, ref_id, pos), 0);
}
THJ_HD int g_code_rc(const Genome& g, uint32_t ref_id, int64_t pos) { return f_comp(g_code(g, ref_id, pos)); }
// The walks over a whole alignment (check_editdist_consistency, bowtie_sam_extra) take the hit's sequence and the genome in
// chunks of planes instead of base by base (three plane loads and a walk over the piece list per base: 60 % of
// thj_k_stitch_fusion).
// Up to `want` (<= 64) bases of the hit's sequence from base i on, not past the end of the piece that holds base i; outside
// the sequence the bases are N (f_seq_code's 5, which the callers clamp to 4).  Returns the chunk's length (>= 1).
THJ_HD int f_seq_chunk(const FRead& r, const FHit& h, int i, int want, Planes& s) {
    if (want > 64) want = 64;
    if (i >= 0) {
        for (int k = 0; k < h.nsq; ++k) {
            const int e = (int)((h.sq >> (4 * k)) & 15), seg = e & 7, pl = f_piece_len(r, seg);
            if (i < pl) {
                const int l = want < pl - i ? want : pl - i;
                if (e & 8) s = rc_piece(r_fetch(r.rp, r.W, seg * r.L + pl - i - l, l), l);
                else s = r_fetch(r.rp, r.W, seg * r.L + i, l);
                return l;
            }
            i -= pl;
        }
    }
    s.lo = s.hi = 0; s.nm = lowmask(want);
    return want;
}
// l (1..64) bases of contig `ref` (clen long): contig[pos + t], or with rc the complement of contig[pos - t]; N outside
THJ_HD Planes f_gen_chunk(const Genome& g, uint32_t ref, int32_t clen, int64_t pos, int l, bool rc) {
    const int64_t lo = rc ? pos - l + 1 : pos;                 // the window [lo, lo + l)
    Planes w; w.lo = w.hi = 0; w.nm = lowmask(l);
    if (lo < (int64_t)clen && lo + l > 0) {
        const int64_t start = lo < 0 ? 0 : lo;
        const int sh = (int)(start - lo);
        const int64_t stop = lo + l < (int64_t)clen ? lo + l : (int64_t)clen;
        const u64 valid = lowmask((int)(stop - start)) << sh;
        const Planes f = g_fetch(g, ref, start);
        w.lo = (f.lo << sh) & valid; w.hi = (f.hi << sh) & valid; w.nm = ((f.nm << sh) & valid) | (lowmask(l) & ~valid);
    }
    return rc ? rc_piece(w, l) : w;
}
THJ_HD u64 f_mism(const Planes& a, const Planes& b, int l) {   // Dna5 codes differ (N equals N)
    return ((((a.lo ^ b.lo) | (a.hi ^ b.hi)) & ~(a.nm | b.nm)) | (a.nm ^ b.nm)) & lowmask(l);
}

THJ_HD int f_right(const FHit& h) {                                  // bwt_map.h:213-243
    int r = h.left;
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]), len = (int)cig_len(h.c[i]);
        if (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL) r += len;
        else if (op == OP_mATCH || op == OP_rEF_SKIP || op == OP_dEL) r -= len;
        else if (f_is_fusion_op(op)) r = len;
    }
    return r;
}
THJ_HD int f_read_len(const FHit& h) {                               // bwt_map.h:141-163
    int l = 0;
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]);
        if (op == OP_MATCH || op == OP_mATCH || op == OP_INS || op == OP_iNS || op == OP_SOFT_CLIP) l += (int)cig_len(h.c[i]);
    }
    return l;
}
THJ_HD bool f_spliced(const FHit& h) {
    for (int i = 0; i < h.n; ++i) { const int op = cig_op(h.c[i]); if (op == OP_REF_SKIP || op == OP_rEF_SKIP) return true; }
    return false;
}
THJ_HD int f_fusion_opcode(const FHit& h) {
    for (int i = 0; i < h.n; ++i) if (f_is_fusion_op(cig_op(h.c[i]))) return cig_op(h.c[i]);
    return 0;
}
THJ_HD bool f_fwd_op(int op) { return op == OP_MATCH || op == OP_REF_SKIP || op == OP_INS || op == OP_DEL; }
THJ_HD bool f_rev_op(int op) { return op == OP_mATCH || op == OP_rEF_SKIP || op == OP_iNS || op == OP_dEL; }
THJ_HD bool f_forwarding_left(const FHit& h) {                       // bwt_map.h:271-289
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]);
        if (f_fwd_op(op)) return true;
        if (f_rev_op(op)) return false;
        if (f_is_fusion_op(op)) break;
    }
    return true;
}
THJ_HD bool f_forwarding_right(const FHit& h) {                      // bwt_map.h:295-313
    for (int i = h.n - 1; i >= 0; --i) {
        const int op = cig_op(h.c[i]);
        if (f_fwd_op(op)) return true;
        if (f_rev_op(op)) return false;
        if (f_is_fusion_op(op)) break;
    }
    return true;
}
THJ_HD bool f_anti2(const FHit& h) {                                 // bwt_map.h:319-329
    const int f = f_fusion_opcode(h);
    if (f == 0 || f == OP_FUS_FF || f == OP_FUS_RR) return h.anti != 0;
    return h.anti == 0;
}
THJ_HD int f_flip_case(int op) {
    switch (op) {
    case OP_MATCH: return OP_mATCH; case OP_mATCH: return OP_MATCH;
    case OP_INS: return OP_iNS; case OP_iNS: return OP_INS;
    case OP_DEL: return OP_dEL; case OP_dEL: return OP_DEL;
    case OP_REF_SKIP: return OP_rEF_SKIP; case OP_rEF_SKIP: return OP_REF_SKIP;
    default: return op;
    }
}
// BowtieHit::reverse, bwt_map.h:331-442 (in place)
THJ_HD void f_reverse(FHit& h) {
    uint32_t right = (uint32_t)h.left, fusion_pos = (uint32_t)h.left;
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]); const uint32_t len = cig_len(h.c[i]);
        if (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL) right += len;
        else if (op == OP_mATCH || op == OP_rEF_SKIP || op == OP_dEL) right -= len;
        else if (f_is_fusion_op(op)) { fusion_pos = right; right = len; }
    }
    const bool fl = f_forwarding_left(h);
    if (fl) fusion_pos -= 1; else fusion_pos += 1;
    const int f = f_fusion_opcode(h);
    int32_t nleft;
    if (f == 0 || f == OP_FUS_FF || f == OP_FUS_RR) nleft = fl ? (int32_t)(right - 1) : (int32_t)(right + 1);
    else nleft = f == OP_FUS_FR ? (int32_t)(right + 1) : (int32_t)(right - 1);
    for (int i = 0, k = h.n - 1; i <= k; ++i, --k) {
        const uint32_t a = h.c[i], b = h.c[k];
        const int oa = cig_op(a), ob = cig_op(b);
        h.c[i] = f_is_fusion_op(ob) ? cig(ob, fusion_pos) : cig(f_flip_case(ob), cig_len(b));
        if (k != i) h.c[k] = f_is_fusion_op(oa) ? cig(oa, fusion_pos) : cig(f_flip_case(oa), cig_len(a));
    }
    const uint32_t t = h.ref_id; h.ref_id = h.ref_id2; h.ref_id2 = t;
    h.left = nleft;
    if (f == OP_FUS_FR || f == OP_FUS_RF) h.anti = h.anti ? 0 : 1;
    f_seq_reverse(h);
}
THJ_HD int f_gap_length(const uint32_t* c, int n) {                  // bwt_map.cpp:32-43
    int e = 0;
    for (int i = 0; i < n; ++i) { const int op = cig_op(c[i]); if (op == OP_INS || op == OP_iNS || op == OP_DEL || op == OP_dEL) e += (int)cig_len(c[i]); }
    return e;
}
// fusions_from_spliced_hit(bh, fusions, auto_sort = false)[0] (fusions.cpp:441-495)
THJ_HD bool f_first_fusion(const FHit& h, uint32_t& fl, uint32_t& fr) {
    uint32_t pos = (uint32_t)h.left;
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]); const uint32_t len = cig_len(h.c[i]);
        if (op == OP_REF_SKIP || op == OP_MATCH || op == OP_DEL) pos += len;
        else if (op == OP_rEF_SKIP || op == OP_mATCH || op == OP_dEL) pos -= len;
        else if (f_is_fusion_op(op)) { pos = (op == OP_FUS_RF || op == OP_FUS_RR) ? pos + 1 : pos - 1; fl = pos; fr = len; return true; }
    }
    return false;
}
THJ_HD void f_reverse_if_needed(FHit& h) {                           // :1985-1999, :2201-2216
    bool rev = h.ref_id > h.ref_id2;
    if (h.ref_id == h.ref_id2) { uint32_t fl, fr; if (f_first_fusion(h, fl, fr)) rev = fl > fr; }
    if (rev) f_reverse(h);
}

// check_editdist_consistency, bwt_map.cpp:2349-2465
THJ_HD bool f_check_editdist(const Genome& g, const FRead& rd, const FHit& h) {
    if (g_len(g, h.ref_id) == 0 || g_len(g, h.ref_id2) == 0) return false;
    uint32_t ref = h.ref_id;
    int32_t clen = g_len(g, ref);
    int pos_seq = 0, mismatch = 0, n_mism = 0;
    int64_t pos_ref = h.left;
    bool saw = false;
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]); const int len = (int)cig_len(h.c[i]);
        if (op == OP_MATCH || op == OP_mATCH) {
            for (int o = 0; o < len;) {
                Planes s;
                const int l = f_seq_chunk(rd, h, pos_seq + o, len - o, s);
                const Planes r = f_gen_chunk(g, ref, clen, op == OP_MATCH ? pos_ref + o : pos_ref - o, l, op != OP_MATCH);
                mismatch += popc(f_mism(s, r, l));
                n_mism += popc(s.nm & r.nm & lowmask(l));
                o += l;
            }
            pos_seq += len;
            pos_ref += op == OP_MATCH ? len : -len;
        } else if (op == OP_INS || op == OP_iNS) pos_seq += len;
        else if (op == OP_DEL || op == OP_REF_SKIP) pos_ref += len;
        else if (op == OP_dEL || op == OP_rEF_SKIP) pos_ref -= len;
        else if (f_is_fusion_op(op)) {
            if (saw) return false;
            ref = h.ref_id2; clen = g_len(g, ref); pos_ref = len; saw = true;
        }
    }
    return mismatch == (int)h.mm || mismatch + n_mism == (int)h.mm;
}

// Fusion::operator< bounds over the sorted fusion list (std::set::upper_bound / lower_bound, :1631-1632)
THJ_HD int f_fus_cmp(const FusKey& a, const FusKey& b) {
    if (a.ref1 != b.ref1) return a.ref1 < b.ref1 ? -1 : 1;
    if (a.ref2 != b.ref2) return a.ref2 < b.ref2 ? -1 : 1;
    if (a.left != b.left) return a.left < b.left ? -1 : 1;
    if (a.right != b.right) return a.right < b.right ? -1 : 1;
    if (a.dir != b.dir) return a.dir < b.dir ? -1 : 1;
    return 0;
}
THJ_HD int64_t f_fus_upper(const FusionSet& F, const FusKey& k) {
    int64_t lo = 0, hi = F.n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (f_fus_cmp(k, F.keys[m]) < 0) hi = m; else lo = m + 1; }
    return lo;
}
THJ_HD int64_t f_fus_lower(const FusionSet& F, const FusKey& k) {
    int64_t lo = 0, hi = F.n;
    while (lo < hi) { const int64_t m = (lo + hi) >> 1; if (f_fus_cmp(F.keys[m], k) < 0) lo = m + 1; else hi = m; }
    return lo;
}

// the cigar of a closed pair: prev's ops with the last one set to `back_len` (dropped when <= 0), the closing op, curr's ops
// with the first one set to `front_len` (dropped when <= 0)
THJ_HD int f_splice(uint32_t* out, const FHit& prev, int back_len, bool back_u32_zero_drop, uint32_t mid, const FHit& curr, int64_t front_len) {
    int n = 0;
    for (int q = 0; q < prev.n; ++q) out[n++] = prev.c[q];
    if (back_u32_zero_drop ? ((uint32_t)back_len & 0x0FFFFFFFu) == 0 : back_len <= 0) --n;
    else out[n - 1] = cig(cig_op(out[n - 1]), (uint32_t)back_len);
    out[n++] = mid;
    for (int q = (front_len > 0 ? 0 : 1); q < curr.n; ++q) out[n++] = q == 0 ? cig(cig_op(curr.c[0]), (uint32_t)front_len) : curr.c[q];
    return n;
}

// merge_chain, long_spanning_reads.cpp:805-2038.  chain[0..n) in chain order; false = BowtieHit().
THJ_HD bool f_merge_chain(const Genome& g, const Params& p, const SpanSets& S, const FusionSet& F, const FRead& rd, FHit* chain, int n,
                          int fusion_dir, FHit& out) {
    const int L = p.segment_length;
    int antisense = chain[0].anti;
    const int left = chain[0].left;
    u64 seq_sq = 0; int seq_nsq = 0;
    int old_read_length = 0;
    for (int i = 0; i < n; ++i) {                                             // :826-831
        if (seq_nsq + chain[i].nsq > 16) return false;
        seq_sq |= chain[i].sq << (4 * seq_nsq); seq_nsq += chain[i].nsq;
        old_read_length += f_read_len(chain[i]);
    }
    {                                                                         // :843-897
        int num_fusions = f_fusion_opcode(chain[0]) == 0 ? 0 : 1;
        bool passed = false;
        for (int k = 1; k < n; ++k) {
            const FHit& prev = chain[k - 1]; const FHit& curr = chain[k];
            if (prev.ref_id != prev.ref_id2 || prev.ref_id2 != curr.ref_id) passed = true;
            if (prev.ref_id2 != curr.ref_id) ++num_fusions;
            if (f_fusion_opcode(curr) != 0) ++num_fusions;
            if (prev.ref_id2 == curr.ref_id) {
                const bool reversed = (fusion_dir == OP_FUS_FR && passed) || (fusion_dir == OP_FUS_RF && !passed);
                const int gap = reversed ? f_right(prev) - curr.left : curr.left - f_right(prev);
                const int maxi = p.max_report_intron < p.fusion_min_dist ? p.max_report_intron : p.fusion_min_dist;
                if (gap < -p.max_insertion_length || (gap > p.max_deletion_length && (gap < p.min_report_intron || gap > maxi))) {
                    passed = true; ++num_fusions;
                }
            }
            if (num_fusions >= 2) return false;
        }
    }
    if (THJ_EXPF(1 << 19)) return false;
    int pi = 0, ci = 1, curr_seg_index = 1;
    bool fusion_passed = false;
    // chain[0..pi] holds the hits done, chain[ti..] those to come (ci = pi + 1 counts them as if they followed at once: a closed pair
    // leaves a hole instead of moving every later hit down one place -- up to four 100-byte copies through scratch per closure)
    int ti = 1;
    while (ci < n) {
        FHit& prev = chain[pi]; FHit& curr = chain[ti];
        antisense = prev.anti;
        if (f_fusion_opcode(prev) != 0 || prev.ref_id2 != curr.ref_id) fusion_passed = true;
        if (!(op_is_match(cig_op(prev.c[prev.n - 1])) || op_is_match(cig_op(curr.c[0])))) return false;
        if (f_spliced(prev) && f_spliced(curr) && prev.asplice != curr.asplice) return false;
        bool found = false;
        int antisense_closure = f_spliced(prev) ? prev.asplice : curr.asplice;
        uint32_t nc[2 * FUS_MAXC + 2]; int nn = 0;
        int mismatch = 0;
        const int prev_end = (int)cig_len(prev.c[prev.n - 1]);
        const int curr_front = (int)cig_len(curr.c[0]);
        bool check_fusion = prev.ref_id2 != curr.ref_id;
        const int prev_right = f_right(prev);
        if (prev.ref_id2 == curr.ref_id) {
            const bool reversed = (fusion_dir == OP_FUS_FR && fusion_passed) || (fusion_dir == OP_FUS_RF && !fusion_passed);
            const uint32_t ref = prev.ref_id2;
            const bool have_ref = g_len(g, ref) != 0;
            int lbnd, rbnd;
            if (reversed) { lbnd = curr.left - 4; rbnd = prev_right + 4; } else { lbnd = prev_right - 4; rbnd = curr.left + 4; }
            const int dist = reversed ? prev_right - curr.left : curr.left - prev_right;
            const bool same_strand = f_anti2(prev) == (curr.anti != 0);
            if (dist < 0 && dist >= -p.max_insertion_length && same_strand) {
                // ---- insertion closure :1010-1306
                if (!have_ref) return false;
                int64_t lb = upper_bound_u64(S.ins_keys, S.n_ins, ins_key(g, ref, (uint32_t)lbnd, 0));
                const int64_t ub = upper_bound_u64(S.ins_keys, S.n_ins, ins_key(g, ref, (uint32_t)rbnd, p.max_insertion_length));
                const u64 cbase = (u64)g.contig_blk[ref - 1] * 64ull;
                for (; lb < ub; ++lb) {
                    const u64 k = S.ins_keys[lb];
                    const int ilen = (int)(k & 15);
                    const int ileft = (int)((int64_t)(k >> 4) - 1 - (int64_t)cbase);
                    if (ilen != (reversed ? curr.left - prev_right : prev_right - curr.left)) continue;
                    int itpr, clti;
                    if (reversed) { itpr = ileft - prev_right; clti = curr.left - ileft; }
                    else { itpr = prev_right - ileft - 1; clti = ileft - curr.left + 1; }
                    if (itpr > prev_end || clti > curr_front) continue;
                    const uint32_t iseq = S.ins_seq[lb];
                    int trm = 0, ins_mm = 0;
                    const int prev_sl = f_seq_len(rd, prev);
                    for (int ri = 0; ri < itpr; ++ri) {
                        int r, o, r2;
                        if (reversed) {
                            r = g_code_rc(g, ref, (int64_t)ileft - ri);
                            o = f_read_code(rd, curr_seg_index * L - itpr + ri);
                            r2 = g_code_rc(g, ref, (int64_t)ileft - (ri - ilen));
                        } else {
                            r = g_code(g, ref, (int64_t)ileft + 1 + ri);
                            o = f_seq_code(rd, prev, prev_sl - itpr + ri);
                            r2 = g_code(g, ref, (int64_t)ileft + 1 + ri - ilen);
                        }
                        if (o > 4) o = 4;
                        if (r == 4 || r != o) ++trm;
                        if (ri < ilen) {
                            int ic = reversed ? f_comp((int)((iseq >> (3 * (ilen - 1 - ri))) & 7u)) : (int)((iseq >> (3 * ri)) & 7u);
                            if (ic == 4 || ic != o) { ++ins_mm; break; }
                        } else if (r2 == 4 || r2 != o) --trm;
                    }
                    for (int ri = 0; ri < clti; ++ri) {
                        const int sp = clti - ri - 1, ip = ilen - ri - 1;
                        int r, o, r2;
                        if (reversed) {
                            r = g_code_rc(g, ref, (int64_t)curr.left - sp);
                            o = f_read_code(rd, curr_seg_index * L + sp);
                            r2 = g_code_rc(g, ref, (int64_t)curr.left - (sp + ilen));
                        } else {
                            r = g_code(g, ref, (int64_t)curr.left + sp);
                            o = f_seq_code(rd, curr, sp);
                            r2 = g_code(g, ref, (int64_t)curr.left + sp + ilen);
                        }
                        if (o > 4) o = 4;
                        if (r == 4 || r != o) ++trm;
                        if (ri < ilen) {
                            int ic = reversed ? f_comp((int)((iseq >> (3 * (ilen - 1 - ip))) & 7u)) : (int)((iseq >> (3 * ip)) & 7u);
                            if (ic == 4 || ic != o) { ++ins_mm; break; }
                        } else if (r2 == 4 || r2 != o) --trm;
                    }
                    if (found) return false;                                           // :1243-1247
                    if (ins_mm == 0) {
                        mismatch = -trm;
                        found = true;
                        nn = f_splice(nc, prev, prev_end - itpr, true, cig(reversed ? OP_iNS : OP_INS, (uint32_t)ilen), curr,
                                      (int64_t)((cig_len(curr.c[0]) + (uint32_t)(itpr - ilen)) & 0x0FFFFFFFu));
                    }
                }
                if (!found) return false;
            } else if (dist > 0 && dist <= p.max_report_intron && same_strand) {
                // ---- junction / deletion closure :1311-1591
                if (!have_ref) return false;
                int64_t lb, ub;
                junc_range(S, junc_key(g, ref, (uint32_t)lbnd, (uint32_t)(rbnd - 8), true), junc_key(g, ref, (uint32_t)(lbnd + 8), (uint32_t)rbnd, false), lb, ub);
                const u64 cbase = (u64)g.contig_blk[ref - 1] * 64ull;
                int best = 0xff;
                const int prev_sl = f_seq_len(rd, prev);
                for (; lb < ub; ++lb) {
                    const u64 k = S.junc_keys[lb];
                    const int jl = (int)((int64_t)(k >> 30) - 1 - (int64_t)cbase);
                    const int jr = jl + (int)((k >> 1) & ((1ull << 29) - 1));
                    int dtl, dtr;
                    if (reversed) { dtl = jl - curr.left; dtr = jr - prev_right - 1; } else { dtl = jl - prev_right + 1; dtr = jr - curr.left; }
                    if (!(dtl >= -4 && dtl <= 4 && dtr >= -4 && dtr <= 4 && dtl == dtr)) continue;
                    if ((reversed && (dtl > prev_end || -dtl > curr_front)) || (!reversed && (dtl > curr_front || -dtl > prev_end))) continue;
                    int new_mm = 0, old_mm = 0;
                    if (dtl > 0) {
                        for (int i = 0; i < dtl; ++i) {
                            int s, a, b;
                            if (reversed) {
                                s = f_read_code(rd, curr_seg_index * L - dtl + i);
                                a = g_code_rc(g, ref, (int64_t)jl - i); b = g_code_rc(g, ref, (int64_t)jr - 1 - i);
                            } else {
                                s = f_seq_code(rd, curr, i);
                                a = g_code(g, ref, (int64_t)prev_right + i); b = g_code(g, ref, (int64_t)curr.left + i);
                            }
                            if (s != a) ++new_mm;
                            if (s != b) ++old_mm;
                        }
                    } else if (dtl < 0) {
                        const int ad = -dtl;
                        for (int i = 0; i < ad; ++i) {
                            int s, a, b;
                            if (reversed) {
                                int avail = rd.rl - curr_seg_index * L; if (avail > ad) avail = ad; if (avail < 0) avail = 0;
                                s = f_read_code(rd, curr_seg_index * L + avail - (ad - i));
                                a = g_code_rc(g, ref, (int64_t)prev_right - i); b = g_code_rc(g, ref, (int64_t)curr.left - i);
                            } else {
                                s = f_seq_code(rd, prev, prev_sl - (ad - i));
                                a = g_code(g, ref, (int64_t)jr + i); b = g_code(g, ref, (int64_t)jl + 1 + i);
                            }
                            if (s != a) ++new_mm;
                            if (s != b) ++old_mm;
                        }
                    }
                    const int diff = new_mm - old_mm;
                    if (diff >= best || new_mm >= 2) continue;
                    best = diff;
                    const int skip = jr - jl - 1;
                    uint32_t mid;
                    if ((uint32_t)skip <= (uint32_t)p.max_deletion_length) {
                        mid = cig(reversed ? OP_dEL : OP_DEL, (uint32_t)skip);
                        antisense_closure = f_spliced(prev) ? prev.asplice : curr.asplice;
                    } else {
                        mid = cig(reversed ? OP_rEF_SKIP : OP_REF_SKIP, (uint32_t)skip);
                        antisense_closure = (int)(k & 1ull);
                    }
                    nn = f_splice(nc, prev, reversed ? prev_end - dtl : prev_end + dtl, false, mid, curr, reversed ? curr_front + dtr : curr_front - dtr);
                    mismatch = diff;
                    found = true;
                }
                if (!found) return false;
            } else if (!(dist == 0 && same_strand)) check_fusion = true;
        }
        if (check_fusion) {                                                            // :1596-1818
            if (THJ_EXPF(128)) return false;
            uint32_t r1 = prev.ref_id2, r2 = curr.ref_id;
            uint32_t fl = (uint32_t)prev_right - 4u, fr = (uint32_t)curr.left - 4u;
            bool reversed = false;
            if (fusion_dir != OP_FUS_FF && (r2 < r1 || (r1 == r2 && fl > fr))) {
                reversed = true;
                uint32_t t = r1; r1 = r2; r2 = t;
                t = fl; fl = fr; fr = t;
            }
            FusKey k1{r1, r2, fl, fr, (uint32_t)OP_FUS_FF}, k2{r1, r2, fl + 8u, fr + 8u, (uint32_t)OP_FUS_FF};
            int64_t lb = f_fus_upper(F, k1);
            const int64_t ub = f_fus_lower(F, k2);
            const uint32_t ref1 = prev.ref_id2, ref2 = curr.ref_id;
            const bool have = g_len(g, ref1) != 0 && g_len(g, ref2) != 0;
            int best = 0xff;
            const int prev_sl = f_seq_len(rd, prev);
            for (; lb < ub; ++lb) {
                int lb_left = (int)F.keys[lb].left, lb_right = (int)F.keys[lb].right;
                if (reversed) { lb_left = (int)F.keys[lb].right; lb_right = (int)F.keys[lb].left; }
                const int dtl = fusion_dir == OP_FUS_RF ? prev_right - lb_left + 1 : lb_left - prev_right + 1;
                const int dtr = fusion_dir == OP_FUS_FR ? curr.left - lb_right : lb_right - curr.left;
                if (!(dtl >= -4 && dtl <= 4 && dtr >= -4 && dtr <= 4 && dtl == dtr)) continue;
                if (dtl > curr_front || -dtl > prev_end) continue;
                if (!have) return false;
                int new_mm = 0, old_mm = 0;
                const bool own_seq = fusion_dir == OP_FUS_FF || fusion_dir == OP_FUS_RR;
                if (dtl > 0) {
                    for (int i = 0; i < dtl; ++i) {
                        const int a = fusion_dir == OP_FUS_RF ? g_code_rc(g, ref1, (int64_t)prev_right - i) : g_code(g, ref1, (int64_t)prev_right + i);
                        const int b = fusion_dir == OP_FUS_FR ? g_code_rc(g, ref2, (int64_t)curr.left - i) : g_code(g, ref2, (int64_t)curr.left + i);
                        const int s = own_seq ? f_seq_code(rd, curr, i) : (i < L ? f_read_code(rd, curr_seg_index * L + i) : 5);
                        if (s != a) ++new_mm;
                        if (s != b) ++old_mm;
                    }
                } else if (dtl < 0) {
                    const int ad = -dtl;
                    for (int i = 0; i < ad; ++i) {
                        const int a = fusion_dir == OP_FUS_FR ? g_code_rc(g, ref2, (int64_t)lb_right - i) : g_code(g, ref2, (int64_t)lb_right + i);
                        const int b = fusion_dir == OP_FUS_RF ? g_code_rc(g, ref1, (int64_t)lb_left - 1 - i) : g_code(g, ref1, (int64_t)lb_left + 1 + i);
                        int s;
                        if (own_seq) s = f_seq_code(rd, prev, prev_sl - (ad - i));
                        else {
                            const int st = (curr_seg_index - 1) * L;
                            int plen = rd.rl - st; if (plen > L) plen = L; if (plen < 0) plen = 0;
                            s = f_read_code(rd, st + plen - (ad - i));
                        }
                        if (s != a) ++new_mm;
                        if (s != b) ++old_mm;
                    }
                }
                const int diff = new_mm - old_mm;
                if (diff >= best || new_mm >= 2) continue;
                best = diff;
                nn = f_splice(nc, prev, prev_end + dtl, false, cig(fusion_dir, (uint32_t)lb_right), curr, curr_front - dtr);
                antisense_closure = f_spliced(prev) ? prev.asplice : curr.asplice;
                mismatch = diff;
                found = true;
            }
            if (!found) return false;
        }
        if (found) {                                                                   // :1822-1870
            if (nn > FUS_MAXC - 1) return false;        // device capacity (the record keeps its last cigar slot for ref_id2)
            FHit m;
            const int mismatches = (int)prev.mm + (int)curr.mm + mismatch;
            m.ref_id = prev.ref_id; m.ref_id2 = curr.ref_id2; m.left = prev.left;
            m.n = nn;
            for (int q = 0; q < FUS_MAXC; ++q) m.c[q] = q < nn ? nc[q] : 0u;
            m.anti = (uint8_t)antisense; m.asplice = (uint8_t)antisense_closure;
            m.mm = (uint8_t)mismatches; m.ed = (uint8_t)(mismatches + f_gap_length(nc, nn));
            m.end = 0; m.pad0 = m.pad1 = 0;
            if (prev.nsq + curr.nsq > 16) return false;
            m.sq = prev.sq | (curr.sq << (4 * prev.nsq)); m.nsq = (uint8_t)(prev.nsq + curr.nsq);
            chain[pi] = m;
            ++ti;
            --n;
            ci = pi + 1;
            ++curr_seg_index;
            continue;
        }
        if (pi + 1 != ti) chain[pi + 1] = chain[ti];
        ++pi; ++ci; ++ti; ++curr_seg_index;
    }
    // :1888-1944 concatenate
    bool saw_as = false, saw_s = false;
    int num_mm = 0;
    FHit& nh = out;                 // built where the caller wants it (the caller clears it when this returns false)
    nh.n = 0;
    for (int s = 0; s < n; ++s) {
        num_mm += chain[s].mm;
        if (f_spliced(chain[s])) {
            if (chain[s].asplice) { if (saw_s) return false; saw_as = true; }
            else { if (saw_as) return false; saw_s = true; }
        }
        int b0 = 0;
        if (nh.n > 0 && cig_op(nh.c[nh.n - 1]) == cig_op(chain[s].c[0])) {
            nh.c[nh.n - 1] = cig(cig_op(nh.c[nh.n - 1]), cig_len(nh.c[nh.n - 1]) + cig_len(chain[s].c[0]));
            b0 = 1;
        }
        for (int b = b0; b < chain[s].n; ++b) { if (nh.n >= FUS_MAXC - 1) return false; nh.c[nh.n++] = chain[s].c[b]; }
    }
    for (int q = nh.n; q < FUS_MAXC; ++q) nh.c[q] = 0;
    nh.ref_id = chain[0].ref_id; nh.ref_id2 = chain[n - 1].ref_id2; nh.left = left;
    nh.anti = (uint8_t)antisense; nh.asplice = saw_as ? 1 : 0;
    nh.mm = (uint8_t)num_mm; nh.ed = (uint8_t)(num_mm + f_gap_length(nh.c, nh.n));
    nh.end = 0; nh.pad0 = nh.pad1 = 0;
    if (fusion_dir == 0 || fusion_dir == OP_FUS_FF || fusion_dir == OP_FUS_RR) { nh.sq = seq_sq; nh.nsq = (uint8_t)seq_nsq; }   // :1959-1978
    else {                                                                               // :1979-1983: the read itself
        nh.sq = 0; nh.nsq = (uint8_t)rd.nsegs;
        for (int k = 0; k < rd.nsegs; ++k) nh.sq |= (u64)k << (4 * k);
    }
    // the quality string is the read's, reversed when the joined sequence is not the read (bowtie2 mode); a later reverse()
    // turns it once more.  pad0 = 1: reversed.
    nh.pad0 = (fusion_dir == 0 || fusion_dir == OP_FUS_FF || fusion_dir == OP_FUS_RR) ? (f_seq_is_read(rd, nh) ? 0 : 1) : 0;
    {
        bool rev = nh.ref_id > nh.ref_id2;
        if (nh.ref_id == nh.ref_id2) { uint32_t fl, fr; if (f_first_fusion(nh, fl, fr)) rev = fl > fr; }
        if (rev) { f_reverse(nh); nh.pad0 ^= 1; }
    }
    if (fusion_dir != 0) nh.anti = f_seq_is_read(rd, nh) ? 0 : 1;                        // :2007-2013
    if (f_read_len(nh) != old_read_length || (!THJ_EXPF(64) && !f_check_editdist(g, rd, nh))) return false;  // :2022-2034
    return true;
}

THJ_HD bool f_valid_hit(const Params& p, const FHit& h) {                                // :2045-2099
    if (h.n == 0) return false;
    for (int i = 1; i < h.n; ++i) {
        const int cop = cig_op(h.c[i]), pop = cig_op(h.c[i - 1]);
        const uint32_t clen = cig_len(h.c[i]);
        if (!op_is_match(cop) && !op_is_match(pop)) return false;
        if ((cop == OP_INS || cop == OP_iNS) && clen > (uint32_t)p.max_insertion_length) return false;
        if ((cop == OP_DEL || cop == OP_dEL) && clen > (uint32_t)p.max_deletion_length) return false;
        if ((cop == OP_REF_SKIP || cop == OP_rEF_SKIP) && clen < (uint32_t)p.min_report_intron) return false;
    }
    return op_is_match(cig_op(h.c[0])) && op_is_match(cig_op(h.c[h.n - 1]));
}

// merge_segment_chain, :2101-2220.  -> the joined hit in `bh` (n == 0: none)
THJ_HD void f_merge_segment_chain(const Genome& g, const Params& p, const SpanSets& S, const FusionSet& F, const FRead& rd,
                                  const FHit* hits, int n, int fusion_dir, FHit* chain /* [FUS_MAXSEG + 1] */, FHit& bh) {
    bh.n = 0;
    if (n > 1) {
        if (fusion_dir == 0 || fusion_dir == OP_FUS_FF || fusion_dir == OP_FUS_RR) {
            for (int i = 0; i < n; ++i) chain[i] = hits[0].anti ? hits[n - 1 - i] : hits[i];
        } else {
            bool saw = false;
            int m = 0;
            for (int i = 0; i < n; ++i) {
                bool pushed = false;
                if (!saw && i > 0) {
                    if (hits[i - 1].ref_id != hits[i].ref_id) saw = true;
                    else if (hits[i - 1].anti != hits[i].anti) saw = true;
                    else {
                        const int dist = hits[i].anti ? hits[i - 1].left - f_right(hits[i]) : hits[i].left - f_right(hits[i - 1]);
                        if (dist >= p.max_report_intron || dist < -p.max_insertion_length) saw = true;
                    }
                }
                if (f_fusion_opcode(hits[i]) == 0 && ((fusion_dir == OP_FUS_FR && saw) || (fusion_dir == OP_FUS_RF && !saw)) &&
                    hits[i].left < f_right(hits[i])) {
                    if (m > FUS_MAXSEG) return;
                    chain[m] = hits[i]; f_reverse(chain[m]); ++m;
                    pushed = true;
                }
                if (i > 0 && f_fusion_opcode(hits[i]) != 0 && hits[i].ref_id != hits[i - 1].ref_id) {
                    if (m > FUS_MAXSEG) return;
                    chain[m] = hits[i]; f_reverse(chain[m]); ++m;
                    pushed = true;
                }
                if (!saw && f_fusion_opcode(hits[i]) != 0) saw = true;
                if (!pushed) { if (m > FUS_MAXSEG) return; chain[m++] = hits[i]; }
            }
            n = m;
        }
        if (!f_merge_chain(g, p, S, F, rd, chain, n, fusion_dir, bh)) { bh.n = 0; return; }
    } else {
        bh = hits[0];
        bh.pad0 = bh.nsq ? (uint8_t)((bh.sq >> 3) & 1) : 0;     // the record's own QUAL: reversed when its SEQ is the reverse complement
        f_reverse_if_needed(bh);
        if (bh.nsq) bh.pad0 = (uint8_t)((bh.sq >> 3) & 1);
    }
    if (!f_valid_hit(p, bh)) bh.n = 0;
}

THJ_HD FHit fhit_from(const SpanHit& h, int seg, bool last_seg) {
    FHit x;
    x.ref_id = h.ref_id; x.ref_id2 = h.ref_id; x.left = h.left;
    x.n = (int)(h.meta >> 24);
    bool fused = false;
    for (int q = 0; q < FUS_MAXC; ++q) {
        x.c[q] = q < x.n && q < 5 ? h.cigar[q] : 0u;
        if (q < x.n && q < 5 && f_is_fusion_op(cig_op(h.cigar[q]))) fused = true;
    }
    if (fused) x.ref_id2 = h.cigar[4];
    x.anti = (h.meta & SH_ANTI) ? 1 : 0; x.asplice = (h.meta & SH_ASPLICE) ? 1 : 0;
    x.mm = (uint8_t)(h.meta >> 8); x.ed = (uint8_t)(h.meta >> 16);
    x.end = (h.meta & SH_END) ? 1 : 0; x.pad0 = x.pad1 = 0;
    (void)last_seg;
    const bool rec_rev = (x.anti != 0) != ((h.meta & SH_FLIPPED) != 0);
    x.sq = (u64)(seg & 7) | (rec_rev ? 8ull : 0ull); x.nsq = 1;
    return x;
}

THJ_HD bool fhit_less(const FHit& a, const FHit& b) {                  // bwt_map.h:180-207
    if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
    if (a.ref_id2 != b.ref_id2) return a.ref_id2 < b.ref_id2;
    if (a.left != b.left) return a.left < b.left;
    if (a.anti != b.anti) return a.anti < b.anti;
    if (a.mm != b.mm) return a.mm < b.mm;
    if (a.ed != b.ed) return a.ed < b.ed;
    if (a.n != b.n) return a.n < b.n;
    for (int i = 0; i < a.n; ++i)
        if (a.c[i] != b.c[i]) {
            const int oa = cig_op(a.c[i]), ob = cig_op(b.c[i]);
            return oa < ob || (oa == ob && cig_len(a.c[i]) < cig_len(b.c[i]));
        }
    return false;
}
THJ_HD bool fhit_eq(const FHit& a, const FHit& b) {                    // bwt_map.h:167-178
    if (a.ref_id != b.ref_id || a.ref_id2 != b.ref_id2 || a.anti != b.anti || a.left != b.left || a.asplice != b.asplice ||
        a.ed != b.ed || a.n != b.n) return false;
    for (int i = 0; i < a.n; ++i) if (a.c[i] != b.c[i]) return false;
    return true;
}

// bowtie_sam_extra, bwt_map.cpp:2467-2648, on a hit that may run down the genome and change contigs.  h.pad0: the hit's
// quality string is the read's reversed.
THJ_HD void f_sam_extra(const Genome& g, const Params& p, const FRead& rd, const FHit& h, Extras& e) {
    int pos_seq = 0, pos_mm = 0, mismatch = 0, opens = 0, conts = 0, AS = 0;
    int64_t pos_ref = h.left;
    uint32_t ref = h.ref_id;
    bool saw = false;
    md_init(e.md);
    e.AS = e.XM = e.XO = e.XG = e.both_n = 0;
    if (g_len(g, h.ref_id) == 0 || g_len(g, h.ref_id2) == 0) return;
    int32_t clen = g_len(g, ref);
    const int slen = f_seq_len(rd, h);
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]); const int len = (int)cig_len(h.c[i]);
        if (op == OP_MATCH || op == OP_mATCH) {
            for (int o = 0; o < len;) {
                Planes s;
                const int l = f_seq_chunk(rd, h, pos_seq + o, len - o, s);
                const Planes r = f_gen_chunk(g, ref, clen, op == OP_MATCH ? pos_ref + o : pos_ref - o, l, op != OP_MATCH);
                u64 mm = f_mism(s, r, l);
                AS -= p.bowtie2_penalty_for_N * popc(s.nm & r.nm & lowmask(l));     // matching N: still penalised (:2552-2556)
                int last = 0;
                while (mm) {
                    const int b = ctz(mm);
                    mm &= mm - 1;
                    ++mismatch;
                    const int sp = pos_seq + o + b;
                    if (sp < slen) {
                        if (((r.nm | s.nm) >> b) & 1ull) AS -= p.bowtie2_penalty_for_N;
                        else {
                            int q = (int)rd.qual[h.pad0 ? rd.rl - 1 - sp : sp] - 33; if (q > 40) q = 40;
                            AS -= p.bowtie2_min_penalty + ((p.bowtie2_max_penalty - p.bowtie2_min_penalty) * q) / 40;
                        }
                    }
                    pos_mm += b - last;
                    md_put_int_char(e.md, pos_mm, "ACGTN"[plane_code(r, b)]);
                    pos_mm = 0; last = b + 1;
                }
                pos_mm += l - last;
                o += l;
            }
            pos_seq += len;
            pos_ref += op == OP_MATCH ? len : -len;
        } else if (op == OP_INS || op == OP_iNS) {
            pos_seq += len;
            AS -= p.bowtie2_read_gap_open + p.bowtie2_read_gap_cont * len;
            ++opens; conts += len;
        } else if (op == OP_DEL || op == OP_dEL) {
            AS -= p.bowtie2_ref_gap_open + p.bowtie2_ref_gap_cont * len;
            ++opens; conts += len;
            md_put_int_char(e.md, pos_mm, '^');
            const int dl = len < 64 ? len : 64;
            if (dl > 0) {
                const Planes r = f_gen_chunk(g, ref, clen, pos_ref, dl, op != OP_DEL);
                for (int k = 0; k < dl; ++k) md_push(e.md, "ACGTN"[plane_code(r, k)]);
            }
            pos_ref += op == OP_DEL ? len : -len;
            pos_mm = 0;
        } else if (op == OP_REF_SKIP) pos_ref += len;
        else if (op == OP_rEF_SKIP) pos_ref -= len;
        else if (f_is_fusion_op(op)) {
            if (saw) { md_init(e.md); return; }
            ref = h.ref_id2; clen = g_len(g, ref); pos_ref = len; saw = true;
        }
    }
    md_put_int(e.md, pos_mm);
    e.AS = AS; e.XM = mismatch; e.XO = opens; e.XG = conts;
}

template <class Sink>
THJ_HD void f_emit(Sink& sink, uint32_t read_idx, int order, const FHit& h, const Extras& e) {
    uint32_t wds[32];
    wds[0] = read_idx; wds[1] = h.ref_id; wds[2] = (uint32_t)h.left;
    wds[3] = (h.anti ? 1u : 0u) | (h.asplice ? 4u : 0u) | ((uint32_t)h.mm << 8) | ((uint32_t)h.ed << 16) | ((uint32_t)h.n << 24);
    wds[4] = ((uint32_t)e.AS & 0xFFFFu) | ((uint32_t)(e.XM & 0xFF) << 16) | ((uint32_t)(e.XO & 0xFF) << 24);
    wds[5] = (uint32_t)(e.XG & 0xFF) | ((uint32_t)(e.md.len > 40 ? 255 : e.md.len) << 8) | ((uint32_t)(order & 0xFFFF) << 16);
    for (int q = 0; q < SPAN_MAXC; ++q) wds[6 + q] = q < h.n ? h.c[q] : 0u;
    if (f_fusion_opcode(h) != 0) wds[6 + SPAN_MAXC - 1] = h.ref_id2;      // a fusion alignment: ref_id2 rides in the last cigar slot
    for (int q = 0; q < 5; ++q) { wds[22 + 2 * q] = (uint32_t)e.md.w[q]; wds[23 + 2 * q] = (uint32_t)(e.md.w[q] >> 32); }
    sink.emit_words(wds);
}

// One read: JoinSegmentsWorker body (long_spanning_reads.cpp:2767-2831) with fusion search on, in three parts (round 6) so that the search
// from each first-segment hit -- the reference gives each its own 10 000 tries, :2634-2664, and they share nothing but the list they
// append to -- can run on a thread of its own (fusion_read_wave below): fusion_read_nsegs (the worker's early outs), fusion_search_roots (the
// dfs from the first-segment hits [i0_begin, i0_end), appending to joined[nj ..)), fusion_tail (sort, unique, filters, records).
// span_read_fusion is the three in a row.
THJ_HD int fusion_read_nsegs(const Params& p, const SpanHit* hits, const uint32_t* so, int nseg) {      // 0: nothing for this read
    if (so[1] == so[0]) return 0;
    int nsegs = 0;
    while (nsegs < nseg && so[nsegs + 1] > so[nsegs]) ++nsegs;
    if (nsegs > FUS_MAXSEG) nsegs = FUS_MAXSEG;
    if (!(hits[so[nsegs - 1]].meta & SH_END)) return 0;
    if (p.bowtie2)
        for (int s = 0; s < nsegs; ++s)
            if ((int)(so[s + 1] - so[s]) > p.max_seg_multihits) return 0;
    if (THJ_EXPF(1 << 30)) return 0;
    return nsegs;
}
// returns SPAN_OK, or SPAN_TOO_MANY_JOINED when a joined alignment found no room (nothing of the read is emitted then)
// The quick "no" of dfs_seg_hits' pair test (round 6).  Once a chain has its fusion (fusion_dir FF / RR) a further plain hit only joins it
// as a neighbour -- same contig, within [-max_insertion_length, max_report_intron] of the previous hit -- and in a k-copy repeat family k - 1
// of the k candidates of every level are not: 15 800 pair tests per first-segment hit at k = 40 and six segments, a few hundred cigar scans
// through scratch each.  For a previous hit and a candidate that are both plain and run up the genome the test's answer follows from five
// words (:2262-2510 read for that case: the contig test :2296-2304, then the distance of :2338-2352 / :2486-2503, whose failure nothing
// later undoes); everything else takes the whole test.
struct FusPrev { bool ok, anti; uint32_t ref; int32_t left, right; };
THJ_HD FusPrev fus_prev_of(const FHit& h) {
    FusPrev q{true, h.anti != 0, h.ref_id, h.left, h.left};
    if (h.ref_id2 != h.ref_id) q.ok = false;
    for (int i = 0; i < h.n; ++i) {
        const int op = cig_op(h.c[i]);
        if (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL) q.right += (int32_t)cig_len(h.c[i]);
        else if (op != OP_INS) q.ok = false;
    }
    return q;
}
struct FusCand { uint32_t ref; int32_t left, right; uint32_t flags; };      // a candidate's five words: flags 1 = plain and up the genome, 2 = antisense
THJ_HD FusCand fus_cand_of(const SpanHit& h) {
    FusCand c{h.ref_id, h.left, h.left, 1u | ((h.meta & SH_ANTI) ? 2u : 0u)};
    const int n = (int)(h.meta >> 24);
    if (n > 5) c.flags &= ~1u;
    for (int i = 0; i < n && i < 5; ++i) {
        const int op = cig_op(h.cigar[i]);
        if (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL) c.right += (int32_t)cig_len(h.cigar[i]);
        else if (op != OP_INS) c.flags &= ~1u;
    }
    return c;
}
THJ_HD bool fus_quick_reject(const Params& p, const FusPrev& pv, const FusCand& c) {
    if (!(c.flags & 1u)) return false;
    if (c.ref != pv.ref) return true;
    if (((c.flags & 2u) != 0) != pv.anti) return false;
    const int d1 = c.left - pv.right;
    const bool out1 = d1 > p.max_report_intron || d1 < -p.max_insertion_length;
    if (!pv.anti) return out1;
    const int d2 = pv.left - c.right;
    return out1 && (d2 > p.max_report_intron || d2 < -p.max_insertion_length);
}

// Out: where the joined alignments go.  slot(tmp) = where the next one is built, commit(slot) = it is one (false: no room)
struct FusListOut {                     // a thread's own list
    FHit* joined; int cap; int nj;
    THJ_HD FHit* slot(FHit& tmp) { return nj < cap ? &joined[nj] : &tmp; }
    THJ_HD bool commit(FHit*) { if (nj >= cap) return false; ++nj; return true; }
    THJ_HD void count(int) {}
};
// cand: fus_cand_of of the read's hits, [0] = hits[so[0]], where a wave has staged them (a candidate of the quick "no" is then an LDS read,
// not a trip to memory a wave alone on its SIMD waits a microsecond for -- 16 000 of them per first-segment hit at k = 40); null: from hits
template <class Out>
THJ_HD int fusion_search_roots(const Genome& g, const Params& p, const SpanSets& S, const FusionSet& F, const SpanHit* hits, const uint32_t* so,
                               const FRead& rd, int nsegs, uint32_t i0_begin, uint32_t i0_end, Out& out, const FusCand* cand = nullptr) {
    const int fs = p.fusion_search;
    FHit stack[FUS_MAXSEG + 1], saved[FUS_MAXSEG + 1], chain[FUS_MAXSEG + 1];
    uint32_t idx[FUS_MAXSEG + 1];
    int fdir[FUS_MAXSEG + 2];
    bool dirty[FUS_MAXSEG + 2];
    FusPrev pv[FUS_MAXSEG + 2];                   // of stack[d - 1] as level d found it (a candidate's changes to it are undone before the next)
    int status = SPAN_OK;
    for (uint32_t i0 = i0_begin; i0 < i0_end; ++i0) {                       // :2634-2664
        stack[0] = fhit_from(hits[i0], 0, nsegs == 1);
        if (f_fusion_opcode(stack[0]) == OP_FUS_RR) f_reverse(stack[0]);
        int num_try = 10000;
        int d = 1;
        fdir[1] = 0;
        if (nsegs > 1) { idx[1] = so[1]; pv[1] = fus_prev_of(stack[0]); }
        while (d >= 1) {
            if (num_try <= 0) break;
            if (d == nsegs) {                                               // leaf: :2592-2606
                --num_try;
                FHit tmp;
                FHit* bh = out.slot(tmp);                                   // joined where it is kept
                out.count(0);
                if (THJ_EXPF(1 << 29)) bh->n = 0; else
                f_merge_segment_chain(g, p, S, F, rd, stack, nsegs, fdir[d], chain, *bh);
                // (no room: nothing of the read is emitted, whatever the rest of the search finds -- the read is done again with room, or reported)
                if (bh->n && !out.commit(bh)) return SPAN_TOO_MANY_JOINED;
                --d;
                if (d >= 1 && dirty[d]) stack[d - 1] = saved[d];
                continue;
            }
            if (idx[d] >= so[d + 1]) {
                --d;
                if (d >= 1 && dirty[d]) stack[d - 1] = saved[d];
                continue;
            }
            const int fusion_dir = fdir[d];
#ifdef THJ_QR_VERIFY                              // tests/hostsim: the quick "no" only predicts; the whole test runs and must agree
            bool qr_said_no = false;
            if (thj_qr_verify) {
                qr_said_no = fs && (fusion_dir == OP_FUS_FF || fusion_dir == OP_FUS_RR) && pv[d].ok && fus_quick_reject(p, pv[d], fus_cand_of(hits[idx[d]]));
                if (qr_said_no) ++thj_qr_said_no;
            } else
#endif
            if (fs && (fusion_dir == OP_FUS_FF || fusion_dir == OP_FUS_RR) && pv[d].ok) {
                uint32_t i = idx[d];
                const uint32_t end = so[d + 1];
                if (cand) while (i < end && fus_quick_reject(p, pv[d], cand[i - so[0]])) ++i;
                else while (i < end && fus_quick_reject(p, pv[d], fus_cand_of(hits[i]))) ++i;
                if (i != idx[d]) { idx[d] = i; continue; }
            }
            out.count(1);
            // The reference works on copies of the two hits and stores them when the pair is accepted; here the new hit is built in
            // its stack slot (free until it is pushed) and the previous one is worked on where it lies: its first change saves the
            // original (saved[d], dirty[d]), a rejected pair puts it back.  (Five 100-byte copies per step through scratch before.)
            FHit& bh = stack[d];
            bh = fhit_from(hits[idx[d]++], d, d == nsegs - 1);
            FHit& bh_prev = stack[d - 1];
            bool prev_dirty = false, pushed = false;
#define FUS_REV(x) do { if ((x) == &bh_prev && !prev_dirty) { saved[d] = bh_prev; prev_dirty = true; } f_reverse(*(x)); } while (0)
            do {
            FHit* prevHit = &bh_prev;
            FHit* currHit = &bh;
            const bool prev_fused = f_fusion_opcode(*prevHit) != 0, curr_fused = f_fusion_opcode(*currHit) != 0;
            const int num_fusions = (prev_fused ? 1 : 0) + (curr_fused ? 1 : 0);
            int dir = prev_fused ? f_fusion_opcode(*prevHit) : f_fusion_opcode(*currHit);
            if (!fs && num_fusions > 0) continue;
            if (num_fusions >= 2) continue;
            if (fusion_dir != 0 && curr_fused) continue;
            if (fusion_dir == OP_FUS_FF || fusion_dir == OP_FUS_RR) {
                if ((currHit->anti && currHit->ref_id != prevHit->ref_id) || (!currHit->anti && currHit->ref_id != prevHit->ref_id2)) continue;
            }
            if ((fusion_dir == OP_FUS_FR || fusion_dir == OP_FUS_RF) && prevHit->ref_id2 != currHit->ref_id) continue;
            if ((fusion_dir == OP_FUS_FR && !currHit->anti) || (fusion_dir == OP_FUS_RF && currHit->anti)) continue;
            if (curr_fused && dir == OP_FUS_RR) FUS_REV(currHit);
            if (fusion_dir == OP_FUS_FR || fusion_dir == OP_FUS_RF ||
                (curr_fused && currHit->ref_id == currHit->ref_id2 && (dir == OP_FUS_FR || dir == OP_FUS_RF))) {
                if (curr_fused) {
                    if ((dir == OP_FUS_FR && currHit->anti) || (dir == OP_FUS_RF && !currHit->anti)) FUS_REV(currHit);
                } else if (fusion_dir == OP_FUS_FR && currHit->anti) FUS_REV(currHit);
            } else if ((num_fusions == 0 && prevHit->anti && currHit->anti && prevHit->ref_id == currHit->ref_id &&
                        (!fs || (prevHit->left <= f_right(*currHit) + p.max_report_intron &&
                                 prevHit->left + p.max_insertion_length >= f_right(*currHit)))) ||
                       (num_fusions == 1 && (dir == OP_FUS_FF || dir == OP_FUS_RR) &&
                        ((!prev_fused && prevHit->anti) || (!curr_fused && currHit->anti)))) {
                FHit* t = prevHit; prevHit = currHit; currHit = t;
            } else if (num_fusions == 0) {
                if (prevHit->ref_id2 == currHit->ref_id && prevHit->anti == currHit->anti) {
                    const int dist = prevHit->anti ? prevHit->left - f_right(*currHit) : currHit->left - f_right(*prevHit);
                    if (dist > p.max_report_intron || dist < -p.max_insertion_length) {
                        if ((prevHit->anti && prevHit->left > currHit->left) || (!prevHit->anti && prevHit->left < currHit->left)) dir = OP_FUS_FF;
                        else dir = OP_FUS_RR;
                    }
                } else {
                    if (prevHit->anti == currHit->anti) {
                        if ((prevHit->anti && prevHit->ref_id > currHit->ref_id) || (!prevHit->anti && prevHit->ref_id < currHit->ref_id)) dir = OP_FUS_FF;
                        else dir = OP_FUS_RR;
                    } else if (!prevHit->anti) dir = OP_FUS_FR;
                    else dir = OP_FUS_RF;
                    if (dir == OP_FUS_FR) FUS_REV(currHit);
                    else if (dir == OP_FUS_RF) FUS_REV(prevHit);
                }
            }
            if (!fs && dir != 0) continue;
            if (num_fusions == 1 && dir != OP_FUS_FF && dir != OP_FUS_RR) {       // :2442-2514
                bool prev_rep = false, curr_rep = false;
                if (prev_fused) {
                    if ((dir == OP_FUS_FR && !currHit->anti) || (dir == OP_FUS_RF && currHit->anti)) continue;
                    if (prevHit->ref_id2 != currHit->ref_id) prev_rep = true;
                    else if ((dir == OP_FUS_FR && prevHit->anti) || (dir == OP_FUS_RF && !prevHit->anti)) prev_rep = true;
                }
                if (curr_fused) {
                    if ((dir == OP_FUS_FR && prevHit->anti) || (dir == OP_FUS_RF && !prevHit->anti)) continue;
                    if (currHit->ref_id != prevHit->ref_id2) curr_rep = true;
                }
                if (prev_rep) FUS_REV(prevHit);
                if (curr_rep) FUS_REV(currHit);
                prev_rep = curr_rep = false;
                if (f_forwarding_right(*prevHit) != f_forwarding_left(*currHit)) { if (prev_fused) curr_rep = true; else prev_rep = true; }
                if (prev_rep) FUS_REV(prevHit);
                if (curr_rep) FUS_REV(currHit);
            }
            const bool same_contig = prevHit->ref_id2 == currHit->ref_id;
            if (!same_contig && num_fusions > 0) continue;
            if (!fs && (!same_contig || num_fusions > 0)) continue;
            if (same_contig && num_fusions >= 1 && f_anti2(*prevHit) != (currHit->anti != 0)) continue;
            int dist = 0;
            if (same_contig) {
                int bh_l, back_right;
                if ((fusion_dir == OP_FUS_FR || fusion_dir == OP_FUS_RF || dir == OP_FUS_FR || dir == OP_FUS_RF) && f_anti2(*prevHit)) {
                    bh_l = f_right(*prevHit) + 1; back_right = currHit->left + 1;
                } else { bh_l = currHit->left; back_right = f_right(*prevHit); }
                dist = bh_l - back_right;
            }
            if (!same_contig || (same_contig && num_fusions == 0 && dir != 0 && fusion_dir == 0) ||
                (same_contig && dist <= p.max_report_intron && dist >= -p.max_insertion_length &&
                 f_forwarding_right(*prevHit) == f_forwarding_left(*currHit))) {
                dirty[d] = prev_dirty;
                fdir[d + 1] = dir == 0 ? fusion_dir : dir;
#ifdef THJ_QR_VERIFY
                if (qr_said_no) ++thj_qr_wrong;
#endif
                ++d;
                if (d < nsegs) { idx[d] = so[d]; pv[d] = fus_prev_of(stack[d - 1]); }
                pushed = true;
            }
            } while (0);
#undef FUS_REV
            if (!pushed && prev_dirty) stack[d - 1] = saved[d];
        }
    }
    return status;
}
// joined[0 .. nj) in generation order (first-segment hit by first-segment hit) -> the read's records.  big: joined is a buffer of 2 * big_cap
// alignments whose second half is the merge sort's scratch (lists of more than 64)
template <class Sink>
THJ_HD int fusion_tail(const Genome& g, const Params& p, const FRead& rd, FHit* joined, int nj, bool ext, int ext_cap, uint32_t read_idx, Sink& sink) {
    if (!ext || nj <= 64) {
        for (int i = 1; i < nj; ++i) {                    // sort + unique (:2805-2807); stable insertion sort
            FHit t = joined[i]; int k = i;
            while (k > 0 && fhit_less(t, joined[k - 1])) { joined[k] = joined[k - 1]; --k; }
            joined[k] = t;
        }
    } else {                                              // long lists (external buffer): stable merge sort
        FHit* a = joined; FHit* b = joined + ext_cap;
        for (int width = 1; width < nj; width <<= 1) {
            for (int lo = 0; lo < nj; lo += 2 * width) {
                const int mid = lo + width < nj ? lo + width : nj, hi = lo + 2 * width < nj ? lo + 2 * width : nj;
                int i = lo, j = mid, k = lo;
                while (i < mid && j < hi) b[k++] = fhit_less(a[j], a[i]) ? a[j++] : a[i++];
                while (i < mid) b[k++] = a[i++];
                while (j < hi) b[k++] = a[j++];
            }
            FHit* t = a; a = b; b = t;
        }
        if (a != joined) for (int i = 0; i < nj; ++i) joined[i] = a[i];
    }
    int w = 0;
    for (int i = 0; i < nj; ++i) if (w == 0 || !fhit_eq(joined[w - 1], joined[i])) joined[w++] = joined[i];
    nj = w;
    int order = 0;
    for (int i = 0; i < nj; ++i) {
        const FHit& h = joined[i];
        const int gapl = (uint8_t)(h.ed - h.mm);
        if ((int)h.mm > p.read_mismatches || gapl > p.read_gap_length || (int)h.ed > p.read_edit_dist) continue;     // :2810-2813
        if (THJ_EXPF(1 << 28)) continue;
        Extras e;
        f_sam_extra(g, p, rd, h, e);
        f_emit(sink, read_idx, order++, h, e);
    }
    return order;                                         // records emitted
}
template <class Sink>
THJ_HD int span_read_fusion(const Genome& g, const Params& p, const SpanSets& S, const FusionSet& F, const SpanHit* hits, const uint32_t* so,
                            int nseg, const u64* rp, int W, int rl, const uint8_t* qual, uint32_t read_idx, Sink& sink, FHit* ext = nullptr, int ext_cap = 0) {
    const int nsegs = fusion_read_nsegs(p, hits, so, nseg);
    if (nsegs == 0) return SPAN_OK;
    FRead rd{rp, W, rl, p.segment_length, nsegs, qual};
    FHit joined_local[FUS_MAXJOIN];               // ext: see span_read
    FHit* joined = ext ? ext : joined_local;
    const int cap = ext ? ext_cap : FUS_MAXJOIN;
    FusListOut out{joined, cap, 0};
    const int status = fusion_search_roots(g, p, S, F, hits, so, rd, nsegs, so[0], so[1], out);
    if (status == SPAN_TOO_MANY_JOINED) return status;
    fusion_tail(g, p, rd, joined, out.nj, ext != nullptr, ext_cap, read_idx, sink);
    return status;
}

// ---- a read with a long list of joined alignments, by the 64 lanes of a wave (thj_k_stitch_huge under --fusion-search)
// A read of a k-copy repeat family has k first-segment hits, every one the root of a search that joins it with the other
// copies' hits as fusion candidates -- (segments - 1) * (k - 1) + 1 joined alignments a root, 7 800 for k = 40 and six segments --
// up to 10 000 merge_segment_chain calls each.  With one thread on such a read long_spanning_reads --fusion-search took 1 000 s a side on
// configs[3] at full size with 5 % of the pairs from a 41-copy family (round 6; 14 s with this, profiles/HISTORY.md).
// Here: lane l searches roots l, l + 64, ...; what it finds goes to a common list by an atomic counter, tagged (root, number within
// the root); the list is put back into the order the one-thread search makes it in (root by root), sorted through an index
// (merge passes, a lane a merge), and unique / filter / records run a lane an alignment, the records' numbers from a running count.
// X: lane, sync() (workgroup barrier; the workgroup is the wave), atomic_add(), ballot(), mark() (a developer's phase timers) -- FusWaveDev in thj_span.hip, the fibers of
// tests/hostsim.  ws: 2 * cap alignments and 3 * cap words.  sh: the workgroup's (LDS).  The result equals span_read_fusion's with
// ext = ws, whatever the timing of the lanes.
static constexpr int FUS_WAVE_MAXROOT = 1024;
// A read for the wave from the start: with fusion search on, every hit of the second segment is a candidate partner of every hit of the
// first (a fusion when they are not neighbours), so a thread alone would try so[1] x so[2] pairs and what hangs below them -- while the
// other 63 reads of its wave wait (round 6: 94 % of long_spanning_reads --fusion-search on the mix was thj_k_stitch_fusion waiting so; with
// 64 pairs as the limit thj_k_stitch_fusion still took 20 ms a launch, with 9: 1.5 ms).
static constexpr uint32_t FUS_HEAVY_PAIRS = 9;
THJ_HD bool fusion_read_heavy(const uint32_t* so, int nseg) {
    return nseg >= 2 && (uint64_t)(so[1] - so[0]) * (so[2] - so[1]) >= FUS_HEAVY_PAIRS;
}
static constexpr int FUS_WAVE_MAXCAND = 1024;     // hits of a read whose five words are staged (16 KB)
struct FusWaveShared { uint32_t n_app, overflow, base[FUS_WAVE_MAXROOT + 1]; FusCand cand[FUS_WAVE_MAXCAND]; };
THJ_HD constexpr size_t fus_wave_ws_bytes(int cap) { return (size_t)cap * (2 * sizeof(FHit) + 12); }
template <class X>
struct FusWaveOut {
    X* x; FHit* buf; uint32_t* ord; uint32_t* n_app; uint32_t cap, root, seq;
    uint32_t n_count[2];                   // leaves, whole pair tests (the developer's counters, x.mark)
    THJ_HD void count(int k) { ++n_count[k]; }
    THJ_HD FHit* slot(FHit& tmp) { return &tmp; }
    THJ_HD bool commit(FHit* h) {
        const uint32_t k = x->atomic_add(n_app, 1u);
        if (k >= cap) return false;
        buf[k] = *h;
        ord[k] = (root << 16) | seq++;
        return true;
    }
};
template <class X, class Sink>
THJ_HD int fusion_read_wave(X& x, const Genome& g, const Params& p, const SpanSets& S, const FusionSet& F, const SpanHit* hits, const uint32_t* so,
                            int nseg, const u64* rp, int W, int rl, const uint8_t* qual, uint32_t read_idx, Sink& sink, char* ws, int cap,
                            FusWaveShared& sh, int& n_records /* the read's, on every lane */) {
    n_records = 0;
    const int nsegs = fusion_read_nsegs(p, hits, so, nseg);
    if (nsegs == 0) return SPAN_OK;
    FRead rd{rp, W, rl, p.segment_length, nsegs, qual};
    FHit* A = (FHit*)ws;
    FHit* B = A + cap;
    uint32_t* ord = (uint32_t*)(B + cap);
    uint32_t* ia = ord + cap;
    uint32_t* ib = ia + cap;
    const uint32_t n_roots = so[1] - so[0];
    if (n_roots > (uint32_t)FUS_WAVE_MAXROOT || cap > 65536) {                    // (10 000 tries a root: its number within the root fits 16 bits)
        int status = SPAN_OK;
        if (x.lane == 0) {
            FusListOut out{A, cap, 0};
            status = fusion_search_roots(g, p, S, F, hits, so, rd, nsegs, so[0], so[1], out);
            sh.overflow = status == SPAN_TOO_MANY_JOINED;
            sh.n_app = 0;
            if (!sh.overflow) sh.n_app = (uint32_t)fusion_tail(g, p, rd, A, out.nj, true, cap, read_idx, sink);
        }
        x.sync();
        const bool ovf = sh.overflow != 0;
        n_records = (int)sh.n_app;
        x.sync();
        return ovf ? SPAN_TOO_MANY_JOINED : SPAN_OK;
    }
    if (x.lane == 0) { sh.n_app = 0; sh.overflow = 0; }
    const uint32_t n_hits = so[nsegs] - so[0];
    const bool staged = n_hits <= (uint32_t)FUS_WAVE_MAXCAND;
    if (staged) for (uint32_t k = (uint32_t)x.lane; k < n_hits; k += 64) sh.cand[k] = fus_cand_of(hits[so[0] + k]);
    x.sync();
    x.mark(-1, 0);
    for (uint32_t root = (uint32_t)x.lane; root < n_roots; root += 64) {
        FusWaveOut<X> out{&x, B, ord, &sh.n_app, (uint32_t)cap, root, 0u, {0u, 0u}};
        if (fusion_search_roots(g, p, S, F, hits, so, rd, nsegs, so[0] + root, so[0] + root + 1, out, staged ? sh.cand : nullptr) == SPAN_TOO_MANY_JOINED) sh.overflow = 1u;
        sh.base[root] = out.seq;
        x.counts(out.n_count[0], out.n_count[1]);
    }
    x.sync();
    const bool ovf = sh.overflow != 0;
    const int nj = (int)sh.n_app;
    x.sync();
    x.mark(0, nj);
    if (ovf) return SPAN_TOO_MANY_JOINED;
    if (x.lane == 0) {                                                             // counts -> first places
        uint32_t acc = 0;
        for (uint32_t r = 0; r < n_roots; ++r) { const uint32_t c = sh.base[r]; sh.base[r] = acc; acc += c; }
    }
    x.sync();
    for (int k = x.lane; k < nj; k += 64) {
        const uint32_t o = ord[k];
        A[sh.base[o >> 16] + (o & 0xFFFFu)] = B[k];
    }
    for (int k = x.lane; k < nj; k += 64) ia[k] = (uint32_t)k;
    x.sync();
    x.mark(1, 0);
    uint32_t* a = ia; uint32_t* b = ib;
    for (int width = 1; width < nj; width <<= 1) {                                 // stable, as fusion_tail's (sort + unique, :2805-2807)
        const int n_merges = (nj + 2 * width - 1) / (2 * width);
        for (int m = x.lane; m < n_merges; m += 64) {
            const int lo = m * 2 * width;
            const int mid = lo + width < nj ? lo + width : nj, hi = lo + 2 * width < nj ? lo + 2 * width : nj;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) b[k++] = fhit_less(A[a[j]], A[a[i]]) ? a[j++] : a[i++];
            while (i < mid) b[k++] = a[i++];
            while (j < hi) b[k++] = a[j++];
        }
        x.sync();
        uint32_t* t = a; a = b; b = t;
    }
    x.mark(2, 0);
    int order0 = 0;
    for (int i0 = 0; i0 < nj; i0 += 64) {
        const int i = i0 + x.lane;
        bool pass = false;
        if (i < nj) {
            const FHit& h = A[a[i]];
            pass = i == 0 || !fhit_eq(A[a[i - 1]], h);
            const int gapl = (uint8_t)(h.ed - h.mm);
            if ((int)h.mm > p.read_mismatches || gapl > p.read_gap_length || (int)h.ed > p.read_edit_dist) pass = false;     // :2810-2813
            if (THJ_EXPF(1 << 28)) pass = false;
        }
        const unsigned long long m = x.ballot(pass);
        if (pass) {
            const FHit& h = A[a[i]];
            Extras e;
            f_sam_extra(g, p, rd, h, e);
            f_emit(sink, read_idx, order0 + popc((u64)(m & ((1ull << x.lane) - 1ull))), h, e);
        }
        order0 += popc((u64)m);
    }
    n_records = order0;
    x.sync();                                                                      // the workspace is the next read's
    x.mark(3, 0);
    return SPAN_OK;
}

}  // namespace thj
