// thj_span_core.h -- per-read logic of the long_spanning_reads kernel.
//
// Host+device like thj_core.h: compiled by hipcc into thj_k_stitch, and by g++ into
// tests/hostsim for single-stepping on a CPU (test-only).  Re-designed for the GPU:
//   * the chain's sequence is never materialised: it is the read's bit-planes, or their
//     reverse complement, addressed by offset;
//   * the junction / insertion sets are sorted arrays of packed 64-bit keys searched
//     with lower/upper_bound instead of std::set walks (long_spanning_reads.cpp:1311,
//     :1019-1020);
//   * mismatch re-counts, the edit-distance consistency check and the AS/XM/MD pass are
//     64-base plane XOR + popcount.
// Reference: long_spanning_reads.cpp:805-2038 (merge_chain), :2045-2099 (valid_hit),
// :2101-2220, :2222-2610 (dfs_seg_hits), :2612-2667, bwt_map.cpp:2349-2648.
#pragma once
#include "thj_core.h"


namespace thj {

enum { OP_MATCH = 1, OP_mATCH = 2, OP_INS = 3, OP_iNS = 4, OP_DEL = 5, OP_dEL = 6, OP_REF_SKIP = 11, OP_rEF_SKIP = 12,
       OP_SOFT_CLIP = 13 };
THJ_HD uint32_t cig(int op, uint32_t len) { return ((uint32_t)op << 28) | (len & 0x0FFFFFFFu); }
THJ_HD int cig_op(uint32_t c) { return (int)(c >> 28); }
THJ_HD uint32_t cig_len(uint32_t c) { return c & 0x0FFFFFFFu; }
THJ_HD bool op_is_match(int op) { return op == OP_MATCH || op == OP_mATCH; }

struct alignas(16) Q16 { uint32_t x, y, z, w; };     // one 16-byte load / store

struct SpanHit {            // == thj_span_hit, 32 bytes
    uint32_t ref_id;
    int32_t left;
    uint32_t meta;          // flags | mismatches<<8 | edit_dist<<16 | n_cigar<<24
    uint32_t cigar[5];
};
enum { SH_ANTI = 1, SH_END = 2, SH_ASPLICE = 4, SH_FUSED = 16 };

static constexpr int SPAN_MAXC = 16;      // cigar ops of a joined alignment
static constexpr int SPAN_MAXSEG = 16;        // segments of a read (a 2 x 250 bp run at --segment-length 25 has ten)
static constexpr int SPAN_MIDSEG = 8;         // ... of the instances the tiers keep for reads of five to eight (2 x 150 bp: six)
static constexpr int SPAN_MAXJOIN = 96;   // joined alignments kept per read before sort+unique (a read in a 40-copy repeat yields 40)

struct Aln {                // working form of a (partially) joined BowtieHit
    uint32_t ref_id;
    int32_t left;
    int32_t n;
    uint32_t c[SPAN_MAXC];
    uint8_t anti, asplice, mm, ed;
    int32_t rlen;           // read bases covered (sum of the segments' seq lengths)
    int32_t valid;          // 0 = BowtieHit()
};

THJ_HD int aln_right(const Aln& h) {
    int r = h.left;
    for (int i = 0; i < h.n; ++i) {
        int op = cig_op(h.c[i]);
        if (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL) r += (int)cig_len(h.c[i]);
    }
    return r;
}
THJ_HD int aln_read_len(const Aln& h) {
    int l = 0;
    for (int i = 0; i < h.n; ++i) {
        int op = cig_op(h.c[i]);
        if (op == OP_MATCH || op == OP_INS || op == OP_SOFT_CLIP) l += (int)cig_len(h.c[i]);
    }
    return l;
}
THJ_HD bool aln_spliced(const Aln& h) {
    for (int i = 0; i < h.n; ++i) if (cig_op(h.c[i]) == OP_REF_SKIP) return true;
    return false;
}
THJ_HD int cig_gap_len(const uint32_t* c, int n) {
    int g = 0;
    for (int i = 0; i < n; ++i) { int op = cig_op(c[i]); if (op == OP_INS || op == OP_DEL) g += (int)cig_len(c[i]); }
    return g;
}

THJ_HD uint32_t cg(const Aln& a, int i) { return a.c[i]; }
THJ_HD uint32_t cg_fixed(const Aln& a, int i) { return a.c[i]; }

// ---- register-resident alignment of the lean tier -------------------------------------------------------
// Up to LEAN_C cigar ops held in named registers: every access is an unrolled compare/select chain, so nothing
// is indexed dynamically and nothing goes to scratch.  Reads that would need more ops go to the generic tier.
static constexpr int LEAN_C = 8;
struct RCig {
    uint32_t v[LEAN_C];
    THJ_HD uint32_t get(int i) const {
        uint32_t r = 0;
#pragma unroll
        for (int k = 0; k < LEAN_C; ++k) r = (i == k) ? v[k] : r;
        return r;
    }
    THJ_HD void set(int i, uint32_t x) {
#pragma unroll
        for (int k = 0; k < LEAN_C; ++k) v[k] = (i == k) ? x : v[k];
    }
};
struct RAln {
    uint32_t ref_id; int32_t left, n;
    RCig c;
    int anti, asplice, mm, ed, rlen, valid;
};
THJ_HD uint32_t cg(const RAln& a, int i) { return a.c.get(i); }
THJ_HD uint32_t cg_fixed(const RAln& a, int i) { return i < LEAN_C ? a.c.v[i < LEAN_C ? i : 0] : 0u; }
THJ_HD int rc_ref_span(const RCig& c, int n) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < LEAN_C; ++k) {
        int op = cig_op(c.v[k]);
        if (k < n && (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL)) r += (int)cig_len(c.v[k]);
    }
    return r;
}
THJ_HD int rc_read_span(const RCig& c, int n) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < LEAN_C; ++k) {
        int op = cig_op(c.v[k]);
        if (k < n && (op == OP_MATCH || op == OP_INS || op == OP_SOFT_CLIP)) r += (int)cig_len(c.v[k]);
    }
    return r;
}
THJ_HD bool rc_spliced(const RCig& c, int n) {
    bool r = false;
#pragma unroll
    for (int k = 0; k < LEAN_C; ++k) r = r || (k < n && cig_op(c.v[k]) == OP_REF_SKIP);
    return r;
}
THJ_HD int rc_gap_len(const RCig& c, int n) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < LEAN_C; ++k) {
        int op = cig_op(c.v[k]);
        if (k < n && (op == OP_INS || op == OP_DEL)) r += (int)cig_len(c.v[k]);
    }
    return r;
}

struct SpanSets {
    const u64* junc_keys;  int64_t n_juncs;      // junc_key() order == Junction::operator<
    const u64* ins_keys;   const uint32_t* ins_seq; int64_t n_ins;   // ins_key() order; seq 3 bits/base
    // optional coarse index over junc_keys: junc_bucket[b] = first key whose left position (key >> 30) is >=
    // b << JUNC_BUCKET_SHIFT, junc_bucket[n_buckets] = n_juncs.  Null: plain binary searches.
    const uint32_t* junc_bucket; int64_t n_buckets;
};
static constexpr int JUNC_BUCKET_SHIFT = 8;

THJ_HD int64_t lower_bound_u64(const u64* a, int64_t n, u64 k) {   // first i with a[i] >= k
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (a[m] < k) lo = m + 1; else hi = m; }
    return lo;
}
THJ_HD int64_t upper_bound_u64(const u64* a, int64_t n, u64 k) {   // first i with a[i] > k
    int64_t lo = 0, hi = n;
    while (lo < hi) { int64_t m = (lo + hi) >> 1; if (a[m] <= k) lo = m + 1; else hi = m; }
    return lo;
}

// [first key > klo, first key >= khi) of the junction keys (klo < khi): what std::set::upper_bound / lower_bound
// give merge_chain (long_spanning_reads.cpp:1311-1330).  A closure looks at a 9-base span of left positions, so with
// the bucket index this is one index fetch and a scan of the zero to two keys found there instead of two binary
// searches of ~15 dependent loads each.
THJ_HD void junc_range(const SpanSets& S, u64 klo, u64 khi, int64_t& lb, int64_t& ub) {
    if (!S.junc_bucket) {
        lb = upper_bound_u64(S.junc_keys, S.n_juncs, klo);
        ub = lower_bound_u64(S.junc_keys, S.n_juncs, khi);
        return;
    }
    int64_t b0 = (int64_t)((klo >> 30) >> JUNC_BUCKET_SHIFT), b1 = (int64_t)((khi >> 30) >> JUNC_BUCKET_SHIFT) + 1;
    if (b0 > S.n_buckets) b0 = S.n_buckets;
    if (b1 > S.n_buckets) b1 = S.n_buckets;
    int64_t i = S.junc_bucket[b0];
    const int64_t e = S.junc_bucket[b1];
    while (i < e && S.junc_keys[i] <= klo) ++i;
    lb = i;
    while (i < e && S.junc_keys[i] < khi) ++i;
    ub = i;
}

// The chain's sequence in genome-forward orientation: the read, or its reverse complement.  A view, not a
// copy: pieces are fetched from the read's planes in global memory (L1/L2-resident) and complemented on the fly.
struct SeqView { const u64* rp; int W, len; bool rc; };
THJ_HD SeqView seq_forward(const u64* rp, int W, int rl) { SeqView s; s.rp = rp; s.W = W; s.len = rl; s.rc = false; return s; }
THJ_HD SeqView seq_revcomp(const u64* rp, int W, int rl) { SeqView s; s.rp = rp; s.W = W; s.len = rl; s.rc = true; return s; }
THJ_HD Planes seq_fetch(const SeqView& s, int start, int len) {
    if (!s.rc) return r_fetch(s.rp, s.W, start, len);
    return rc_piece(r_fetch(s.rp, s.W, s.len - start - len, len), len);
}
// base code 0..3, 4 = N
THJ_HD int seq_code(const SeqView& s, int i) {
    int j = s.rc ? s.len - 1 - i : i;
    int w = j >> 6, b = j & 63;
    if ((s.rp[2 * s.W + w] >> b) & 1ull) return 4;
    int c = (int)(((s.rp[w] >> b) & 1ull) | (((s.rp[s.W + w] >> b) & 1ull) << 1));
    return s.rc ? 3 - c : c;
}
// rc(read) == read ?  (merge_chain :1966-1978 compares the joined sequence with the read)
THJ_HD bool read_is_own_revcomp(const u64* rp, int W, int rl) {
    for (int off = 0; off < rl; off += 64) {
        int l = rl - off < 64 ? rl - off : 64;
        Planes f = r_fetch(rp, W, off, l);
        Planes r = rc_piece(r_fetch(rp, W, rl - off - l, l), l);
        if (f.lo != r.lo || f.hi != r.hi || f.nm != r.nm) return false;
    }
    return true;
}
THJ_HD int plane_code(const Planes& p, int b) {
    if ((p.nm >> b) & 1ull) return 4;
    return (int)(((p.lo >> b) & 1ull) | (((p.hi >> b) & 1ull) << 1));
}

// Dna5 mismatch mask of up to 64 bases: N equals N, N differs from a base
THJ_HD u64 dna5_mism(const Planes& a, const Planes& b, int len) {
    return ((a.lo ^ b.lo) | (a.hi ^ b.hi) | (a.nm ^ b.nm)) & lowmask(len);
}
// the same whatever the lo / hi bits of an N position hold (base codes 0..3, 4 = N, compared as numbers)
THJ_HD u64 raw_mism(const Planes& a, const Planes& b, int len) {
    return ((((a.lo ^ b.lo) | (a.hi ^ b.hi)) & ~(a.nm & b.nm)) | (a.nm ^ b.nm)) & lowmask(len);
}

// ---- check_editdist_consistency (bwt_map.cpp:2349-2465) ------------------------------------
THJ_HD bool check_editdist(const Genome& g, const Aln& h, const SeqView& sv) {
    if (g_len(g, h.ref_id) == 0) return false;
    int pos_seq = 0;
    int64_t pos_ref = h.left;
    int mismatch = 0, n_mism = 0;
    for (int i = 0; i < h.n; ++i) {
        int op = cig_op(h.c[i]);
        int len = (int)cig_len(h.c[i]);
        if (op == OP_MATCH) {
            for (int o = 0; o < len; o += 64) {
                int l = len - o < 64 ? len - o : 64;
                if (pos_seq + o + l > sv.len) l = sv.len - pos_seq - o;
                if (l <= 0) break;
                Planes r = g_fetch(g, h.ref_id, pos_ref + o);
                Planes s = seq_fetch(sv, pos_seq + o, l);
                mismatch += popc(dna5_mism(r, s, l));
                n_mism += popc(r.nm & s.nm & lowmask(l));
            }
            pos_seq += len; pos_ref += len;
        } else if (op == OP_INS) pos_seq += len;
        else if (op == OP_DEL || op == OP_REF_SKIP) pos_ref += len;
    }
    return mismatch == (int)h.mm || mismatch + n_mism == (int)h.mm;
}

// ---- one adjacent pair of merge_chain's main loop (long_spanning_reads.cpp:900-1870) -------------
// prev covers read bases [.., P) of the chain's forward sequence `sv`, curr covers [P, P+curr.rlen).
// The search for the closing event is scalar work shared by every tier; how the two cigars are spliced is
// left to the caller's representation (arrays in the generic tier, registers in the lean tier).
enum { CL_FAIL = 0, CL_KEEP = 1, CL_INS = 2, CL_JUNC = 3 };
struct Closure {
    int kind;
    int itpr, ilen;          // CL_INS: bases of prev's tail that move right of the insertion, insertion length
    int dtl, skip, janti;    // CL_JUNC: boundary shift (-4..4), skipped reference bases, junction strand
    int mismatch;            // change of the mismatch count
};

THJ_HD Closure closure_search(const Genome& g, const Params& p, const SpanSets& S, const SeqView& sv, int P, uint32_t ref,
                              int prev_right, int prev_end_len, int curr_left, int curr_front_len, bool same_strand) {
    Closure cl;
    cl.kind = CL_FAIL; cl.itpr = cl.ilen = cl.dtl = cl.skip = cl.janti = cl.mismatch = 0;
    const int lbnd = prev_right - 4, rbnd = curr_left + 4;
    const int dist = curr_left - prev_right;
    if (dist < 0 && dist >= -p.max_insertion_length && same_strand) {
        // ---- insertion closure :1010-1306
        if (g_len(g, ref) == 0) return cl;
        int64_t lb = upper_bound_u64(S.ins_keys, S.n_ins, ins_key(g, ref, (uint32_t)lbnd, 0));
        int64_t ub = upper_bound_u64(S.ins_keys, S.n_ins, ins_key(g, ref, (uint32_t)rbnd, p.max_insertion_length));
        const u64 cbase = (u64)g.contig_blk[ref - 1] * 64ull;
        bool found = false;
        for (; lb < ub; ++lb) {
            const u64 k = S.ins_keys[lb];
            const int ilen = (int)(k & 15);
            const int ileft = (int)((int64_t)(k >> 4) - 1 - (int64_t)cbase);
            if (ilen != prev_right - curr_left) continue;
            const int itpr = prev_right - ileft - 1;
            const int clti = ileft - curr_left + 1;
            if (itpr > prev_end_len || clti > curr_front_len) continue;
            const uint32_t iseq = S.ins_seq[lb];
            int trm = 0, ins_mm = 0;
            if (itpr > 0) {
                Planes rf = g_fetch(g, ref, (int64_t)ileft + 1);       // ref[ileft+1, prev_right)
                for (int ri = 0; ri < itpr; ++ri) {
                    int r = plane_code(rf, ri);
                    int o = seq_code(sv, P - itpr + ri);
                    if (r == 4 || r != o) ++trm;
                    if (ri < ilen) {
                        int ic = (int)((iseq >> (3 * ri)) & 7u);
                        if (ic == 4 || ic != o) { ++ins_mm; break; }
                    } else {
                        int r2 = plane_code(rf, ri - ilen);
                        if (r2 == 4 || r2 != o) --trm;
                    }
                }
            }
            if (clti > 0) {
                Planes rf = g_fetch(g, ref, curr_left);                 // ref[curr.left, ileft+1)
                for (int ri = 0; ri < clti; ++ri) {
                    int sp = clti - ri - 1, ip = ilen - ri - 1;
                    int r = plane_code(rf, sp);
                    int o = seq_code(sv, P + sp);
                    if (r == 4 || r != o) ++trm;
                    if (ri < ilen) {
                        int ic = (int)((iseq >> (3 * ip)) & 7u);
                        if (ic == 4 || ic != o) { ++ins_mm; break; }
                    } else {
                        int r2 = plane_code(rf, sp + ilen);
                        if (r2 == 4 || r2 != o) --trm;
                    }
                }
            }
            if (found) { cl.kind = CL_FAIL; return cl; }                               // :1243-1247
            if (ins_mm == 0) {
                found = true;
                cl.kind = CL_INS; cl.itpr = itpr; cl.ilen = ilen; cl.mismatch = -trm;
            }
        }
        return cl;                                                                     // CL_FAIL unless found
    }
    if (dist > 0 && dist <= p.max_report_intron && same_strand) {
        // ---- junction / deletion closure :1311-1591
        if (THJ_EXPF(1 << 20)) return cl;
        if (g_len(g, ref) == 0) return cl;
        // the candidates: the keys in (klo, khi), in order.  With the bucket index: from the first key of klo's 256-base bucket on,
        // FOUR KEYS A ROUND TRIP (a bucket of a deletion-rich exon holds a dozen keys; walked one dependent load at a time the
        // scan was 0.14 of thj_k_join_closure's 0.21 ms); without it: the two binary searches
        const u64 klo = junc_key(g, ref, (uint32_t)lbnd, (uint32_t)(rbnd - 8), true), khi = junc_key(g, ref, (uint32_t)(lbnd + 8), (uint32_t)rbnd, false);
        int64_t i, e;
        if (THJ_EXPF(1 << 22)) return cl;
        if (!S.junc_bucket) { i = upper_bound_u64(S.junc_keys, S.n_juncs, klo); e = lower_bound_u64(S.junc_keys, S.n_juncs, khi); }
        else {
            int64_t b0 = (int64_t)((klo >> 30) >> JUNC_BUCKET_SHIFT), b1 = (int64_t)((khi >> 30) >> JUNC_BUCKET_SHIFT) + 1;
            if (b0 > S.n_buckets) b0 = S.n_buckets;
            if (b1 > S.n_buckets) b1 = S.n_buckets;
            i = S.junc_bucket[b0]; e = S.junc_bucket[b1];
            // a crowded bucket (a hundred deletions in one exon of the bench's mix): bisect to the first key past klo instead of walking there
            if (e - i > 8) i += upper_bound_u64(S.junc_keys + i, e - i, klo);
        }
        const u64 cbase = (u64)g.contig_blk[ref - 1] * 64ull;
        int best_diff = 0xff;
        bool past = false;
        for (; i < e && !past; i += 4) {
            u64 kk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kk[j] = S.junc_keys[i + j < e ? i + j : e - 1];
            uint32_t cand = 0;                    // which of the four lie in (klo, khi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = !past && i + j < e && kk[j] > klo;
                if (in && kk[j] >= khi) past = true;
                cand |= (in && kk[j] < khi) ? 1u << j : 0u;
            }
            while (cand) {
                const int j = ctz((u64)cand);
                cand &= cand - 1;
                const u64 k = j == 0 ? kk[0] : (j == 1 ? kk[1] : (j == 2 ? kk[2] : kk[3]));
                const int jl = (int)((int64_t)(k >> 30) - 1 - (int64_t)cbase);
                const int jr = jl + (int)((k >> 1) & ((1ull << 29) - 1));
                const int dtl = jl - prev_right + 1, dtr = jr - curr_left;
                if (!(dtl >= -4 && dtl <= 4 && dtr >= -4 && dtr <= 4 && dtl == dtr)) continue;
                if (dtl > curr_front_len || -dtl > prev_end_len) continue;
                int new_mm = 0, old_mm = 0;
                if (dtl != 0 && !THJ_EXPF(1 << 21)) {
                    // the boundary moves by |dtl| <= 4 bases: those bases of the read against the reference on the new side and on the old
                    // one (raw char vs Dna5: N == N, N != a base) -- both genome pieces and the read piece fetched together, compared as planes
                    const int ad = dtl > 0 ? dtl : -dtl;
                    // dtl > 0: new_cmp = ref[prev_right, jl+1), old_cmp = ref[curr.left, jr), the read's bases [P, P + dtl)
                    // dtl < 0: new_cmp = ref[jr, curr.left),    old_cmp = ref[jl+1, prev_right), the read's bases [P - ad, P)
                    const Planes nr = g_fetch_abs(g, cbase + (u64)(dtl > 0 ? prev_right : jr));
                    const Planes orf = g_fetch_abs(g, cbase + (u64)(dtl > 0 ? curr_left : jl + 1));
                    const Planes sq = seq_fetch(sv, dtl > 0 ? P : P - ad, ad);
                    new_mm = popc(raw_mism(nr, sq, ad));
                    old_mm = popc(raw_mism(orf, sq, ad));
                }
                const int diff = new_mm - old_mm;
                if (diff >= best_diff || new_mm >= 2) continue;
                best_diff = diff;
                cl.kind = CL_JUNC; cl.dtl = dtl; cl.skip = jr - jl - 1; cl.janti = (int)(k & 1ull); cl.mismatch = diff;
            }
        }
        return cl;                                                                     // CL_FAIL unless found
    }
    if (dist == 0 && same_strand) cl.kind = CL_KEEP;       // anything else: check_fusion with an empty fusion set
    return cl;
}

// PAIR_KEEP: the two abut (dist == 0) and stay separate chain elements; PAIR_MERGED: `m` replaces both;
// PAIR_FAIL: merge_chain returns BowtieHit().
enum { PAIR_FAIL = 0, PAIR_KEEP = 1, PAIR_MERGED = 2 };

THJ_HD int close_pair(const Genome& g, const Params& p, const SpanSets& S, const SeqView& sv, int P, const Aln& prev,
                      const Aln& curr, Aln& m) {
    if (!(op_is_match(cig_op(prev.c[prev.n - 1])) || op_is_match(cig_op(curr.c[0])))) return PAIR_FAIL;     // :924-928
    const bool psp = aln_spliced(prev), csp = aln_spliced(curr);
    if (psp && csp && prev.asplice != curr.asplice) return PAIR_FAIL;                                        // :936-943
    if (prev.ref_id != curr.ref_id) return PAIR_FAIL;      // check_fusion with an empty fusion set (:1596-1818)
    const Closure cl = closure_search(g, p, S, sv, P, prev.ref_id, aln_right(prev), (int)cig_len(prev.c[prev.n - 1]), curr.left,
                                      (int)cig_len(curr.c[0]), prev.anti == curr.anti);
    if (cl.kind == CL_FAIL) return PAIR_FAIL;
    if (cl.kind == CL_KEEP) return PAIR_KEEP;
    int anti_closure = psp ? prev.asplice : curr.asplice;
    uint32_t nc[SPAN_MAXC + 8];
    int nn = prev.n;
    for (int q = 0; q < prev.n; ++q) nc[q] = prev.c[q];
    int first;                 // 0: curr's first op survives (with length `flen`), 1: it is consumed
    uint32_t flen;
    if (cl.kind == CL_INS) {
        uint32_t bl = (cig_len(nc[nn - 1]) - (uint32_t)cl.itpr) & 0x0FFFFFFFu;
        if (bl == 0) --nn; else nc[nn - 1] = cig(cig_op(nc[nn - 1]), bl);
        nc[nn++] = cig(OP_INS, (uint32_t)cl.ilen);
        flen = (cig_len(curr.c[0]) + (uint32_t)(cl.itpr - cl.ilen)) & 0x0FFFFFFFu;
        first = flen > 0 ? 0 : 1;
    } else {
        int nlb = (int)cig_len(nc[nn - 1]) + cl.dtl;
        int nrf = (int)cig_len(curr.c[0]) - cl.dtl;
        if (nlb > 0) nc[nn - 1] = cig(cig_op(nc[nn - 1]), (uint32_t)nlb); else --nn;
        if ((uint32_t)cl.skip <= (uint32_t)p.max_deletion_length) nc[nn++] = cig(OP_DEL, (uint32_t)cl.skip);
        else { nc[nn++] = cig(OP_REF_SKIP, (uint32_t)cl.skip); anti_closure = cl.janti; }
        flen = (uint32_t)nrf;
        first = nrf > 0 ? 0 : 1;
    }
    for (int q = first; q < curr.n; ++q) {
        if (nn >= SPAN_MAXC + 8) return PAIR_FAIL;
        nc[nn++] = q == 0 ? cig(cig_op(curr.c[0]), flen) : curr.c[q];
    }
    if (nn > SPAN_MAXC) return PAIR_FAIL;                   // capacity (documented limit)
    int mismatches = (int)prev.mm + (int)curr.mm + cl.mismatch;                        // :1822-1870
    m.ref_id = prev.ref_id; m.left = prev.left; m.n = nn;
    for (int q = 0; q < nn; ++q) m.c[q] = nc[q];
    m.anti = prev.anti; m.asplice = (uint8_t)anti_closure;
    m.mm = (uint8_t)mismatches;
    m.ed = (uint8_t)(mismatches + cig_gap_len(nc, nn));
    m.rlen = prev.rlen + curr.rlen; m.valid = 1;
    return PAIR_MERGED;
}

// the pre-check of merge_chain (:843-891): at most one fusion-like gap
THJ_HD bool gap_is_fusion_like(const Params& p, int gap) {
    int maxi = p.max_report_intron < 10000000 ? p.max_report_intron : 10000000;
    return gap < -p.max_insertion_length || (gap > p.max_deletion_length && (gap < p.min_report_intron || gap > maxi));
}

// final concatenation state of merge_chain (:1888-1944), fed one chain element at a time
struct ChainOut {
    Aln out; bool saw_as, saw_s; int num_mm;
};
THJ_HD void chain_out_init(ChainOut& c) { c.out.n = 0; c.saw_as = c.saw_s = false; c.num_mm = 0; }
THJ_HD bool chain_out_add(ChainOut& c, const Aln& e) {
    c.num_mm += e.mm;
    if (aln_spliced(e)) {
        if (e.asplice) { if (c.saw_s) return false; c.saw_as = true; }
        else { if (c.saw_as) return false; c.saw_s = true; }
    }
    Aln& o = c.out;
    int b0 = 0;
    if (o.n > 0 && cig_op(o.c[o.n - 1]) == cig_op(e.c[0])) {
        o.c[o.n - 1] = cig(cig_op(o.c[o.n - 1]), cig_len(o.c[o.n - 1]) + cig_len(e.c[0]));
        b0 = 1;
    }
    for (int b = b0; b < e.n; ++b) { if (o.n >= SPAN_MAXC) return false; o.c[o.n++] = e.c[b]; }
    return true;
}
THJ_HD void chain_out_finish(ChainOut& c, uint32_t ref_id, int left, int antisense, int rlen) {
    Aln& o = c.out;
    o.ref_id = ref_id; o.left = left;
    o.anti = (uint8_t)antisense; o.asplice = c.saw_as ? 1 : 0;
    o.mm = (uint8_t)c.num_mm; o.ed = (uint8_t)(c.num_mm + cig_gap_len(o.c, o.n));
    o.rlen = rlen; o.valid = 1;
}

// ---- merge_chain (long_spanning_reads.cpp:805-2038), fusion_dir == FUSION_NOTHING ----------
// chain[0..n) ordered left to right; sv = the chain's forward-orientation sequence.
THJ_HD bool merge_chain(const Genome& g, const Params& p, const SpanSets& S, const SeqView& sv, Aln* chain, int n,
                        Aln& out) {
    const int left = chain[0].left;
    int antisense = chain[0].anti;
    int old_read_length = 0;
    for (int i = 0; i < n; ++i) old_read_length += aln_read_len(chain[i]);
    {
        int num_fusions = 0;
        for (int k = 1; k < n; ++k) {
            if (chain[k - 1].ref_id != chain[k].ref_id) ++num_fusions;
            else if (gap_is_fusion_like(p, chain[k].left - aln_right(chain[k - 1]))) ++num_fusions;
            if (num_fusions >= 2) return false;
        }
    }
    int pi = 0, ci = 1;
    int P = chain[0].rlen;            // read bases covered by chain[0..pi]
    while (ci < n) {
        antisense = chain[pi].anti;
        Aln m;
        int r = close_pair(g, p, S, sv, P, chain[pi], chain[ci], m);
        if (r == PAIR_FAIL) return false;
        P += chain[ci].rlen;
        if (r == PAIR_MERGED) {
            chain[pi] = m;
            for (int q = ci; q + 1 < n; ++q) chain[q] = chain[q + 1];
            --n;
            ci = pi + 1;
        } else { ++pi; ++ci; }
    }
    ChainOut co;
    chain_out_init(co);
    for (int s = 0; s < n; ++s) if (!chain_out_add(co, chain[s])) return false;
    chain_out_finish(co, chain[0].ref_id, left, antisense, sv.len);
    out = co.out;
    if (aln_read_len(out) != old_read_length || !check_editdist(g, out, sv)) return false;   // :2014-2033
    return true;
}

// valid_hit (long_spanning_reads.cpp:2045-2099)
template <class A>
THJ_HD bool valid_hit(const Params& p, const A& h) {
    if (!h.valid) return false;
    for (int i = 1; i < h.n; ++i) {
        int cop = cig_op(cg(h, i)), pop = cig_op(cg(h, i - 1));
        uint32_t len = cig_len(cg(h, i));
        if (!op_is_match(cop) && !op_is_match(pop)) return false;
        if (cop == OP_INS && len > (uint32_t)p.max_insertion_length) return false;
        if (cop == OP_DEL && len > (uint32_t)p.max_deletion_length) return false;
        if (cop == OP_REF_SKIP && len < (uint32_t)p.min_report_intron) return false;
    }
    return op_is_match(cig_op(cg(h, 0))) && op_is_match(cig_op(cg(h, h.n - 1)));
}

THJ_HD Aln aln_from_hit(const SpanHit& h, int seg, int L, int rl) {
    Aln a;
    a.ref_id = h.ref_id; a.left = h.left;
    a.n = (int)(h.meta >> 24); if (a.n > 5) a.n = 5;
    for (int i = 0; i < a.n; ++i) a.c[i] = h.cigar[i];
    a.anti = (h.meta & SH_ANTI) ? 1 : 0; a.asplice = (h.meta & SH_ASPLICE) ? 1 : 0;
    a.mm = (uint8_t)((h.meta >> 8) & 0xFF); a.ed = (uint8_t)((h.meta >> 16) & 0xFF);
    int st = seg * L; if (st > rl) st = rl;
    int ln = (h.meta & SH_END) ? rl - st : L; if (ln > rl - st) ln = rl - st;
    a.rlen = ln; a.valid = 1;
    return a;
}

// BowtieHit::operator< and == (bwt_map.h:167-207) for one read's joined hits
THJ_HD bool aln_less(const Aln& a, const Aln& b) {
    if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
    if (a.left != b.left) return a.left < b.left;
    if (a.anti != b.anti) return a.anti < b.anti;
    if (a.mm != b.mm) return a.mm < b.mm;
    if (a.ed != b.ed) return a.ed < b.ed;
    if (a.n != b.n) return a.n < b.n;
    for (int i = 0; i < a.n; ++i)
        if (a.c[i] != b.c[i]) {
            int oa = cig_op(a.c[i]), ob = cig_op(b.c[i]);
            return oa < ob || (oa == ob && cig_len(a.c[i]) < cig_len(b.c[i]));
        }
    return false;
}
THJ_HD bool aln_eq(const Aln& a, const Aln& b) {
    if (a.ref_id != b.ref_id || a.anti != b.anti || a.left != b.left || a.asplice != b.asplice || a.ed != b.ed || a.n != b.n)
        return false;
    for (int i = 0; i < a.n; ++i) if (a.c[i] != b.c[i]) return false;
    return true;
}

// MD:Z string built in registers: 40 chars in five 64-bit words (little-endian byte order = memory order)
struct MdBuf { u64 w[5]; int len; };
THJ_HD void md_init(MdBuf& m) { m.w[0] = m.w[1] = m.w[2] = m.w[3] = m.w[4] = 0; m.len = 0; }
THJ_HD void md_push(MdBuf& m, char c) {
    if (m.len < 40) {
        const int k = m.len >> 3;
        const u64 v = (u64)(uint8_t)c << ((m.len & 7) * 8);
        m.w[0] |= k == 0 ? v : 0; m.w[1] |= k == 1 ? v : 0; m.w[2] |= k == 2 ? v : 0; m.w[3] |= k == 3 ? v : 0; m.w[4] |= k == 4 ? v : 0;
    }
    ++m.len;
}
// Up to four characters at once (`tok` = the characters in memory order, low byte first).  One masked OR per word
// of the buffer instead of one per character and word: the MD string is built from <run length><mismatch base> tokens.
THJ_HD void md_append(MdBuf& m, uint32_t tok, int n) {
    const int k = m.len >> 3, sh = (m.len & 7) * 8;      // k >= 5 (string already too long): nothing is stored
    const u64 lo = (u64)tok << sh;
    const u64 hi = sh > 32 ? (u64)tok >> (64 - sh) : 0ull;
#pragma unroll
    for (int i = 0; i < 5; ++i) m.w[i] |= (k == i ? lo : 0ull) | (k + 1 == i ? hi : 0ull);
    m.len += n;
}
// decimal digits of v (0..999: runs are shorter than a read, at most 256 bases) as a token; returns their number
THJ_HD int md_int_token(int v, uint32_t& tok) {
    const uint32_t d2 = (uint32_t)v / 100u, r = (uint32_t)v - d2 * 100u, d1 = r / 10u, d0 = r - d1 * 10u;
    const uint32_t t3 = (0x30u + d2) | ((0x30u + d1) << 8) | ((0x30u + d0) << 16);
    const uint32_t t2 = (0x30u + d1) | ((0x30u + d0) << 8);
    tok = v >= 100 ? t3 : (v >= 10 ? t2 : 0x30u + d0);
    return v >= 100 ? 3 : (v >= 10 ? 2 : 1);
}
THJ_HD void md_put_int(MdBuf& m, int v) {
    uint32_t tok;
    const int n = md_int_token(v, tok);
    md_append(m, tok, n);
}
// <run length><base>: the token of one mismatch
THJ_HD void md_put_int_char(MdBuf& m, int v, char c) {
    uint32_t tok;
    const int n = md_int_token(v, tok);
    md_append(m, tok | ((uint32_t)(uint8_t)c << (8 * n)), n + 1);
}

struct OutAln {             // == thj_aln, 128 bytes
    uint32_t read_idx;
    uint32_t ref_id;
    int32_t left;
    uint8_t flags, mismatches, edit_dist, n_cigar;
    int16_t AS;
    uint8_t XM, XO, XG, md_len;
    uint16_t order;         // position among the read's output records
    uint32_t cigar[SPAN_MAXC];
    char md[40];
};

struct Extras { int AS, XM, XO, XG, both_n; MdBuf md; };

// bowtie_sam_extra (bwt_map.cpp:2467-2648).  qual = this read's phred+33 bytes; qual_rev: the joined hit's
// qual is the reversed read qual (merge_chain :1966-1978).  Also counts the N==N positions, which is what
// check_editdist_consistency (:2349-2465) needs beside XM.  Returns false when the MD string does not fit.
template <class A>
THJ_HD bool sam_extra(const Genome& g, const Params& p, const A& h, const SeqView& sv, const uint8_t* qual, int qlen,
                      bool qual_rev, Extras& e) {
    int pos_seq = 0, pos_mm = 0, mismatch = 0, opens = 0, conts = 0, AS = 0, both_n = 0;
    int64_t pos_ref = h.left;
    md_init(e.md);
    for (int i = 0; i < h.n; ++i) {
        const uint32_t ci = cg(h, i);
        int op = cig_op(ci);
        int len = (int)cig_len(ci);
        if (op == OP_MATCH) {
            for (int off = 0; off < len; off += 64) {
                int l = len - off < 64 ? len - off : 64;
                if (pos_seq + off + l > sv.len) l = sv.len - pos_seq - off;
                if (l <= 0) break;
                Planes r = g_fetch(g, h.ref_id, pos_ref + off);
                Planes s = seq_fetch(sv, pos_seq + off, l);
                u64 mm = dna5_mism(r, s, l);
                u64 bothn = r.nm & s.nm & lowmask(l);
                AS -= p.bowtie2_penalty_for_N * popc(bothn);        // matching N: still penalised (:2552-2556)
                both_n += popc(bothn);
                int last = 0;
                while (mm) {
                    int b = ctz(mm);
                    mm &= mm - 1;
                    ++mismatch;
                    int sp = pos_seq + off + b;
                    if (sp < qlen) {
                        if (((r.nm | s.nm) >> b) & 1ull) AS -= p.bowtie2_penalty_for_N;
                        else {
                            int q = (int)qual[qual_rev ? qlen - 1 - sp : sp] - 33; if (q > 40) q = 40;
                            // int(min + (max-min)*q/40.0): exact in integers (the fraction is a multiple of 1/40)
                            AS -= p.bowtie2_min_penalty + ((p.bowtie2_max_penalty - p.bowtie2_min_penalty) * q) / 40;
                        }
                    }
                    pos_mm += b - last;
                    md_put_int_char(e.md, pos_mm, "ACGTN"[plane_code(r, b)]);
                    pos_mm = 0; last = b + 1;
                }
                pos_mm += l - last;
            }
            pos_seq += len; pos_ref += len;
        } else if (op == OP_INS) {
            pos_seq += len;
            AS -= p.bowtie2_read_gap_open + p.bowtie2_read_gap_cont * len;
            ++opens; conts += len;
        } else if (op == OP_DEL) {
            AS -= p.bowtie2_ref_gap_open + p.bowtie2_ref_gap_cont * len;
            ++opens; conts += len;
            md_put_int_char(e.md, pos_mm, '^');
            Planes r = g_fetch(g, h.ref_id, pos_ref);
            for (int k = 0; k < len && k < 64; ++k) md_push(e.md, "ACGTN"[plane_code(r, k)]);
            pos_ref += len; pos_mm = 0;
        } else if (op == OP_REF_SKIP) pos_ref += len;
    }
    md_put_int(e.md, pos_mm);
    e.AS = AS; e.XM = mismatch; e.XO = opens; e.XG = conts; e.both_n = both_n;
    return e.md.len <= 40;
}

// one 128-byte thj_aln record assembled in registers (layout of OutAln) and handed to the sink as 32 words
template <class Sink, class A>
THJ_HD void emit_aln(Sink& sink, uint32_t read_idx, int order, const A& h, const Extras& e) {
    uint32_t wds[32];
    wds[0] = read_idx; wds[1] = h.ref_id; wds[2] = (uint32_t)h.left;
    wds[3] = (h.anti ? 1u : 0u) | (h.asplice ? 4u : 0u) | ((uint32_t)h.mm << 8) | ((uint32_t)h.ed << 16) | ((uint32_t)h.n << 24);
    wds[4] = ((uint32_t)e.AS & 0xFFFFu) | ((uint32_t)(e.XM & 0xFF) << 16) | ((uint32_t)(e.XO & 0xFF) << 24);
    // an MD string that does not fit the record's 40 characters is left to the host (md_len = THJ_MD_ON_HOST: thj_md_string)
    wds[5] = (uint32_t)(e.XG & 0xFF) | ((uint32_t)(e.md.len > 40 ? 255 : e.md.len) << 8) | ((uint32_t)(order & 0xFFFF) << 16);
#pragma unroll
    for (int q = 0; q < SPAN_MAXC; ++q) wds[6 + q] = q < h.n ? cg_fixed(h, q) : 0u;
#pragma unroll
    for (int q = 0; q < 5; ++q) { wds[22 + 2 * q] = (uint32_t)e.md.w[q]; wds[23 + 2 * q] = (uint32_t)(e.md.w[q] >> 32); }
    sink.emit_words(wds);
}

enum { SPAN_OK = 0, SPAN_TOO_MANY_JOINED = 1, SPAN_MD_OVERFLOW = 2, SPAN_NEED_GENERIC = 3 };
enum { SPAN_INCOMPAT = 5 };    // span_read_lean only, never counted as a status: nothing joined because the read's only chain is not compatible the plain way

// One read: JoinSegmentsWorker body (long_spanning_reads.cpp:2767-2831).  Emits through
// sink.emit(const OutAln&) in output order; returns a SPAN_* status.
template <class Sink>
THJ_HD int span_read(const Genome& g, const Params& p, const SpanSets& S, const SpanHit* hits, const uint32_t* so, int nseg,
                     const u64* rp, int W, int rl, const uint8_t* qual, uint32_t read_idx, Sink& sink, Aln* ext = nullptr, int ext_cap = 0) {
    if (so[1] == so[0]) return SPAN_OK;                         // worker iterates over first-segment groups
    int nsegs = 0;
    while (nsegs < nseg && so[nsegs + 1] > so[nsegs]) ++nsegs;   // look_right stops at the first empty segment (:151)
    if (nsegs > SPAN_MAXSEG) nsegs = SPAN_MAXSEG;
    if (!(hits[so[nsegs - 1]].meta & SH_END)) return SPAN_OK;    // :2777-2785
    if (p.bowtie2)
        for (int s = 0; s < nsegs; ++s)
            if ((int)(so[s + 1] - so[s]) > p.max_seg_multihits) return SPAN_OK;   // :2625-2632
    const int L = p.segment_length;
    // ext: a caller-owned buffer of 2 * ext_cap alignments for the reads whose joined alignments do not fit SPAN_MAXJOIN (the
    // second half is the merge sort's scratch); without it such a read reports SPAN_TOO_MANY_JOINED and emits nothing
    Aln joined_local[SPAN_MAXJOIN];
    Aln* joined = ext ? ext : joined_local;
    const int cap = ext ? ext_cap : SPAN_MAXJOIN;
    int nj = 0;
    int status = SPAN_OK;
    SeqView fwd = seq_forward(rp, W, rl);
    SeqView rev = seq_revcomp(rp, W, rl);
    uint32_t idx[SPAN_MAXSEG];
    Aln stack[SPAN_MAXSEG];
    for (uint32_t i0 = so[0]; i0 < so[1]; ++i0) {                // :2634-2664
        stack[0] = aln_from_hit(hits[i0], 0, L, rl);
        int num_try = 10000;
        int depth = 1;
        idx[1] = so[1];
        // iterative dfs_seg_hits (:2222-2610, fusion_search == false)
        while (depth >= 1) {
            if (num_try <= 0) break;
            if (depth == nsegs) {
                --num_try;
                // merge_segment_chain :2101-2220
                Aln bh;
                bool ok;
                if (nsegs > 1) {
                    Aln chain[SPAN_MAXSEG];
                    const bool anti = stack[0].anti;
                    for (int q = 0; q < nsegs; ++q) chain[q] = anti ? stack[nsegs - 1 - q] : stack[q];
                    ok = merge_chain(g, p, S, anti ? rev : fwd, chain, nsegs, bh);
                } else { bh = stack[0]; ok = true; }
                if (ok && valid_hit(p, bh)) {
                    if (nj < cap) joined[nj++] = bh; else return SPAN_TOO_MANY_JOINED;      // (nothing is emitted then: the read is done again with room, or reported)
                }
                --depth;
                continue;
            }
            if (idx[depth] >= so[depth + 1]) { --depth; continue; }
            const SpanHit sh = hits[idx[depth]++];
            Aln cand = aln_from_hit(sh, depth, L, rl);
            const Aln& prev = stack[depth - 1];
            bool okc = false;
            if (prev.ref_id == cand.ref_id && prev.anti == cand.anti) {
                int dist = prev.anti ? prev.left - aln_right(cand) : cand.left - aln_right(prev);   // :2352-2378, :2531-2556
                okc = dist <= p.max_report_intron && dist >= -p.max_insertion_length;
            }
            if (okc) {
                stack[depth] = cand;
                ++depth;
                if (depth < nsegs) idx[depth] = so[depth];
            }
        }
    }
    if (THJ_EXPF(4096)) return SPAN_OK;
    if (status == SPAN_TOO_MANY_JOINED) return status;          // nothing is emitted: the read is done again with room (or reported)
    // sort + unique (:2805-2807): insertion sort (stable; what std::sort does below 16 elements); a stable merge sort for the
    // long lists of the external buffer
    if (!ext || nj <= 64) {
        for (int i = 1; i < nj; ++i) {
            Aln t = joined[i]; int k = i;
            while (k > 0 && aln_less(t, joined[k - 1])) { joined[k] = joined[k - 1]; --k; }
            joined[k] = t;
        }
    } else {
        Aln* a = joined; Aln* b = joined + ext_cap;
        for (int width = 1; width < nj; width <<= 1) {
            for (int lo = 0; lo < nj; lo += 2 * width) {
                const int mid = lo + width < nj ? lo + width : nj, hi = lo + 2 * width < nj ? lo + 2 * width : nj;
                int i = lo, j = mid, k = lo;
                while (i < mid && j < hi) b[k++] = aln_less(a[j], a[i]) ? a[j++] : a[i++];
                while (i < mid) b[k++] = a[i++];
                while (j < hi) b[k++] = a[j++];
            }
            Aln* t = a; a = b; b = t;
        }
        if (a != joined) for (int i = 0; i < nj; ++i) joined[i] = a[i];
    }
    int w = 0;
    for (int i = 0; i < nj; ++i) if (w == 0 || !aln_eq(joined[w - 1], joined[i])) joined[w++] = joined[i];
    nj = w;
    int order = 0;
    for (int i = 0; i < nj; ++i) {
        const Aln& h = joined[i];
        int gapl = (uint8_t)(h.ed - h.mm);
        if ((int)h.mm > p.read_mismatches || gapl > p.read_gap_length || (int)h.ed > p.read_edit_dist) continue;   // :2810-2813
        // merge_chain :1966-1978: qual reversed when the joined sequence differs from the read;
        // a single-segment hit carries the BAM record's own SEQ/QUAL (reversed when antisense)
        bool qrev;
        const SeqView& sv = h.anti ? rev : fwd;
        if (nsegs == 1) qrev = h.anti;
        else qrev = h.anti ? !read_is_own_revcomp(rp, W, rl) : false;
        Extras e;
        sam_extra(g, p, h, sv, qual, rl, qrev, e);
        emit_aln(sink, read_idx, order++, h, e);
    }
    return status;
}


// ---- lean tier: reads whose segments all have exactly one hit (the overwhelmingly common case) ----------
// Same results as span_read for those reads with a fraction of the private state: the DFS is a single
// chain, merge_chain runs as a stream (accumulated output + previous element + current hit), nothing to
// sort/unique, and the edit-distance consistency check (bwt_map.cpp:2349-2465) shares its plane pass with
// bowtie_sam_extra (:2467-2648).  Any other read returns SPAN_NEED_GENERIC and goes to span_read.
THJ_HD RAln raln_from_hit(const SpanHit& h, int seg, int L, int rl) {
    RAln a;
    a.ref_id = h.ref_id; a.left = h.left;
    a.n = (int)(h.meta >> 24); if (a.n > 5) a.n = 5;
#pragma unroll
    for (int i = 0; i < LEAN_C; ++i) a.c.v[i] = (i < 5 && i < a.n) ? h.cigar[i < 5 ? i : 0] : 0u;
    a.anti = (h.meta & SH_ANTI) ? 1 : 0; a.asplice = (h.meta & SH_ASPLICE) ? 1 : 0;
    a.mm = (int)((h.meta >> 8) & 0xFF); a.ed = (int)((h.meta >> 16) & 0xFF);
    int st = seg * L; if (st > rl) st = rl;
    int ln = (h.meta & SH_END) ? rl - st : L; if (ln > rl - st) ln = rl - st;
    a.rlen = ln; a.valid = 1;
    return a;
}

// final concatenation of merge_chain (:1888-1944) on registers; 0 = ok, 1 = merge_chain fails, 2 = out of ops
struct RChainOut { RCig c; int n; bool saw_as, saw_s; int num_mm; };
THJ_HD int rchain_add(RChainOut& o, const RAln& e) {
    o.num_mm += e.mm;
    if (rc_spliced(e.c, e.n)) {
        if (e.asplice) { if (o.saw_s) return 1; o.saw_as = true; }
        else { if (o.saw_as) return 1; o.saw_s = true; }
    }
    int b0 = 0;
    if (o.n > 0) {
        const uint32_t last = o.c.get(o.n - 1);
        if (cig_op(last) == cig_op(e.c.v[0])) { o.c.set(o.n - 1, cig(cig_op(last), cig_len(last) + cig_len(e.c.v[0]))); b0 = 1; }
    }
    if (o.n + e.n - b0 > LEAN_C) return 2;
    // append e's ops b0.. at o.n..: e's array shifted up by k = o.n - b0 slots with a three-stage barrel shifter (one
    // select per slot and stage), then merged above o.n -- instead of one eight-way select chain per appended op
    const int k = o.n - b0, end = o.n + e.n - b0;
    uint32_t t[LEAN_C];
#pragma unroll
    for (int i = 0; i < LEAN_C; ++i) t[i] = e.c.v[i];
#pragma unroll
    for (int sh = 1; sh < LEAN_C; sh <<= 1) {
        const bool on = (k & sh) != 0;
#pragma unroll
        for (int i = LEAN_C - 1; i >= 0; --i) t[i] = on ? (i >= sh ? t[i - sh] : 0u) : t[i];
    }
#pragma unroll
    for (int i = 0; i < LEAN_C; ++i) o.c.v[i] = (i >= o.n && i < end) ? t[i] : o.c.v[i];
    o.n = end;
    return 0;
}

// Hits staged as 16-byte heads -- contig, left, flags, first cigar op: everything a plain-match hit carries -- with the
// rare longer cigar's tail fetched from global memory on demand.
struct alignas(16) SpanHitHead { uint32_t ref_id; int32_t left; uint32_t meta; uint32_t cigar0; };
// the 16-byte head of hit i of the batch: from the dense head array when the batch has one (half the bytes of the 32-byte
// records, and nothing else in its cache lines), else the first half of the record
THJ_HD Q16 load_head(const SpanHit* hits, const SpanHitHead* gheads, u64 i) {
    return gheads ? *(const Q16*)(gheads + i) : *(const Q16*)(hits + i);
}
THJ_HD SpanHit staged_hit(const SpanHitHead* heads, const SpanHit* g0, int k) {
    const SpanHitHead hh = heads[k];
    SpanHit h;
    h.ref_id = hh.ref_id; h.left = hh.left; h.meta = hh.meta; h.cigar[0] = hh.cigar0;
    h.cigar[1] = h.cigar[2] = h.cigar[3] = h.cigar[4] = 0;
    if ((hh.meta >> 24) > 1) {
        const Q16 t = ((const Q16*)(g0 + k))[1];
        h.cigar[1] = t.x; h.cigar[2] = t.y; h.cigar[3] = t.z; h.cigar[4] = t.w;
    }
    return h;
}
struct StagedHits {         // the chain under construction: segment s's hit is staged hit number (sel >> 4s) & 15
    const SpanHitHead* heads; const SpanHit* g0; u64 sel;
    THJ_HD SpanHit operator[](int s) const { return staged_hit(heads, g0, (int)((sel >> (4 * s)) & 15)); }
};

// ---- lean machinery shared by tiers 1 and 2 -----------------------------------------------------------------
// lean_join: ONE chain -- hits[s] is the hit chosen for segment s -- through merge_chain on register cigars.
enum { LJ_NONE = 0, LJ_OK = 1, LJ_PUNT = 2, LJ_INCOMPAT = 3, LJ_DEFER = 4 };   // no alignment / `res` holds the joined hit / needs more cigar ops than LEAN_C /
                                                                  // no alignment because two neighbours are not compatible the plain way (with fusion search on
                                                                  // such a chain takes a fusion direction; every other failure is the same failure there)
// ABUT: the caller wants only the chains whose neighbours all abut (dist == 0 for every pair: merge_chain leaves each pair as it is,
// :1591, and never looks at a junction, an insertion or the genome) -- any other chain returns LJ_DEFER untouched, and the code that
// searches closures is not even instantiated.  thj_k_join runs its entries this way first and the deferred ones, compacted, afterwards.
template <bool ABUT = false, class Hits>      // Hits: `hits[s]` is the hit chosen for segment s (a plain array, or StagedHits)
THJ_HD int lean_join(const Genome& g, const Params& p, const SpanSets& S, const Hits& hits, int nsegs,
                     const u64* rp, int W, int rl, RAln& res) {
    const int L = p.segment_length;
    const SpanHit h0 = hits[0];
    const bool anti = (h0.meta & SH_ANTI) != 0;
    if (nsegs == 1) {
        res = raln_from_hit(h0, 0, L, rl);
    } else {
        // dfs_seg_hits compatibility of the single candidate per segment (:2352-2378, :2531-2556)
        int old_read_length = 0, num_fusions = 0;
        {
            RAln prev = raln_from_hit(h0, 0, L, rl);
            int prev_right = prev.left + rc_ref_span(prev.c, prev.n);
            old_read_length = rc_read_span(prev.c, prev.n);
            for (int s = 1; s < nsegs; ++s) {
                RAln cand = raln_from_hit(hits[s], s, L, rl);
                const int cand_right = cand.left + rc_ref_span(cand.c, cand.n);
                if (prev.ref_id != cand.ref_id || prev.anti != cand.anti) return LJ_INCOMPAT;
                int dist = prev.anti ? prev.left - cand_right : cand.left - prev_right;
                if (dist > p.max_report_intron || dist < -p.max_insertion_length) return LJ_INCOMPAT;
                if (ABUT && dist != 0) return LJ_DEFER;
                if (gap_is_fusion_like(p, dist)) ++num_fusions;          // merge_chain pre-check (:843-891): same gaps, chain order
                old_read_length += rc_read_span(cand.c, cand.n);
                prev.ref_id = cand.ref_id; prev.anti = cand.anti; prev.left = cand.left; prev_right = cand_right;
            }
        }
        if (num_fusions >= 2) return LJ_NONE;
        if (THJ_EXPF(512)) return LJ_NONE;
        SeqView sv = anti ? seq_revcomp(rp, W, rl) : seq_forward(rp, W, rl);
        RChainOut co;
#pragma unroll
        for (int k = 0; k < LEAN_C; ++k) co.c.v[k] = 0;
        co.n = 0; co.saw_as = co.saw_s = false; co.num_mm = 0;
        const int k0 = anti ? nsegs - 1 : 0, step = anti ? -1 : 1;
        RAln prev = raln_from_hit(hits[k0], k0, L, rl);
        const int left0 = prev.left;
        int P = prev.rlen;
        for (int q = 1; q < nsegs; ++q) {
            const int k = k0 + q * step;
            const RAln curr = raln_from_hit(hits[k], k, L, rl);
            // merge_chain's main loop, one adjacent pair (:900-1870)
            const uint32_t plast = prev.c.get(prev.n - 1), cfirst = curr.c.v[0];
            if (!(op_is_match(cig_op(plast)) || op_is_match(cig_op(cfirst)))) return LJ_NONE;               // :924-928
            const bool psp = rc_spliced(prev.c, prev.n), csp = rc_spliced(curr.c, curr.n);
            if (psp && csp && prev.asplice != curr.asplice) return LJ_NONE;                                  // :936-943
            if (prev.ref_id != curr.ref_id) return LJ_NONE;
            Closure cl;
            if (ABUT) { cl.kind = CL_KEEP; cl.itpr = cl.ilen = cl.dtl = cl.skip = cl.janti = cl.mismatch = 0; }      // dist == 0, one strand (pre-check)
            else cl = closure_search(g, p, S, sv, P, prev.ref_id, prev.left + rc_ref_span(prev.c, prev.n), (int)cig_len(plast),
                                     curr.left, (int)cig_len(cfirst), prev.anti == curr.anti);
            if (cl.kind == CL_FAIL) return LJ_NONE;
            P += curr.rlen;
            if (cl.kind == CL_KEEP) {
                const int rc = rchain_add(co, prev);
                if (rc) return rc == 2 ? LJ_PUNT : LJ_NONE;
                prev = curr;
                continue;
            }
            if (ABUT) continue;                 // (never reached: every pair is kept)
            if (prev.n + 1 + curr.n > LEAN_C) return LJ_PUNT;
            int anti_closure = psp ? prev.asplice : curr.asplice;
            int nn = prev.n, first;
            uint32_t flen;
            if (cl.kind == CL_INS) {
                const uint32_t bl = (cig_len(plast) - (uint32_t)cl.itpr) & 0x0FFFFFFFu;
                if (bl == 0) --nn; else prev.c.set(nn - 1, cig(cig_op(plast), bl));
                prev.c.set(nn++, cig(OP_INS, (uint32_t)cl.ilen));
                flen = (cig_len(cfirst) + (uint32_t)(cl.itpr - cl.ilen)) & 0x0FFFFFFFu;
                first = flen > 0 ? 0 : 1;
            } else {
                const int nlb = (int)cig_len(plast) + cl.dtl, nrf = (int)cig_len(cfirst) - cl.dtl;
                if (nlb > 0) prev.c.set(nn - 1, cig(cig_op(plast), (uint32_t)nlb)); else --nn;
                if ((uint32_t)cl.skip <= (uint32_t)p.max_deletion_length) prev.c.set(nn++, cig(OP_DEL, (uint32_t)cl.skip));
                else { prev.c.set(nn++, cig(OP_REF_SKIP, (uint32_t)cl.skip)); anti_closure = cl.janti; }
                flen = (uint32_t)nrf;
                first = nrf > 0 ? 0 : 1;
            }
#pragma unroll
            for (int b = 0; b < 5; ++b)
                if (b >= first && b < curr.n) prev.c.set(nn++, b == 0 ? cig(cig_op(cfirst), flen) : curr.c.v[b]);
            const int mismatches = prev.mm + curr.mm + cl.mismatch;                                          // :1822-1870
            prev.n = nn; prev.asplice = anti_closure;
            prev.mm = mismatches & 0xFF;
            prev.ed = (mismatches + rc_gap_len(prev.c, nn)) & 0xFF;
            prev.rlen += curr.rlen;
        }
        {
            const int rc = rchain_add(co, prev);
            if (rc) return rc == 2 ? LJ_PUNT : LJ_NONE;
        }
        if (THJ_EXPF(1024)) return LJ_NONE;
        res.c = co.c; res.n = co.n;
        res.ref_id = h0.ref_id; res.left = left0; res.anti = anti ? 1 : 0; res.asplice = co.saw_as ? 1 : 0;
        res.mm = co.num_mm & 0xFF; res.ed = (co.num_mm + rc_gap_len(co.c, co.n)) & 0xFF;
        res.rlen = rl; res.valid = 1;
        if (rc_read_span(res.c, res.n) != old_read_length) return LJ_NONE;
    }
    return LJ_OK;
}

// lean_finish: filters of JoinSegmentsWorker (:2810-2813), check_editdist_consistency, bowtie_sam_extra, the record.
// `order` = rank of the record among the read's output records, advanced when one is emitted.
// the filters of a joined hit and its tags: false = the hit is not reported
THJ_HD bool lean_finish_check(const Genome& g, const Params& p, const RAln& res, int nsegs, const u64* rp, int W, int rl, const uint8_t* qual, Extras& e) {
    if (!valid_hit(p, res)) return false;
    if (THJ_EXPF(16384)) return false;
    int gapl = (res.ed - res.mm) & 0xFF;
    if (res.mm > p.read_mismatches || gapl > p.read_gap_length || res.ed > p.read_edit_dist) return false;
    SeqView sv = res.anti ? seq_revcomp(rp, W, rl) : seq_forward(rp, W, rl);
    bool qrev;
    if (nsegs == 1) qrev = res.anti;
    else qrev = res.anti ? !read_is_own_revcomp(rp, W, rl) : false;
    sam_extra(g, p, res, sv, qual, rl, qrev, e);
    // check_editdist_consistency (done inside merge_chain in the reference) shares sam_extra's counts
    if (nsegs > 1 && !(e.XM == res.mm || e.XM + e.both_n == res.mm)) return false;
    return true;
}
template <class Sink>
THJ_HD int lean_finish(const Genome& g, const Params& p, const RAln& res, int nsegs, const u64* rp, int W, int rl,
                       const uint8_t* qual, uint32_t read_idx, int& order, Sink& sink) {
    Extras e;
    if (!lean_finish_check(g, p, res, nsegs, rp, W, rl, qual, e)) return SPAN_OK;
    emit_aln(sink, read_idx, order, res, e);
    ++order;
    return SPAN_OK;
}

// Tier 1.  `heads`: room for the heads of the read's nseg hits (LDS in the kernel).  The hits of a one-hit-per-segment
// read are consecutive records: their heads are fetched once, back to back, and every later step reads the staged
// copy instead of paying another HBM round trip (fetching the full 32-byte records was measured at 0.23 ms of a
// 0.81 ms launch; only a segment hit that is itself spliced has a tail to fetch).
template <int MS = SPAN_MAXSEG, class Sink>
THJ_HD int span_read_lean(const Genome& g, const Params& p, const SpanSets& S, const SpanHit* ghits, const uint32_t* so, int nseg,
                          const u64* rp, int W, int rl, const uint8_t* qual, uint32_t read_idx, SpanHitHead* heads, Sink& sink,
                          const SpanHitHead* gheads = nullptr) {
    uint32_t sof[MS + 1];
#pragma unroll
    for (int s = 0; s <= MS; ++s) sof[s] = s <= nseg ? so[s <= nseg ? s : 0] : 0u;
    if (sof[1] == sof[0]) return SPAN_OK;
    int nsegs = 0;
    {
        bool open = true;
#pragma unroll
        for (int s = 0; s < MS; ++s) { open = open && s < nseg && sof[s + 1] > sof[s]; nsegs += open ? 1 : 0; }
    }
    bool single = true;
#pragma unroll
    for (int s = 0; s < MS; ++s) single = single && (s >= nsegs || sof[s + 1] - sof[s] == 1u);
    if (!single) {
        uint32_t last_so = sof[0];
#pragma unroll
        for (int s = 1; s < MS; ++s) last_so = (s == nsegs - 1) ? sof[s] : last_so;
        if (!(ghits[last_so].meta & SH_END)) return SPAN_OK;
        return SPAN_NEED_GENERIC;
    }
    if (THJ_EXPF(65536)) return SPAN_OK;
    {
        Q16 tmp[MS];                            // the heads of the read's (consecutive) hits, all in flight together
#pragma unroll
        for (int k = 0; k < MS; ++k) tmp[k] = load_head(ghits, gheads, (u64)sof[0] + (u64)(k < nsegs ? k : 0));      // unconditional: see span_read_contig_pre
#pragma unroll
        for (int k = 0; k < MS; ++k) if (k < nsegs) ((Q16*)heads)[k] = tmp[k];
    }
    const StagedHits hits{heads, ghits + sof[0], 0xFEDCBA9876543210ull};      // segment s's hit is staged hit s
    if (!(heads[nsegs - 1].meta & SH_END)) return SPAN_OK;
    if (THJ_EXPF(256)) return SPAN_OK;
    RAln res;
    const int jr = lean_join(g, p, S, hits, nsegs, rp, W, rl, res);
    if (jr == LJ_PUNT) return SPAN_NEED_GENERIC;
    if (jr == LJ_INCOMPAT) return SPAN_INCOMPAT;
    if (jr == LJ_NONE) return SPAN_OK;
    int order = 0;
    return lean_finish(g, p, res, nsegs, rp, W, rl, qual, read_idx, order, sink);
}


// ---- multihit reads on the lean machinery -------------------------------------------------------------------
// dfs_seg_hits (long_spanning_reads.cpp:2222-2610) over one hit per segment as in the generic span_read, but every
// complete chain is joined with lean_join on register cigars and the joined hits are compact RAln records: by the
// packed tier (span_pack_wave: a chain per lane) and by the generic tier's first attempt (span_read_multi: a read per
// thread).  Reads that need more cigar ops (LJ_PUNT) or more joined hits than those hold are redone by span_read.
static constexpr int MULTI_MAXJOIN = 16;
THJ_HD bool raln_less(const RAln& a, const RAln& b) {          // BowtieHit::operator< (bwt_map.h:180-207)
    if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
    if (a.left != b.left) return a.left < b.left;
    if (a.anti != b.anti) return a.anti < b.anti;
    if (a.mm != b.mm) return a.mm < b.mm;
    if (a.ed != b.ed) return a.ed < b.ed;
    if (a.n != b.n) return a.n < b.n;
#pragma unroll
    for (int i = 0; i < LEAN_C; ++i)
        if (i < a.n && a.c.v[i] != b.c.v[i]) {
            const int oa = cig_op(a.c.v[i]), ob = cig_op(b.c.v[i]);
            return oa < ob || (oa == ob && cig_len(a.c.v[i]) < cig_len(b.c.v[i]));
        }
    return false;
}
THJ_HD bool raln_eq(const RAln& a, const RAln& b) {            // BowtieHit::operator== (bwt_map.h:167-178)
    if (a.ref_id != b.ref_id || a.anti != b.anti || a.left != b.left || a.asplice != b.asplice || a.ed != b.ed || a.n != b.n) return false;
    bool same = true;
#pragma unroll
    for (int i = 0; i < LEAN_C; ++i) same = same && (i >= a.n || a.c.v[i] == b.c.v[i]);
    return same;
}

// sort + unique of the joined hits (:2805-2807), then the per-hit filters and the records
template <class Sink>
THJ_HD int multi_finish(const Genome& g, const Params& p, RAln* joined, int nj, int nsegs, const u64* rp, int W, int rl,
                        const uint8_t* qual, uint32_t read_idx, Sink& sink) {
    if (THJ_EXPF(4096)) return SPAN_OK;
    for (int i = 1; i < nj; ++i) {               // insertion sort (stable; what std::sort does below 16 elements)
        RAln t = joined[i]; int k = i;
        while (k > 0 && raln_less(t, joined[k - 1])) { joined[k] = joined[k - 1]; --k; }
        joined[k] = t;
    }
    int w = 0;
    for (int i = 0; i < nj; ++i) if (w == 0 || !raln_eq(joined[w - 1], joined[i])) joined[w++] = joined[i];
    nj = w;
    int order = 0, status = SPAN_OK;
    for (int i = 0; i < nj; ++i) {
        const int st = lean_finish(g, p, joined[i], nsegs, rp, W, rl, qual, read_idx, order, sink);
        if (st != SPAN_OK) status = st;
    }
    return status;
}

// small arrays in registers: every access an unrolled compare / select chain
template <int N> THJ_HD int rsel_get(const int (&a)[N], int i) {
    int r = 0;
#pragma unroll
    for (int k = 0; k < N; ++k) r = (i == k) ? a[k] : r;
    return r;
}
template <int N> THJ_HD void rsel_set(int (&a)[N], int i, int x) {
#pragma unroll
    for (int k = 0; k < N; ++k) a[k] = (i == k) ? x : a[k];
}

// MAXJ: joined alignments a read may have here.  Tier 3 gives every copy of a 40-copy repeat its alignment (max_seg_multihits = 40):
// with room for 16 such a read ran the whole search, gave up at the 17th and ran it again through span_read's general arrays.
template <int MAXJ = MULTI_MAXJOIN, class Sink>
THJ_HD int span_read_multi(const Genome& g, const Params& p, const SpanSets& S, const SpanHit* ghits, const uint32_t* so, int nseg,
                           const u64* rp, int W, int rl, const uint8_t* qual, uint32_t read_idx, SpanHit* stage, Sink& sink) {
    if (so[1] == so[0]) return SPAN_OK;                         // worker iterates over first-segment groups
    int nsegs = 0;
    while (nsegs < nseg && so[nsegs + 1] > so[nsegs]) ++nsegs;   // look_right stops at the first empty segment (:151)
    if (nsegs > SPAN_MAXSEG) nsegs = SPAN_MAXSEG;
    if (!(ghits[so[nsegs - 1]].meta & SH_END)) return SPAN_OK;   // :2777-2785
    if (p.bowtie2)
        for (int s = 0; s < nsegs; ++s)
            if ((int)(so[s + 1] - so[s]) > p.max_seg_multihits) return SPAN_OK;   // :2625-2632
    if (THJ_EXPF(2048)) return SPAN_OK;
    const int L = p.segment_length;
    RAln joined[MAXJ]; int nj = 0;
    uint32_t idx[SPAN_MAXSEG];               // next candidate of each depth
    int pleft[SPAN_MAXSEG], pright[SPAN_MAXSEG];   // left / right of the hit chosen at each depth
    for (uint32_t i0 = so[0]; i0 < so[1]; ++i0) {                // :2634-2664
        stage[0] = ghits[i0];
        const uint32_t ref0 = stage[0].ref_id;
        const bool anti0 = (stage[0].meta & SH_ANTI) != 0;
        {
            const RAln a0 = raln_from_hit(stage[0], 0, L, rl);
            pleft[0] = a0.left; pright[0] = a0.left + rc_ref_span(a0.c, a0.n);
        }
        int num_try = 10000;
        int depth = 1;
        idx[1] = so[1];
        while (depth >= 1) {
            if (num_try <= 0) break;
            if (depth == nsegs) {
                --num_try;
                RAln res;
                const int jr = lean_join(g, p, S, stage, nsegs, rp, W, rl, res);
                if (jr == LJ_PUNT) return SPAN_NEED_GENERIC;
                if (jr == LJ_OK && valid_hit(p, res)) {
                    if (nj >= MAXJ) return SPAN_NEED_GENERIC;
                    joined[nj++] = res;
                }
                --depth;
                continue;
            }
            if (idx[depth] >= so[depth + 1]) { --depth; continue; }
            // the next candidates four at a time (their heads: contig, left, flags, first cigar op): a read of a 40-copy repeat walks
            // 40 candidates per parent and level and takes one -- one dependent HBM round trip each when they are fetched singly
            const uint32_t i0c = idx[depth];
            const uint32_t nb = so[depth + 1] - i0c < 4u ? so[depth + 1] - i0c : 4u;
            Q16 hd[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) if ((uint32_t)k < nb) hd[k] = *(const Q16*)(ghits + i0c + k);
            bool took = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (took || (uint32_t)k >= nb) continue;
                const bool canti = (hd[k].z & SH_ANTI) != 0;
                if (ref0 != hd[k].x || anti0 != canti) continue;              // every hit of a chain shares contig and strand
                SpanHit sh;
                sh.ref_id = hd[k].x; sh.left = (int32_t)hd[k].y; sh.meta = hd[k].z; sh.cigar[0] = hd[k].w;
                sh.cigar[1] = sh.cigar[2] = sh.cigar[3] = sh.cigar[4] = 0;
                if ((hd[k].z >> 24) > 1) sh = ghits[i0c + k];
                const RAln cand = raln_from_hit(sh, depth, L, rl);
                const int cright = cand.left + rc_ref_span(cand.c, cand.n);
                const int dist = anti0 ? pleft[depth - 1] - cright : cand.left - pright[depth - 1];   // :2352-2378, :2531-2556
                if (dist <= p.max_report_intron && dist >= -p.max_insertion_length) {
                    took = true;
                    idx[depth] = i0c + (uint32_t)k + 1;
                    stage[depth] = sh;
                    pleft[depth] = cand.left; pright[depth] = cright;
                    ++depth;
                    if (depth < nsegs) idx[depth] = so[depth];
                }
            }
            if (!took) idx[depth] = i0c + nb;
        }
    }
    return multi_finish(g, p, joined, nj, nsegs, rp, W, rl, qual, read_idx, sink);
}

struct StagedHits8 {        // as StagedHits, eight bits per segment
    const SpanHitHead* heads; const SpanHit* g0; u64 sel;
    THJ_HD SpanHit operator[](int s) const { return staged_hit(heads, g0, (int)((sel >> (8 * s)) & 255)); }
};

// ---- tier 2, packed: multihit reads as CHAINS spread over the lanes of a wave ----------------------------------------------
// dfs_seg_hits (long_spanning_reads.cpp:2222-2610) only chains hits of one contig and strand that lie within
// [-max_insertion_length, max_report_intron] of each other (:2352-2378, :2531-2556), and its num_try budget is per first-segment
// hit and spent at leaves only (:2596, :2634-2664): the searches from different first-segment hits ("roots") never share state.
// A read whose segments map to c copies of a repeat is therefore c independent one-hit-per-segment chains, and a chain is what tier 1
// joins on registers.  A wave takes 64 entries of the multihit list at a time:
//   phase 0  lane = list entry: segment offsets, the worker's early outs (:2777-2785, :2625-2632), root / hit counts, prefix sums;
//   rounds   consecutive entries with at most MAXROOTS roots and MAXHITS hits in all: the 16-byte heads of all their hits are staged in
//            LDS (plus each hit's right end), lane = root walks dfs_seg_hits over the staged hits -- comparisons only -- first to count
//            its complete chains, then (after a prefix sum) to write them to the round's chain list;
//   sub-rounds of up to 64 chains cut at read boundaries: lane = chain runs lean_join (merge_chain on register cigars) and
//            valid_hit; the joined hits of a read sit in adjacent lanes and are ranked by BowtieHit::operator< with ties in generation
//            order (a stable sort, what multi_finish's insertion sort yields), adjacent duplicates are dropped (:2805-2807), filters
//            and tags run on every kept hit's own lane, record ranks come from a ballot over the sorted positions.
// What it does not take goes on to the generic tier, whole (nothing of such a read is emitted here): more than MAXROOTS roots or
// MAXHITS hits, more than 64 chains, a chain list that is full, a join that needs more than LEAN_C cigar ops.
// X: wave operations (lane, ballot, bcast, incl_scan, shfl, wmax, wsync), all called in wave-uniform control flow.
static constexpr int PACK_ENTRIES = 64, PACK_MAXCHAINS = 64;
template <int MS, int MAXHITS, int CL>
struct PackLds {
    SpanHitHead heads[MAXHITS];                 // the round's hits, read after read
    int32_t hright[MAXHITS];                    // their right ends
    uint8_t hslot[MAXHITS];                     // their entries
    uint8_t hnext[MAXHITS], hncomp[MAXHITS];    // the first hit of the next segment a hit chains with (number among the read's hits), and how many do
    uint32_t rexcl[PACK_ENTRIES + 1], hexcl[PACK_ENTRIES + 1];     // roots / hits of the (active) entries before entry i
    uint32_t so0[PACK_ENTRIES], rd[PACK_ENTRIES];                  // the entry's first hit in the batch, its read
    uint16_t segoff[PACK_ENTRIES][MS + 1];      // segment offsets relative to the read's first hit
    uint16_t cbeg[PACK_ENTRIES], cend[PACK_ENTRIES];               // the entry's chains in the round's chain list
    uint8_t nsegs[PACK_ENTRIES], punt[PACK_ENTRIES];
    u64 csel[CL];                               // a chain: eight bits per segment, the hit's number among the read's hits
    uint8_t cslot[CL];                          // ... and its entry
    uint8_t perm[64];
};
THJ_HD int pack_slot(const uint32_t* excl, int lo, int hi, uint32_t v) {       // last i in [lo, hi) with excl[i] <= v (excl[lo] <= v)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (excl[mid] <= v) lo = mid; else hi = mid; }
    return lo;
}
THJ_HD u64 pack_range_mask(int b, int e) { return lowmask(e) & ~lowmask(b); }  // bits [b, e), 0 <= b <= e <= 64
// right end of a hit from its head (one-op hits: everything but spliced segment hits) or its record
THJ_HD int pack_hit_right(const Q16& q, const SpanHit* ghits, u64 i) {
    int n = (int)(q.z >> 24); if (n > 5) n = 5;
    int r = (int)q.y;
    if (n <= 1) {
        const int op = cig_op(q.w);
        if (n == 1 && (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL)) r += (int)cig_len(q.w);
        return r;
    }
    const SpanHit sh = ghits[i];
    for (int k = 0; k < n; ++k) {
        const int op = cig_op(sh.cigar[k]);
        if (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL) r += (int)cig_len(sh.cigar[k]);
    }
    return r;
}
// dfs_seg_hits from root i0 over the staged hits of one read (hd / hr / off relative to its first hit): the complete chains, counted
// and -- when `store` -- written to csel / cslot from position pos0 on (below pos_lim).  More than `cap` chains: cap + 1.
template <int MS>
THJ_HD int pack_dfs(const SpanHitHead* hd, const int32_t* hr, const uint16_t* off, int nsegs, int i0, const Params& p, int cap,
                    bool store, bool store_sel, u64* csel, uint8_t* cslot, int pos0, int pos_lim, int slot) {
    int idx[MS], pleft[MS], pright[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) idx[s] = pleft[s] = pright[s] = 0;
    const SpanHitHead h0 = hd[i0];
    const uint32_t ref0 = h0.ref_id;
    const bool anti0 = (h0.meta & SH_ANTI) != 0;
    pleft[0] = h0.left; pright[0] = hr[i0];
    u64 sel = (u64)i0;
    int n = 0, depth = 1;
    idx[1 < MS ? 1 : 0] = (int)off[1];
    while (depth >= 1) {
        if (depth == nsegs) {
            if (n >= cap) return cap + 1;
            if (store) {
                const int pos = pos0 + n;
                if (pos < pos_lim) { cslot[pos] = (uint8_t)slot; if (store_sel) csel[pos] = sel; }
            }
            ++n; --depth;
            continue;
        }
        const int cur = rsel_get(idx, depth);
        if (cur >= (int)off[depth + 1]) { --depth; continue; }
        rsel_set(idx, depth, cur + 1);
        const SpanHitHead c = hd[cur];
        if (c.ref_id != ref0 || ((c.meta & SH_ANTI) != 0) != anti0) continue;      // every hit of a chain shares contig and strand
        const int cright = hr[cur];
        const int dist = anti0 ? rsel_get(pleft, depth - 1) - cright : c.left - rsel_get(pright, depth - 1);   // :2352-2378, :2531-2556
        if (dist > p.max_report_intron || dist < -p.max_insertion_length) continue;
        sel = (sel & ~(255ull << (8 * depth))) | ((u64)cur << (8 * depth));
        rsel_set(pleft, depth, c.left); rsel_set(pright, depth, cright);
        ++depth;
        if (depth < nsegs) rsel_set(idx, depth, (int)off[depth]);
    }
    return n;
}
template <class X> THJ_HD void pack_shfl_cigar(X& x, const RAln& a, int src, RAln& o) {
    o.anti = (int)x.shfl((uint32_t)a.anti, src); o.asplice = (int)x.shfl((uint32_t)a.asplice, src);
    o.mm = (int)x.shfl((uint32_t)a.mm, src); o.ed = (int)x.shfl((uint32_t)a.ed, src); o.n = (int)x.shfl((uint32_t)a.n, src);
#pragma unroll
    for (int k = 0; k < LEAN_C; ++k) o.c.v[k] = x.shfl(a.c.v[k], src);
}
// One batch of list entries: lane i has entry `my_read` (has_entry).  Returns whether the lane's entry goes on to the generic tier.
template <int MS, int MAXROOTS, int MAXHITS, int CL, class X, class Sink>
THJ_HD bool span_pack_wave(X& x, const Genome& g, const Params& p, const SpanSets& S, const SpanHit* ghits, const SpanHitHead* gheads,
                           const uint32_t* seg_off, int nseg, const u64* planes, int W, const uint16_t* read_len, const uint8_t* quals,
                           int qual_stride, uint32_t my_read, bool has_entry, PackLds<MS, MAXHITS, CL>& L, Sink& sink,
                           unsigned long long* tm = nullptr) {
    // tm (developer timing, THJ_PACK_TIMING): ticks of x.clock() spent in [0] phase 0, [1] staging, [2] the two searches and their prefix
    // sums, [3] joins, [4] rank + unique, [5] filters + tags + records; [6] rounds, [7] sub-rounds
#define PK_MARK(k) do { if (tm) { const unsigned long long now_ = x.clock(); tm[k] += now_ - t_last; t_last = now_; } } while (0)
    unsigned long long t_last = tm ? x.clock() : 0ull;
    static_assert(MAXROOTS <= 64 && MAXHITS <= 256 && CL >= PACK_MAXCHAINS && CL <= 65535, "pack limits");
    const int lane = x.lane;
    // ---- phase 0: lane = entry
    int status = 0;                             // 0 nothing to do (or no entry), 1 active, 2 generic tier
    uint32_t roots = 0, nh = 0;
    if (has_entry) {
        const uint32_t* so = seg_off + (size_t)my_read * nseg;
        uint32_t sof[MS + 1];
#pragma unroll
        for (int s = 0; s <= MS; ++s) sof[s] = s <= nseg ? so[s <= nseg ? s : 0] : 0u;
        int nsegs = 0;
        {
            bool open = true;
#pragma unroll
            for (int s = 0; s < MS; ++s) { open = open && s < nseg && sof[s + 1] > sof[s]; nsegs += open ? 1 : 0; }
        }
        if (nsegs > 0) {                        // the worker iterates over first-segment groups
            uint32_t last_so = sof[0], end_so = sof[0];
#pragma unroll
            for (int s = 1; s <= MS; ++s) { last_so = (s == nsegs - 1) ? sof[s] : last_so; end_so = (s == nsegs) ? sof[s] : end_so; }
            bool ok = (load_head(ghits, gheads, (u64)last_so).z & SH_END) != 0;          // :2777-2785
            if (ok && p.bowtie2) {
#pragma unroll
                for (int s = 0; s < MS; ++s) ok = ok && !(s < nsegs && (int)(sof[s + 1] - sof[s]) > p.max_seg_multihits);   // :2625-2632
            }
            if (ok) {
                roots = sof[1] - sof[0]; nh = end_so - sof[0];
                // one hit per segment: tier 1 has been there and needs more cigar ops than the registers hold
                status = (roots > (uint32_t)MAXROOTS || nh > (uint32_t)MAXHITS || nh == (uint32_t)nsegs) ? 2 : 1;
            }
        }
        L.so0[lane] = sof[0]; L.rd[lane] = my_read; L.nsegs[lane] = (uint8_t)nsegs;
#pragma unroll
        for (int s = 0; s <= MS; ++s) L.segoff[lane][s] = (uint16_t)(sof[s] - sof[0]);
    }
    L.punt[lane] = 0;
    const uint32_t rc = status == 1 ? roots : 0u, hc = status == 1 ? nh : 0u;
    const uint32_t rincl = x.incl_scan(rc), hincl = x.incl_scan(hc);
    L.rexcl[lane] = rincl - rc; L.hexcl[lane] = hincl - hc;
    if (lane == 63) { L.rexcl[64] = rincl; L.hexcl[64] = hincl; }
    x.wsync();
    PK_MARK(0);
    // ---- rounds
    int cur = 0;
    while (cur < PACK_ENTRIES) {
        const uint32_t rbase = L.rexcl[cur], hbase = L.hexcl[cur];
        const u64 m_over = x.ballot(lane >= cur && (rincl - rbase > (uint32_t)MAXROOTS || hincl - hbase > (uint32_t)MAXHITS));
        const int end = m_over ? ctz(m_over) : PACK_ENTRIES;
        const uint32_t R = L.rexcl[end] - rbase, H = L.hexcl[end] - hbase;
        if (R == 0) { cur = end; continue; }
        // the heads of the round's hits, and each hit's right end
        for (uint32_t j = (uint32_t)lane; j < H; j += 64u) {
            const int slot = pack_slot(L.hexcl, cur, end, hbase + j);
            const u64 gi = (u64)L.so0[slot] + (hbase + j - L.hexcl[slot]);
            const Q16 q = load_head(ghits, gheads, gi);
            ((Q16*)L.heads)[j] = q;
            L.hright[j] = pack_hit_right(q, ghits, gi);
            L.hslot[j] = (uint8_t)slot;
        }
        x.wsync();
        PK_MARK(1);
        if (tm) tm[6] += 1;
        // lane = hit: which hits of the next segment it chains with (:2352-2378, :2531-2556) -- the first one and their number.  A root
        // whose path meets one such hit per segment has that one chain (the case of reads from c copies of a repeat: c roots, c
        // chains); dfs_seg_hits proper (pack_dfs) runs only from roots that meet a choice.
        for (uint32_t j = (uint32_t)lane; j < H; j += 64u) {
            const int slot = (int)L.hslot[j];
            const int hb = (int)(L.hexcl[slot] - hbase), nsg = (int)L.nsegs[slot];
            const int k = (int)j - hb;
            int d = 0;
#pragma unroll
            for (int s = 1; s < MS; ++s) d += (s < nsg && (int)L.segoff[slot][s] <= k) ? 1 : 0;
            int first = 0, nc = 0;
            if (d + 1 < nsg) {
                const SpanHitHead me = L.heads[j];
                const int myright = L.hright[j];
                const bool anti = (me.meta & SH_ANTI) != 0;
                const int c1 = (int)L.segoff[slot][d + 2];
                for (int c = (int)L.segoff[slot][d + 1]; c < c1; ++c) {
                    const SpanHitHead o = L.heads[hb + c];
                    const int dist = anti ? me.left - L.hright[hb + c] : o.left - myright;
                    const bool okc = o.ref_id == me.ref_id && ((o.meta & SH_ANTI) != 0) == anti && dist <= p.max_report_intron && dist >= -p.max_insertion_length;
                    first = (okc && nc == 0) ? c : first;
                    nc += okc ? 1 : 0;
                }
            }
            L.hnext[j] = (uint8_t)first; L.hncomp[j] = (uint8_t)(nc > 255 ? 255 : nc);
        }
        x.wsync();
        // lane = root: count the chains, place them, write them
        const bool is_root = (uint32_t)lane < R;
        int slot = 0, i0 = 0, hb = 0, nsg = 0;
        uint32_t n_roots = 0;
        if (is_root) {
            slot = pack_slot(L.rexcl, cur, end, rbase + (uint32_t)lane);
            i0 = (int)(rbase + (uint32_t)lane - L.rexcl[slot]);
            hb = (int)(L.hexcl[slot] - hbase);
            nsg = (int)L.nsegs[slot];
            n_roots = L.rexcl[slot + 1] - L.rexcl[slot];
        }
        int cnt = 0;
        bool choice = false;                    // the root's search meets a hit with several continuations
        u64 sel1 = 0;                           // its one chain otherwise
        if (is_root) {
            int k = i0;
            sel1 = (u64)i0;
            cnt = 1;
            for (int d = 1; d < nsg; ++d) {
                const int nc = (int)L.hncomp[hb + k];
                if (nc != 1) { cnt = 0; choice = nc > 1; break; }
                k = (int)L.hnext[hb + k];
                sel1 |= (u64)k << (8 * d);
            }
            if (choice) {
                cnt = pack_dfs<MS>(L.heads + hb, L.hright + hb, L.segoff[slot], nsg, i0, p, PACK_MAXCHAINS, false, false, nullptr, nullptr, 0, 0, slot);
                if (cnt > PACK_MAXCHAINS) { L.punt[slot] = 1; cnt = 0; }
            }
        }
        x.wsync();
        if (is_root && L.punt[slot]) cnt = 0;
        const uint32_t cincl = x.incl_scan((uint32_t)cnt), cexcl = cincl - (uint32_t)cnt;
        if (is_root && i0 == 0) L.cbeg[slot] = (uint16_t)cexcl;
        if (is_root && (uint32_t)i0 == n_roots - 1) L.cend[slot] = (uint16_t)cincl;
        x.wsync();
        if (is_root && ((uint32_t)L.cend[slot] - (uint32_t)L.cbeg[slot] > (uint32_t)PACK_MAXCHAINS || (uint32_t)L.cend[slot] > (uint32_t)CL)) L.punt[slot] = 1;
        x.wsync();
        const uint32_t C = x.bcast(cincl, 63), Cend = C < (uint32_t)CL ? C : (uint32_t)CL;
        if (is_root && cnt > 0) {
            if (choice) pack_dfs<MS>(L.heads + hb, L.hright + hb, L.segoff[slot], nsg, i0, p, PACK_MAXCHAINS, true, !L.punt[slot], L.csel, L.cslot, (int)cexcl, CL, slot);
            else if (cexcl < (uint32_t)CL) { L.cslot[cexcl] = (uint8_t)slot; L.csel[cexcl] = sel1; }
        }
        x.wsync();
        PK_MARK(2);
        // ---- sub-rounds: lane = chain
        uint32_t c0 = 0;
        while (c0 < Cend) {
            const uint32_t j = c0 + (uint32_t)lane;
            const bool has = j < Cend;
            const int cs = has ? (int)L.cslot[j] : 0;
            const uint32_t cb = has ? (uint32_t)L.cbeg[cs] : 0u, ce = has ? (uint32_t)L.cend[cs] : 0u;
            const bool cut = has && ce > c0 + 64u;
            const u64 m_cut = x.ballot(cut);
            uint32_t next = c0 + 64u;
            if (m_cut) {
                const int fs = (int)L.cslot[c0 + (uint32_t)ctz(m_cut)];
                next = (uint32_t)L.cbeg[fs];
                if (next == c0) next = (uint32_t)L.cend[fs];          // a read with more chains than a sub-round holds (it has punted)
            }
            bool live = has && !cut && !L.punt[cs];
            RAln res;
            res.ref_id = 0; res.left = 0; res.n = 0; res.anti = res.asplice = res.mm = res.ed = res.rlen = res.valid = 0;
#pragma unroll
            for (int k = 0; k < LEAN_C; ++k) res.c.v[k] = 0;
            bool okj = false;
            uint32_t r = 0; int rl = 0, nsj = 0;
            const u64* rp = planes;
            if (live) {
                r = L.rd[cs]; rl = (int)read_len[r]; nsj = (int)L.nsegs[cs];
                rp = planes + (size_t)r * 3 * W;
                const StagedHits8 ch{L.heads + (L.hexcl[cs] - hbase), ghits + L.so0[cs], L.csel[j]};
                const int jr = lean_join(g, p, S, ch, nsj, rp, W, rl, res);
                if (jr == LJ_PUNT) L.punt[cs] = 1;
                else okj = jr == LJ_OK && valid_hit(p, res);
            }
            x.wsync();
            PK_MARK(3);
            if (tm) tm[7] += 1;
            live = live && !L.punt[cs];
            const bool val = live && okj;
            const int gb = live ? (int)(cb - c0) : lane, ge = live ? (int)(ce - c0) : lane;
            // rank among the read's joined hits: BowtieHit::operator<, ties in generation order
            const uint32_t maxlen = x.wmax((uint32_t)(ge - gb));
            int rank = 0;
            for (uint32_t k = 0; k < maxlen; ++k) {
                const bool in = (int)k < ge - gb;
                const int src = in ? gb + (int)k : lane;
                const uint32_t oref = x.shfl(res.ref_id, src), oleft = x.shfl((uint32_t)res.left, src), oval = x.shfl(val ? 1u : 0u, src);
                const bool cmp = in && val && oval != 0u && src != lane;
                const bool tie = cmp && oref == res.ref_id && (int32_t)oleft == res.left;
                if (x.ballot(tie)) {
                    RAln o;
                    pack_shfl_cigar(x, res, src, o);
                    o.ref_id = oref; o.left = (int32_t)oleft;
                    if (tie) rank += (raln_less(o, res) || (src < lane && !raln_less(res, o))) ? 1 : 0;
                }
                if (cmp && !tie) rank += (oref < res.ref_id || (oref == res.ref_id && (int32_t)oleft < res.left)) ? 1 : 0;
            }
            const u64 gmask = pack_range_mask(gb, ge);
            const int nval = popc(x.ballot(val) & gmask);
            if (val) L.perm[gb + rank] = (uint8_t)lane;
            x.wsync();
            // unique: equal to the hit before it in sorted order
            const int pl = (val && rank > 0) ? (int)L.perm[gb + rank - 1] : lane;
            const uint32_t pref = x.shfl(res.ref_id, pl), pleft = x.shfl((uint32_t)res.left, pl);
            const bool maybe = val && rank > 0 && pref == res.ref_id && (int32_t)pleft == res.left;
            bool dup = false;
            if (x.ballot(maybe)) {
                RAln o;
                pack_shfl_cigar(x, res, pl, o);
                o.ref_id = pref; o.left = (int32_t)pleft;
                if (maybe) dup = raln_eq(o, res);
            }
            PK_MARK(4);
            Extras e;
            bool emit = false;
            if (val && !dup) emit = lean_finish_check(g, p, res, nsj, rp, W, rl, quals + (size_t)r * qual_stride, e);
            // which sorted positions emit: the lane at position q asks the lane that holds the q-th hit
            const bool hosts = live && lane - gb < nval;
            const uint32_t es = x.shfl(emit ? 1u : 0u, hosts ? (int)L.perm[lane] : lane);
            const u64 smask = x.ballot(hosts && es != 0u);
            if (emit) emit_aln(sink, r, popc(smask & pack_range_mask(gb, gb + rank)), res, e);
            if (live && lane == gb) sink.set_count(r, popc(smask & gmask));
            x.wsync();
            PK_MARK(5);
            c0 = next;
        }
        cur = end;
    }
#undef PK_MARK
    return status == 2 || (status == 1 && L.punt[lane] != 0);
}

// ---- chains: tier 0 -> thj_k_join -> thj_k_finish ------------------------------------------------------------------------------
// The reference separates JOINING a chain of segment hits (merge_chain, long_spanning_reads.cpp:805-2038) from FINISHING the joined
// hit (check_editdist_consistency + bowtie_sam_extra + print_bamhit, bwt_map.cpp:2349-2648, :1888-2093).  So do the kernels: a
// read of at most four segments with one hit per segment that tier 0 does not finish -- a spliced or indel read: one of its
// segment hits is spliced (the junction-db mapping's aM gN bM record) or two of them do not abut -- is handed on as a 32-byte
// CHAIN ENTRY that names its hits; the join fetches those records (all in flight together, no trip through the CSR row), runs
// merge_chain on register cigars and writes a compact JOINED HIT (contig, left, <= 8 cigar ops, mismatches); the finish kernel
// runs over the dense list of joined hits.  One chain of a multihit read travels the same way, with its rank among the read's
// chains (q of k, ordered by contig and left: the joined hits' BowtieHit::operator< order when those differ).  Reads of more
// segments, and every read under --fusion-search, take span_read_lean.
struct alignas(16) ChainEntry { uint32_t read, meta, spare0, spare1; uint32_t hit[4]; };     // hit[s]: segment s's hit, its number in the batch
// meta: [0..2] segments (1..4) | [4..6] q | [7..9] k - 1 | [10..19] read length
static constexpr int CHAIN_MAXSEG = 4;
THJ_HD uint32_t chain_meta(int nsegs, int q, int k, int rl) { return (uint32_t)nsegs | ((uint32_t)q << 4) | ((uint32_t)(k - 1) << 7) | ((uint32_t)rl << 10); }
THJ_HD int chain_nsegs(uint32_t m) { return (int)(m & 7u); }
THJ_HD int chain_q(uint32_t m) { return (int)((m >> 4) & 7u); }
THJ_HD int chain_k(uint32_t m) { return (int)((m >> 7) & 7u) + 1; }
THJ_HD int chain_rl(uint32_t m) { return (int)((m >> 10) & 1023u); }
// A joined hit on its way to the finish kernel: two 16-byte words (+ one for cigar ops 4..7).
// a.x = read (JOINED_PAD: no entry there), a.y = contig, a.z = left, a.w = meta; b = cigar ops 0..3; c = ops 4..7 (n > 4 only)
// meta: [0..3] n (0: nothing joined) | [4] antisense | [5] antisense splice | [6] the read has one segment | [7..9] q | [10..12] k - 1 | [13..20] mismatches | [21..28] edit distance
static constexpr uint32_t JOINED_PAD = 0xFFFFFFFFu;
THJ_HD uint32_t joined_none_meta(int q, int k) { return ((uint32_t)q << 7) | ((uint32_t)(k - 1) << 10); }
THJ_HD int joined_n(uint32_t m) { return (int)(m & 15u); }
THJ_HD int joined_q(uint32_t m) { return (int)((m >> 7) & 7u); }
THJ_HD int joined_k(uint32_t m) { return (int)((m >> 10) & 7u) + 1; }
THJ_HD uint32_t joined_meta(const RAln& r, bool one_seg, int q, int k) {
    return (uint32_t)r.n | (r.anti ? 16u : 0u) | (r.asplice ? 32u : 0u) | (one_seg ? 64u : 0u) | ((uint32_t)q << 7) | ((uint32_t)(k - 1) << 10) |
           ((uint32_t)(r.mm & 0xFF) << 13) | ((uint32_t)(r.ed & 0xFF) << 21);
}
THJ_HD void joined_unpack(const Q16& a, const Q16& b, const Q16& c, RAln& r, bool& one_seg, int& q, int& k) {
    r.ref_id = a.y; r.left = (int32_t)a.z;
    const uint32_t m = a.w;
    r.n = (int)(m & 15u); r.anti = (m & 16u) ? 1 : 0; r.asplice = (m & 32u) ? 1 : 0; one_seg = (m & 64u) != 0;
    q = (int)((m >> 7) & 7u); k = (int)((m >> 10) & 7u) + 1;
    r.mm = (int)((m >> 13) & 0xFFu); r.ed = (int)((m >> 21) & 0xFFu);
    r.c.v[0] = b.x; r.c.v[1] = b.y; r.c.v[2] = b.z; r.c.v[3] = b.w;
    const bool more = r.n > 4;
    r.c.v[4] = more ? c.x : 0u; r.c.v[5] = more ? c.y : 0u; r.c.v[6] = more ? c.z : 0u; r.c.v[7] = more ? c.w : 0u;
    r.rlen = 0; r.valid = 1;
}
// the join of one chain: merge_chain on register cigars (lean_join), valid_hit and the filters of JoinSegmentsWorker (:2810-2813) --
// everything that decides on the cigar alone.  LJ_OK: `res` goes on to the finish kernel.  `hits[s]`: the chain's hit of segment s.
template <bool ABUT = false, class Hits>
THJ_HD int chain_join(const Genome& g, const Params& p, const SpanSets& S, const Hits& hits, uint32_t meta, const u64* rp, int W, RAln& res) {
    const int jr = lean_join<ABUT>(g, p, S, hits, chain_nsegs(meta), rp, W, chain_rl(meta), res);
    if (jr != LJ_OK) return jr;
    if (!valid_hit(p, res)) return LJ_NONE;
    const int gapl = (res.ed - res.mm) & 0xFF;
    if (res.mm > p.read_mismatches || gapl > p.read_gap_length || res.ed > p.read_edit_dist) return LJ_NONE;
    return LJ_OK;
}

// ---- the chains of a multihit read without a sort (thj_k_chains) -------------------------------------------------------------
// dfs_seg_hits (long_spanning_reads.cpp:2222-2610) chains hits of one contig and strand that lie within [-max_insertion_length,
// max_report_intron] of each other, from every first-segment hit on its own (num_try is per first-segment hit, :2634-2664).  When no
// hit of the read has more than ONE such successor in the next segment, every first-segment hit starts at most one chain, and a read
// whose segments map to c copies of a repeat -- the bulk of the multihit reads -- is c independent chains.  JoinSegmentsWorker then
// sorts the joined hits by BowtieHit::operator< (bwt_map.h:180-207: contig, left, strand, then mismatches ... cigar), drops equal
// neighbours and numbers what passes the filters (:2805-2813).  A joined hit's contig, left and strand are those of its chain's
// leftmost hit, known before the join: when the chains of a read differ in (contig, left, strand), their order is known up front (and
// no two are equal), so each can travel as a chain entry with its rank q among the read's k chains, be joined and finished on a
// lane of its own, and the finish kernel only has to count the lower-ranked siblings that were reported.  Anything else -- a hit
// with two successors, more than CHAINS_MAXHITS hits or CHAINS_MAX chains, two chains with one (contig, left, strand) -- is declined
// and stays with thj_k_stitch_pack.
static constexpr int CHAINS_MAXHITS = 16, CHAINS_MAX = 8;
enum { CHAINS_DECLINE = -1 };
// Tab: the read's hits, numbered from its first (segment s: numbers off[s] .. off[s + 1]): head(j) = {ref_id, left, meta, right end}
// Out: sel[c] = the hit numbers of chain c, four bits per segment; q[c] = its rank.  Returns the number of chains (0: the read has none).
template <class Tab>
THJ_HD int chains_discover(const Params& p, const Tab& tab, const uint32_t* off /* [nsegs + 1], relative */, int nsegs, uint32_t (&sel)[CHAINS_MAX], int (&q)[CHAINS_MAX]) {
    const int nh = (int)off[nsegs], roots = (int)off[1];
    if (nh > CHAINS_MAXHITS || roots > CHAINS_MAX || nsegs > CHAIN_MAXSEG) return CHAINS_DECLINE;
    // every hit's successor in the next segment (four bits each), and whether it has one
    u64 succ = 0; uint32_t has = 0, gap = 0;        // gap: the hit does not abut its successor (the join will search a closure there: one more cigar op)
    for (int s = 0; s + 1 < nsegs; ++s)
        for (int j = (int)off[s]; j < (int)off[s + 1]; ++j) {
            const SpanHitHead me = tab.head(j);
            const bool anti = (me.meta & SH_ANTI) != 0;
            int first = 0, nc = 0; bool fgap = false;
            for (int c = (int)off[s + 1]; c < (int)off[s + 2]; ++c) {
                const SpanHitHead o = tab.head(c);
                const int dist = anti ? me.left - (int32_t)o.cigar0 : o.left - (int32_t)me.cigar0;      // :2352-2378, :2531-2556 (cigar0 holds the right end)
                const bool okc = o.ref_id == me.ref_id && ((o.meta & SH_ANTI) != 0) == anti && dist <= p.max_report_intron && dist >= -p.max_insertion_length;
                fgap = (okc && nc == 0) ? dist != 0 : fgap;
                first = (okc && nc == 0) ? c : first;
                nc += okc ? 1 : 0;
            }
            if (nc > 1) return CHAINS_DECLINE;
            succ |= (u64)first << (4 * j);
            has |= nc ? 1u << j : 0u;
            gap |= fgap ? 1u << j : 0u;
        }
    int k = 0;
    for (int i = 0; i < roots; ++i) {
        int j = i; uint32_t sl = (uint32_t)i; bool ok = true;
        int ops = (int)(tab.head(i).meta >> 24);          // cigar ops the joined hit can have at most: the hits' own and one per closure
        for (int d = 1; d < nsegs; ++d) {
            if (!((has >> j) & 1u)) { ok = false; break; }
            ops += (int)((gap >> j) & 1u);
            const int nx = (int)((succ >> (4 * j)) & 15u);
            j = nx; sl |= (uint32_t)nx << (4 * d);
            ops += (int)(tab.head(nx).meta >> 24);
        }
        if (ok && ops > LEAN_C) return CHAINS_DECLINE;      // (such a chain could need the general tier: the whole read stays together)
        if (ok) {
#pragma unroll
            for (int c = 0; c < CHAINS_MAX; ++c) sel[c] = c == k ? sl : sel[c];
            ++k;
        }
    }
    // ranks by (contig, left of the leftmost hit, strand); two chains with one key: declined
    for (int a = 0; a < k; ++a) {
        uint32_t sa = 0;
#pragma unroll
        for (int c = 0; c < CHAINS_MAX; ++c) sa = c == a ? sel[c] : sa;
        const SpanHitHead ra = tab.head((int)(sa & 15u));
        const bool aa = (ra.meta & SH_ANTI) != 0;
        const int32_t la = aa ? tab.head((int)((sa >> (4 * (nsegs - 1))) & 15u)).left : ra.left;
        int rank = 0;
        for (int b = 0; b < k; ++b) {
            if (b == a) continue;
            uint32_t sb = 0;
#pragma unroll
            for (int c = 0; c < CHAINS_MAX; ++c) sb = c == b ? sel[c] : sb;
            const SpanHitHead rb = tab.head((int)(sb & 15u));
            const bool ab = (rb.meta & SH_ANTI) != 0;
            const int32_t lb = ab ? tab.head((int)((sb >> (4 * (nsegs - 1))) & 15u)).left : rb.left;
            if (rb.ref_id == ra.ref_id && lb == la && ab == aa) return CHAINS_DECLINE;
            const bool less = rb.ref_id != ra.ref_id ? rb.ref_id < ra.ref_id : (lb != la ? lb < la : (!ab && aa));
            rank += less ? 1 : 0;
        }
#pragma unroll
        for (int c = 0; c < CHAINS_MAX; ++c) q[c] = c == a ? rank : q[c];
    }
    return k;
}
// the padded size of a group of k chains: groups of one size are laid out at multiples of that size, so that none straddles a wave
THJ_HD int chains_padded(int k) { return k <= 1 ? 1 : (k <= 2 ? 2 : (k <= 4 ? 4 : 8)); }

// ---- tier 0: reads whose single hits per segment are plain matches that abut in read order ----------------
// (an unspliced read cut into segments: ~60 % of real data).  merge_chain leaves every pair untouched
// (dist == 0, :1591), the final concatenation (:1888-1944) fuses the MATCH ops into one, so the joined hit is
// {leftmost left, [len M], sum of mismatches}.  Returns SPAN_NEED_LEAN when the read is not of that shape.
enum { SPAN_NEED_LEAN = 4, SPAN_LEAN_CLASSES = 4, SPAN_NEED_CHAIN = 6 };     // NEED_LEAN / NEED_CHAIN carry the read's class in bits 8.. of the status

// ---- tier 0 helpers ------------------------------------------------------------------------------------------
// Where the read's planes come from: global memory (any W), or registers for reads of up to 128 bases (W <= 2), loaded
// before anything else of the read so that their latency hides behind the offset and hit fetches.
struct MemRead {
    const u64* rp; int W;
    THJ_HD Planes fetch(int start, int len) const { return r_fetch(rp, W, start, len); }
};
struct RegRead {
    u64 v[6];               // lo[0..1], hi[0..1], nm[0..1]; second words 0 when W == 1
    THJ_HD Planes fetch(int start, int len) const {
        const bool w1 = (start >> 6) != 0;
        const unsigned sft = (unsigned)(start & 63);
        const u64 m = lowmask(len);
        Planes r;
        r.lo = funnel(w1 ? v[1] : v[0], w1 ? 0ull : v[1], sft) & m;
        r.hi = funnel(w1 ? v[3] : v[2], w1 ? 0ull : v[3], sft) & m;
        r.nm = funnel(w1 ? v[5] : v[4], w1 ? 0ull : v[5], sft) & m;
        return r;
    }
};
template <class Src>
THJ_HD bool src_is_own_revcomp(const Src& src, int rl) {         // read_is_own_revcomp on either source
    for (int off = 0; off < rl; off += 64) {
        int l = rl - off < 64 ? rl - off : 64;
        Planes f = src.fetch(off, l);
        Planes r = rc_piece(src.fetch(rl - off - l, l), l);
        if (f.lo != r.lo || f.hi != r.hi || f.nm != r.nm) return false;
    }
    return true;
}
struct ContigAcc { MdBuf md; int mismatch, both_n, AS, pos_mm; };
// bowtie_sam_extra over one 64-base piece of the single MATCH op (bwt_map.cpp:2467-2648)
THJ_HD void contig_piece(const Params& p, const Planes& r, const Planes& sq, int l, int off, const uint8_t* qual, bool qrev, int rl, ContigAcc& a) {
    u64 m = dna5_mism(r, sq, l);
    if (THJ_EXPF(32)) m = 0;
    u64 bn = r.nm & sq.nm & lowmask(l);
    a.both_n += popc(bn);
    a.AS -= p.bowtie2_penalty_for_N * popc(bn);
    int last = 0;
    while (m) {
        int b = ctz(m);
        m &= m - 1;
        ++a.mismatch;
        int sp = off + b;
        if (((r.nm | sq.nm) >> b) & 1ull) a.AS -= p.bowtie2_penalty_for_N;
        else {
            int q = THJ_EXPF(2) ? 30 : (int)qual[qrev ? rl - 1 - sp : sp] - 33; if (q > 40) q = 40;
            a.AS -= p.bowtie2_min_penalty + ((p.bowtie2_max_penalty - p.bowtie2_min_penalty) * q) / 40;
        }
        a.pos_mm += b - last;
        md_put_int_char(a.md, a.pos_mm, "ACGTN"[plane_code(r, b)]);
        a.pos_mm = 0; last = b + 1;
    }
    a.pos_mm += l - last;
}
template <class Src, class Sink>
THJ_HD int contig_finish(const Genome& g, const Params& p, const Src& src, uint32_t ref_id, int left, int total, int mm8, bool anti, int nsegs,
                         int rl, const uint8_t* qual, uint32_t read_idx, Sink& sink) {
    bool qrev = false;
    if (nsegs == 1) qrev = anti;
    else if (anti) qrev = !src_is_own_revcomp(src, rl);      // merge_chain :1966-1978: reversed qual unless rc(read) == read
    ContigAcc a;
    md_init(a.md);
    a.mismatch = a.both_n = a.AS = a.pos_mm = 0;
    // the first two pieces (reads of up to 128 bases: all of it) are fetched together, then consumed
    {
        Planes r[2], sq[2]; int l[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int off = 64 * c;
            int lc = total - off < 64 ? total - off : 64;
            if (off + lc > rl) lc = rl - off;
            l[c] = lc;
            if (lc > 0) {
                r[c] = g_fetch(g, ref_id, THJ_EXPF(4) ? 0 : (int64_t)left + off);
                sq[c] = anti ? rc_piece(src.fetch(rl - off - lc, lc), lc) : src.fetch(off, lc);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (l[c] > 0) contig_piece(p, r[c], sq[c], l[c], 64 * c, qual, qrev, rl, a);
    }
    for (int off = 128; off < total; off += 64) {
        int l = total - off < 64 ? total - off : 64;
        if (off + l > rl) l = rl - off;
        if (l <= 0) break;
        Planes r = g_fetch(g, ref_id, (int64_t)left + off);
        Planes sq = anti ? rc_piece(src.fetch(rl - off - l, l), l) : src.fetch(off, l);
        contig_piece(p, r, sq, l, off, qual, qrev, rl, a);
    }
    md_put_int(a.md, a.pos_mm);
    const MdBuf& md = a.md;
    const int mismatch = a.mismatch, both_n = a.both_n, AS = a.AS;
    if (nsegs > 1 && !(mismatch == mm8 || mismatch + both_n == mm8)) return SPAN_OK;   // check_editdist_consistency
    uint32_t wds[32];
    wds[0] = read_idx; wds[1] = ref_id; wds[2] = (uint32_t)left;
    wds[3] = (anti ? 1u : 0u) | ((uint32_t)mm8 << 8) | ((uint32_t)mm8 << 16) | (1u << 24);
    wds[4] = ((uint32_t)AS & 0xFFFFu) | ((uint32_t)(mismatch & 0xFF) << 16);                 // AS, XM, XO = 0
    wds[5] = ((uint32_t)(md.len > 40 ? 255 : md.len) << 8);                                  // XG = 0, md_len (255: on the host), order = 0
    wds[6] = cig(OP_MATCH, (uint32_t)total);
#pragma unroll
    for (int q = 7; q < 22; ++q) wds[q] = 0;
#pragma unroll
    for (int q = 0; q < 5; ++q) { wds[22 + 2 * q] = (uint32_t)md.w[q]; wds[23 + 2 * q] = (uint32_t)(md.w[q] >> 32); }
    sink.emit_words(wds);
    return SPAN_OK;
}

// ---- the finish of a joined hit (thj_k_finish) -----------------------------------------------------------------------------------
// check_editdist_consistency + bowtie_sam_extra (bwt_map.cpp:2349-2648) + the record, as lean_finish_check / sam_extra do them, for
// a joined hit that has passed valid_hit and the filters (chain_join): tier 0's finishing path generalised from one MATCH run to
// the runs of a cigar.  The genome pieces of the first FIN_PRE 64-base pieces of its MATCH ops are fetched together (a read with
// one junction has two or three), then the ops are walked in order.  `src`: the read's planes (registers for reads of up to 128 bases).
static constexpr int FIN_PRE = 3;
template <class Src>
THJ_HD bool joined_extras(const Genome& g, const Params& p, const RAln& h, bool one_seg, const Src& src, int rl, const uint8_t* qual, Extras& e) {
    const bool anti = h.anti != 0;
    bool qrev;
    if (one_seg) qrev = anti;
    else qrev = anti ? !src_is_own_revcomp(src, rl) : false;        // merge_chain :1966-1978
    // pass 1: where the first FIN_PRE pieces lie (a plain loop over the ops: code size matters more here than a select per op)
    int64_t pr0 = h.left, pr1 = h.left, pr2 = h.left;
    {
        int np = 0, pos_seq = 0; int64_t pos_ref = h.left;
        for (int i = 0; i < h.n && np < FIN_PRE; ++i) {
            const uint32_t ci = h.c.get(i);
            const int op = cig_op(ci), len = (int)cig_len(ci);
            if (op == OP_MATCH) {
                for (int off = 0; off < len && np < FIN_PRE; off += 64) {
                    if (pos_seq + off >= rl) break;                  // (pass 2 leaves the op there too)
                    pr0 = np == 0 ? pos_ref + off : pr0; pr1 = np == 1 ? pos_ref + off : pr1; pr2 = np == 2 ? pos_ref + off : pr2;
                    ++np;
                }
                pos_seq += len; pos_ref += len;
            } else if (op == OP_INS) pos_seq += len;
            else if (op == OP_DEL || op == OP_REF_SKIP) pos_ref += len;
        }
    }
    // unconditional: an unused entry reads the alignment's first piece again
    const Planes gp0 = g_fetch(g, h.ref_id, pr0), gp1 = g_fetch(g, h.ref_id, pr1), gp2 = g_fetch(g, h.ref_id, pr2);
    // pass 2: bowtie_sam_extra over the ops; the pieces of the MATCH ops all go through the one contig_piece below
    ContigAcc a;
    md_init(a.md);
    a.mismatch = a.both_n = a.AS = a.pos_mm = 0;
    int opens = 0, conts = 0, piece = 0, pos_seq = 0, i = 0, off = 0;
    int64_t pos_ref = h.left;
    while (i < h.n) {
        const uint32_t ci = h.c.get(i);
        const int op = cig_op(ci), len = (int)cig_len(ci);
        if (op == OP_MATCH) {
            int l = len - off < 64 ? len - off : 64;
            if (pos_seq + off + l > rl) l = rl - pos_seq - off;
            if (l > 0) {
                Planes r;
                if (piece < FIN_PRE) r = piece == 0 ? gp0 : (piece == 1 ? gp1 : gp2);
                else r = g_fetch(g, h.ref_id, pos_ref + off);
                const Planes sq = anti ? rc_piece(src.fetch(rl - (pos_seq + off) - l, l), l) : src.fetch(pos_seq + off, l);
                contig_piece(p, r, sq, l, pos_seq + off, qual, qrev, rl, a);
                ++piece;
            }
            off += 64;
            if (off >= len || l <= 0) { pos_seq += len; pos_ref += len; off = 0; ++i; }
            continue;
        }
        if (op == OP_INS) {
            pos_seq += len;
            a.AS -= p.bowtie2_read_gap_open + p.bowtie2_read_gap_cont * len;
            ++opens; conts += len;
        } else if (op == OP_DEL) {
            a.AS -= p.bowtie2_ref_gap_open + p.bowtie2_ref_gap_cont * len;
            ++opens; conts += len;
            md_put_int_char(a.md, a.pos_mm, '^');
            const Planes r = g_fetch(g, h.ref_id, pos_ref);
            for (int k = 0; k < len && k < 64; ++k) md_push(a.md, "ACGTN"[plane_code(r, k)]);
            pos_ref += len; a.pos_mm = 0;
        } else if (op == OP_REF_SKIP) pos_ref += len;
        ++i;
    }
    md_put_int(a.md, a.pos_mm);
    e.md = a.md; e.AS = a.AS; e.XM = a.mismatch; e.XO = opens; e.XG = conts; e.both_n = a.both_n;
    // check_editdist_consistency (inside merge_chain in the reference) shares the pass's counts
    if (!one_seg && !(e.XM == h.mm || e.XM + e.both_n == h.mm)) return false;
    return true;
}

// one joined hit (its packed words) -> the filters' last part and the tags; false: not reported.  The caller emits the record
// (emit_aln) once it knows the record's rank among the read's records.
THJ_HD bool joined_prepare(const Genome& g, const Params& p, const Q16& ja, const Q16& jb, const Q16& jc, const u64* planes, int W,
                           const uint16_t* read_len, const uint8_t* quals, int qual_stride, RAln& res, Extras& e) {
    bool one_seg; int q, k;
    joined_unpack(ja, jb, jc, res, one_seg, q, k);
    const uint32_t r = ja.x;
    const u64* rp = planes + (u64)r * (uint32_t)(3 * W);
    const int rl = (int)read_len[r];
    const uint8_t* qual = quals + (u64)r * (uint32_t)qual_stride;
    if (W <= 2) {
        RegRead rw;
#pragma unroll
        for (int i = 0; i < 6; ++i) rw.v[i] = 0;
        if (W == 2) {
#pragma unroll
            for (int i = 0; i < 6; ++i) rw.v[i] = rp[i];
        } else { rw.v[0] = rp[0]; rw.v[2] = rp[1]; rw.v[4] = rp[2]; }
        return joined_extras(g, p, res, one_seg, rw, rl, qual, e);
    }
    return joined_extras(g, p, res, one_seg, MemRead{rp, W}, rl, qual, e);
}
template <class Sink>
THJ_HD bool joined_finish(const Genome& g, const Params& p, const Q16& ja, const Q16& jb, const Q16& jc, const u64* planes, int W,
                          const uint16_t* read_len, const uint8_t* quals, int qual_stride, int order, Sink& sink) {
    RAln res; Extras e;
    if (!joined_prepare(g, p, ja, jb, jc, planes, W, read_len, quals, qual_stride, res, e)) return false;
    emit_aln(sink, ja.x, order, res, e);
    return true;
}
THJ_HD void joined_pack(const RAln& res, uint32_t read, bool one_seg, int q, int k, Q16& ja, Q16& jb, Q16& jc) {
    ja.x = read; ja.y = res.ref_id; ja.z = (uint32_t)res.left; ja.w = joined_meta(res, one_seg, q, k);
    jb.x = res.c.v[0]; jb.y = res.c.v[1]; jb.z = res.c.v[2]; jb.w = res.c.v[3];
    jc.x = res.c.v[4]; jc.y = res.c.v[5]; jc.z = res.c.v[6]; jc.w = res.c.v[7];
}

// the read's segment offsets, fetched in one go (the kernel does this one read ahead)
template <int MS>
THJ_HD void contig_offsets(const uint32_t* so, int nseg, uint32_t (&sv)[MS + 1]) {
#pragma unroll
    for (int s = 0; s <= MS; ++s) sv[s] = s <= nseg ? so[s <= nseg ? s : 0] : 0u;
}
template <int MS = SPAN_MAXSEG, class Sink>
THJ_HD int span_read_contig_pre(const Genome& g, const Params& p, const SpanHit* hits, const uint32_t (&sv)[MS + 1], int nseg,
                                const u64* rp, int W, int rl, const uint8_t* qual, uint32_t read_idx, Sink& sink,
                                const SpanHitHead* gheads = nullptr, ChainEntry* ent = nullptr, bool chains = true) {
    // Memory round trips, not arithmetic, bound this tier: the read's planes (when they fit six registers) and the
    // segment offsets are fetched in one go, then every hit head in one go, and only then is anything decided.
    RegRead rw;
#pragma unroll
    for (int k = 0; k < 6; ++k) rw.v[k] = 0;
    if (W == 2) {
#pragma unroll
        for (int k = 0; k < 6; ++k) rw.v[k] = rp[k];
    } else if (W == 1) { rw.v[0] = rp[0]; rw.v[2] = rp[1]; rw.v[4] = rp[2]; }
    if (sv[1] == sv[0]) return SPAN_OK;
    int nsegs = 0;
    {
        bool open = true;
#pragma unroll
        for (int s = 0; s < MS; ++s) { open = open && s < nseg && sv[s + 1] > sv[s]; nsegs += open ? 1 : 0; }
    }
    bool single = true;
#pragma unroll
    for (int s = 0; s < MS; ++s) single = single && (s >= nsegs || sv[s + 1] - sv[s] == 1u);
    uint32_t last_so = sv[0];
#pragma unroll
    for (int s = 1; s < MS; ++s) last_so = (s == nsegs - 1) ? sv[s] : last_so;
    if (!single) {
        if (!(hits[last_so].meta & SH_END)) return SPAN_OK;
        return SPAN_NEED_GENERIC;
    }
    // one hit per segment: the read's hits are hits[sv[0] .. sv[0] + nsegs)
    SpanHitHead hh[MS];
#pragma unroll
    for (int s = 0; s < MS; ++s) {
        // unconditional (a segment the read does not have reads the first hit again and is never looked at): under `if (s < nsegs)` every
        // load sat in a branch of its own with an s_waitcnt vmcnt(0) behind it -- the heads came one round trip after the other
        const Q16 q = load_head(hits, gheads, (u64)sv[0] + (u64)(s < nsegs ? s : 0));
        hh[s] = SpanHitHead{q.x, (int32_t)q.y, q.z, q.w};
    }
    uint32_t last_meta = hh[0].meta;
#pragma unroll
    for (int s = 1; s < MS; ++s) last_meta = (s == nsegs - 1) ? hh[s].meta : last_meta;
    if (!(last_meta & SH_END)) return SPAN_OK;
    if (p.fusion_search) {                 // a fused segment hit (junction-db fusion contig): only the fusion tier reads its cigar
        uint32_t any = 0;
#pragma unroll
        for (int s = 0; s < MS; ++s) any |= s < nsegs ? hh[s].meta : 0u;
        if (any & SH_FUSED) return SPAN_NEED_GENERIC;
    }
    const SpanHitHead h0 = hh[0];
    const bool anti = (h0.meta & SH_ANTI) != 0;
    bool lean = (h0.meta >> 24) != 1u || cig_op(h0.cigar0) != OP_MATCH;     // (class 0)
    int total = (int)cig_len(h0.cigar0);
    int mm = (int)((h0.meta >> 8) & 0xFF);
    int left = h0.left, edge = anti ? h0.left : h0.left + total;       // where the next segment must abut
    int gap_at = 0;                                   // first segment that does not abut its predecessor (0: none / other reason)
#pragma unroll
    for (int s = 1; s < MS; ++s) {
        if (s < nsegs) {
            const SpanHitHead h = hh[s];
            const int len = (int)cig_len(h.cigar0);
            lean = lean || (h.meta >> 24) != 1u || cig_op(h.cigar0) != OP_MATCH;
            lean = lean || h.ref_id != h0.ref_id || ((h.meta & SH_ANTI) != 0) != anti;    // lean path decides (no alignment)
            const bool was = lean;
            if (anti) { lean = lean || h.left + len != edge; edge = h.left; left = h.left; }
            else { lean = lean || h.left != edge; edge = h.left + len; }
            gap_at = (lean && !was && gap_at == 0) ? s : gap_at;
            total += len;
            mm += (int)((h.meta >> 8) & 0xFF);
        }
    }
    // the class of a handed-down read = the step of lean_join's chain loop that will meet its (first) gap: tier 1 walks its
    // work list class by class, so that the lanes of a wave run the closure code in the same iteration
    if (lean) {
        const int cls = gap_at ? (anti ? nsegs - gap_at : gap_at) & (SPAN_LEAN_CLASSES - 1) : 0;
        // a read of at most CHAIN_MAXSEG segments travels as a chain entry (its hits are consecutive records)
        if (ent != nullptr && chains && nsegs <= CHAIN_MAXSEG && !p.fusion_search) {
            ent->read = read_idx; ent->meta = chain_meta(nsegs, 0, 1, rl); ent->spare0 = ent->spare1 = 0;
#pragma unroll
            for (int s = 0; s < CHAIN_MAXSEG; ++s) ent->hit[s] = sv[0] + (uint32_t)(s < nsegs ? s : 0);
            return SPAN_NEED_CHAIN | (cls << 8);
        }
        return SPAN_NEED_LEAN | (cls << 8);
    }
    if (THJ_EXPF(16)) return SPAN_OK;
    const int mm8 = mm & 0xFF;                       // BowtieHit keeps mismatches / edit_dist in unsigned chars
    if (mm8 > p.read_mismatches || mm8 > p.read_edit_dist) return SPAN_OK;          // :2810-2813 (gap length 0)
    if (g_len(g, h0.ref_id) == 0) return SPAN_OK;    // check_editdist_consistency / bowtie_sam_extra need the contig
    if (W <= 2) return contig_finish(g, p, rw, h0.ref_id, left, total, mm8, anti, nsegs, rl, qual, read_idx, sink);
    return contig_finish(g, p, MemRead{rp, W}, h0.ref_id, left, total, mm8, anti, nsegs, rl, qual, read_idx, sink);
}
template <int MS = SPAN_MAXSEG, class Sink>
THJ_HD int span_read_contig(const Genome& g, const Params& p, const SpanHit* hits, const uint32_t* so, int nseg,
                            const u64* rp, int W, int rl, const uint8_t* qual, uint32_t read_idx, Sink& sink, ChainEntry* ent = nullptr) {
    uint32_t sv[MS + 1];
    contig_offsets<MS>(so, nseg, sv);
    return span_read_contig_pre<MS>(g, p, hits, sv, nseg, rp, W, rl, qual, read_idx, sink, nullptr, ent);
}

}  // namespace thj
