// thj_core.h -- bit-parallel building blocks of the junction-discovery kernels.
//
// Everything here is a pure function of its arguments (loads through const
// pointers, results through a caller-supplied sink), written once and compiled
// for gfx950 by hipcc as __device__ code.  The same header also compiles as
// plain C++ so that tests/hostsim can single-step the logic on a CPU; that
// build is test-only and is never linked into libthj_hip.so.
//
// Data model (see include/thj.h): genome = 32-byte blocks of 64 bases
// {lo plane, hi plane, N mask, 0}; reads = {lo[W], hi[W], N[W]} planes.  With
// bases as bit-planes a whole splice-window comparison is a handful of 64-bit
// XOR / AND / popcount operations per thread -- no per-base loops, no MFMA.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define THJ_HD __host__ __device__ __forceinline__
#else
#define THJ_HD inline
#endif

// THJ_EXP: developer-only build (tools/build_exp.sh, tools/exp_bench.py) that can switch parts of a kernel off to
// see what they cost.  One flag word per translation unit, set from the THJ_EXP_FLAGS environment variable.
#if defined(THJ_EXP) && defined(__HIPCC__)
static __device__ int thj_exp_flags;
#define THJ_EXPF(b) (thj_exp_flags & (b))
#else
#define THJ_EXPF(b) 0
#endif

namespace thj {

typedef unsigned long long u64;

// ---- bit helpers ---------------------------------------------------------
THJ_HD int popc(u64 x) { return __builtin_popcountll(x); }
THJ_HD int popc32(uint32_t x) { return __builtin_popcount(x); }
THJ_HD int ctz(u64 x) { return __builtin_ctzll(x); }   // x != 0
THJ_HD int clz(u64 x) { return __builtin_clzll(x); }   // x != 0
THJ_HD u64 lowmask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }   // n >= 0
THJ_HD u64 brev64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __brevll(x);
#else
    x = ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);
    x = ((x >> 2) & 0x3333333333333333ull) | ((x & 0x3333333333333333ull) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    return __builtin_bswap64(x);
#endif
}
// x >> s | y << (64 - s) for s in [0, 63] without the undefined 64-bit shift
THJ_HD u64 funnel(u64 x, u64 y, unsigned s) { return (x >> s) | ((y << 1) << (63u - s)); }

// ---- records ---------------------------------------------------------------
struct Hit {            // == thj_hit, 16 bytes
    uint32_t ref_id;
    int32_t left;
    int32_t right;
    uint32_t meta;      // flags | edit_dist<<8 | mismatches<<16 | read_len<<24
};
THJ_HD bool hit_anti(const Hit& h) { return (h.meta & 1u) != 0; }
THJ_HD bool hit_end(const Hit& h) { return (h.meta & 2u) != 0; }
THJ_HD int hit_ed(const Hit& h) { return (int)((h.meta >> 8) & 0xFF); }
THJ_HD int hit_rlen(const Hit& h) { return (int)((h.meta >> 24) & 0xFF); }

struct Genome {
    const u64* blocks;            // 4 u64 per 64-base block
    const uint32_t* contig_blk;   // [n_contigs+1]
    const int32_t* contig_len;    // [n_contigs]; 0 = no sequence
    int32_t n_contigs;
};

struct Params {          // == thj_params (include/thj.h)
    int32_t segment_length, segment_mismatches, min_segment_intron, max_segment_intron;
    int32_t max_insertion_length, max_deletion_length, max_seg_multihits;
    int32_t inner_dist_mean, inner_dist_std_dev, library_type, bowtie2, read_side;
    int32_t min_report_intron, max_report_intron, min_anchor_len;
    int32_t read_mismatches, read_gap_length, read_edit_dist;
    int32_t bowtie2_max_penalty, bowtie2_min_penalty, bowtie2_penalty_for_N;
    int32_t bowtie2_read_gap_open, bowtie2_read_gap_cont, bowtie2_ref_gap_open, bowtie2_ref_gap_cont;
    int32_t fusion_anchor_length, fusion_min_dist, fusion_search;
};

struct Planes { u64 lo, hi, nm; };

// 64 bases of contig `ref_id` (1-based) starting at `pos` (>= 0).
THJ_HD Planes g_fetch(const Genome& g, uint32_t ref_id, int64_t pos) {
    u64 gpos = (u64)g.contig_blk[ref_id - 1] * 64ull + (u64)pos;
    const u64* p = g.blocks + (gpos >> 6) * 4;
    unsigned s = (unsigned)(gpos & 63);
    Planes r;
    r.lo = funnel(p[0], p[4], s);
    r.hi = funnel(p[1], p[5], s);
    r.nm = funnel(p[2], p[6], s);
    return r;
}
// the same at a global base position (contig_blk[ref_id - 1] * 64 + pos: for callers that hold the contig's base already)
THJ_HD Planes g_fetch_abs(const Genome& g, u64 gpos) {
    const u64* p = g.blocks + (gpos >> 6) * 4;
    unsigned s = (unsigned)(gpos & 63);
    Planes r;
    r.lo = funnel(p[0], p[4], s);
    r.hi = funnel(p[1], p[5], s);
    r.nm = funnel(p[2], p[6], s);
    return r;
}
THJ_HD int32_t g_len(const Genome& g, uint32_t ref_id) {
    return (ref_id == 0 || (int32_t)ref_id > g.n_contigs) ? 0 : g.contig_len[ref_id - 1];
}

// `len` (1..64) bases of a read starting at `start`; rp = this read's planes.
THJ_HD Planes r_fetch(const u64* rp, int W, int start, int len) {
    int w = start >> 6;
    unsigned s = (unsigned)(start & 63);
    Planes r;
    u64 m = lowmask(len);
    bool two = (w + 1 < W);
    r.lo = funnel(rp[w], two ? rp[w + 1] : 0, s) & m;
    r.hi = funnel(rp[W + w], two ? rp[W + w + 1] : 0, s) & m;
    r.nm = funnel(rp[2 * W + w], two ? rp[2 * W + w + 1] : 0, s) & m;
    return r;
}

// reverse complement of a `len`-base piece; N stays N (reads.cpp:189-207,
// seqan::reverseComplement on String<char>).
THJ_HD Planes rc_piece(Planes a, int len) {
    u64 m = lowmask(len);
    u64 l = ~a.lo & ~a.nm & m, h = ~a.hi & ~a.nm & m, n = a.nm & m;
    Planes r;
    r.lo = brev64(l) >> (64 - len);
    r.hi = brev64(h) >> (64 - len);
    r.nm = brev64(n) >> (64 - len);
    return r;
}

// ---- juncs_db: bases of a genome piece as text (juncs_db.cpp:73-233 print_splice / print_insertion / print_fusion) ----
// out[0..len) = contig[start .. start+len), or its reverse complement (seqan::reverseComplement on Dna5: N stays N)
THJ_HD void piece_text(const Genome& g, uint32_t ref_id, int32_t start, int32_t len, bool rc, char* out) {
    for (int32_t off = 0; off < len; off += 64) {
        const int l = len - off < 64 ? len - off : 64;
        const Planes p = g_fetch(g, ref_id, (int64_t)start + off);
        for (int k = 0; k < l; ++k) {
            int code = ((p.nm >> k) & 1ull) ? 4 : (int)(((p.lo >> k) & 1ull) | (((p.hi >> k) & 1ull) << 1));
            if (rc) { if (code < 4) code = 3 - code; out[len - 1 - (off + k)] = "ACGTN"[code]; }
            else out[off + k] = "ACGTN"[code];
        }
    }
}

// ---- 128-base variants --------------------------------------------------------------------------------------
// With segment_length > 32 a 2L read piece (indel search) or an L+16 support read (window scan) no longer fits one
// 64-bit plane word; the same algorithms then run on 128-bit words (two registers per plane, the compiler splits
// them).  `WT` below is u64 or u128.
typedef unsigned __int128 u128;
THJ_HD int popc(u128 x) { return popc((u64)x) + popc((u64)(x >> 64)); }
THJ_HD int ctz(u128 x) { return (u64)x ? ctz((u64)x) : 64 + ctz((u64)(x >> 64)); }            // x != 0
THJ_HD int clz(u128 x) { return (u64)(x >> 64) ? clz((u64)(x >> 64)) : 64 + clz((u64)x); }    // x != 0
template <class WT> THJ_HD WT lowmask_t(int n) { return n >= (int)(8 * sizeof(WT)) ? ~(WT)0 : (((WT)1 << n) - (WT)1); }

template <class WT> struct PlanesT { WT lo, hi, nm; };
template <class WT> struct Fetch;
template <> struct Fetch<u64> {
    static THJ_HD PlanesT<u64> genome(const Genome& g, uint32_t ref_id, int64_t pos) { Planes a = g_fetch(g, ref_id, pos); return {a.lo, a.hi, a.nm}; }
    static THJ_HD PlanesT<u64> read(const u64* rp, int W, int start, int len) { Planes a = r_fetch(rp, W, start, len); return {a.lo, a.hi, a.nm}; }
    static THJ_HD PlanesT<u64> rc(PlanesT<u64> a, int len) { Planes r = rc_piece(Planes{a.lo, a.hi, a.nm}, len); return {r.lo, r.hi, r.nm}; }
};
template <> struct Fetch<u128> {
    static THJ_HD PlanesT<u128> genome(const Genome& g, uint32_t ref_id, int64_t pos) {
        Planes a = g_fetch(g, ref_id, pos), b = g_fetch(g, ref_id, pos + 64);
        return {(u128)a.lo | ((u128)b.lo << 64), (u128)a.hi | ((u128)b.hi << 64), (u128)a.nm | ((u128)b.nm << 64)};
    }
    static THJ_HD PlanesT<u128> read(const u64* rp, int W, int start, int len) {      // len 1..128
        Planes a = r_fetch(rp, W, start, len < 64 ? len : 64);
        Planes b{0, 0, 0};
        if (len > 64) b = r_fetch(rp, W, start + 64, len - 64);
        return {(u128)a.lo | ((u128)b.lo << 64), (u128)a.hi | ((u128)b.hi << 64), (u128)a.nm | ((u128)b.nm << 64)};
    }
    static THJ_HD PlanesT<u128> rc(PlanesT<u128> a, int len) {
        const u128 m = lowmask_t<u128>(len);
        const u128 l = ~a.lo & ~a.nm & m, h = ~a.hi & ~a.nm & m, n = a.nm & m;
        auto rev = [](u128 x) { return ((u128)brev64((u64)x) << 64) | (u128)brev64((u64)(x >> 64)); };
        return {rev(l) >> (128 - len), rev(h) >> (128 - len), rev(n) >> (128 - len)};
    }
};

// ---- juncs_from_ref_segs<RecordSegmentJuncs>, POINT_DIR_BOTH ----------------
// One RefSeg window, all three motif pairs fused (segment_juncs.cpp:2052-2377 x
// :3618-3649).  Only the two window ends are touched.  Sink: junction(ref,left,right,anti).
template <class WT, class Sink>
THJ_HD void window_scan(const Genome& g, const Params& p, uint32_t ref_id, int32_t seg_left, int32_t seg_right,
                        bool antisense, PlanesT<WT> sup, int read_len, Sink& sink) {
    constexpr int BITS = (int)(8 * sizeof(WT));
    int32_t clen = g_len(g, ref_id);
    if (clen == 0) return;                                        // :2105-2108
    if (seg_left < 0 || seg_right >= clen - 1) return;            // :2154
    int seg_len = seg_right - seg_left;
    if (read_len < 2 || read_len > BITS - 2 || seg_len < read_len) return;
    if ((int64_t)seg_left + seg_len - read_len - 2 < 0) return;   // unreachable: find_gaps windows span >= min intron

    bool skip_fwd = false, skip_rev = false;                      // :2110-2138
    if (p.library_type == 2) {
        if (p.read_side == 1) { if (antisense) skip_rev = true; else skip_fwd = true; }
        else if (p.read_side == 2) { if (antisense) skip_fwd = true; else skip_rev = true; }
    }
    if (p.library_type == 3) {
        if (p.read_side == 1) { if (antisense) skip_fwd = true; else skip_rev = true; }
        else if (p.read_side == 2) { if (antisense) skip_rev = true; else skip_fwd = true; }
    }

    const PlanesT<WT> gl = Fetch<WT>::genome(g, ref_id, seg_left);                                  // window start (N -> A: mask ignored)
    const PlanesT<WT> gr = Fetch<WT>::genome(g, ref_id, (int64_t)seg_left + seg_len - read_len - 2); // window end, 2 bases early

    const WT M = lowmask_t<WT>(read_len);
    // left_mismatches[]: loop runs i in [0, read_len-1) and stops at the third mismatch (:2187-2203)
    const WT mL = ((gl.lo ^ sup.lo) | (gl.hi ^ sup.hi) | sup.nm) & lowmask_t<WT>(read_len - 1);
    int to = read_len - 2;
    {
        WT t = mL;
        t &= t - 1;
        t &= t - 1;
        if (t) to = ctz(t);
    }
    // right_mismatches[]: from the top down, stops at the third mismatch and leaves
    // the entries below it at 0 (:2205-2218)
    const WT mR = (((gr.lo >> 2) ^ sup.lo) | ((gr.hi >> 2) ^ sup.hi) | sup.nm) & M;
    int t3 = -1;
    {
        WT u = mR;
        if (u) u &= ~((WT)1 << (BITS - 1 - clz(u)));
        if (u) u &= ~((WT)1 << (BITS - 1 - clz(u)));
        if (u) t3 = BITS - 1 - clz(u);
    }

    // dinucleotide masks: bit i set <=> bases (i, i+1) spell the dinucleotide
    const WT lA = ~gl.lo & ~gl.hi, lC = gl.lo & ~gl.hi, lG = ~gl.lo & gl.hi, lT = gl.lo & gl.hi;
    const WT rA = ~gr.lo & ~gr.hi, rC = gr.lo & ~gr.hi, rG = ~gr.lo & gr.hi, rT = gr.lo & gr.hi;
    const WT l_GT = lG & (lT >> 1), l_GC = lG & (lC >> 1), l_AT = lA & (lT >> 1), l_CT = lC & (lT >> 1);
    const WT r_AG = rA & (rG >> 1), r_AC = rA & (rC >> 1), r_GC = rG & (rC >> 1), r_AT = rA & (rT >> 1);
    // partner sits at window offset pos = seg_len-(read_len-i)-2, i.e. index i of `gr`
    WT fwd = (l_GT & r_AG) | (l_GC & r_AG) | (l_AT & r_AC);     // donor..acceptor
    WT rev = (l_CT & r_AC) | (l_CT & r_GC) | (l_GT & r_AT);     // rc(acceptor)..rc(donor)
    if (skip_fwd) fwd = 0;
    if (skip_rev) rev = 0;
    const WT range = lowmask_t<WT>(to + 1);
    WT cand = (fwd | rev) & range;
    while (cand) {
        int i = ctz(cand);
        cand &= cand - 1;
        int lm = popc(mL & lowmask_t<WT>(i));                           // left_mismatches[i-1]
        int rm = i > t3 ? popc(mR >> i) : (i == t3 ? 3 : 0);      // right_mismatches[i]
        if (lm + rm <= 2) {
            bool is_fwd = ((fwd >> i) & (WT)1) != 0;
            // RecordSegmentJuncs::record :1681-1695
            sink.junction(ref_id, (uint32_t)(seg_left + i - 1), (uint32_t)(seg_left + seg_len - read_len + i),
                          !is_fwd);
        }
    }
}

// ---- simpleSplitAlignment (segment_juncs.cpp:2390-2456) ----------------------
// mL/mR: mismatch masks of the shorter sequence against the left-/right-anchored
// reference.  Returns the first best insert position, -1 if len < 2.
template <class WT>
THJ_HD int split_bits(WT mL, WT mR, int len, int& min_err) {
    int best = len + 1, bp = -1;
    if (len >= 2) {
        int e = popc((mR & lowmask_t<WT>(len)) >> 1) + (int)(mL & (WT)1);
        for (int p = 1; p < len; ++p) {
            if (e < best) { best = e; bp = p; }
            e += (int)((mL >> p) & (WT)1) - (int)((mR >> p) & (WT)1);
        }
    }
    min_err = best;
    return bp;
}

// detect_small_deletion (segment_juncs.cpp:2557-2627). rd = read piece (rc'd when antisense).
template <class WT, class Sink>
THJ_HD void small_deletion(const Genome& g, PlanesT<WT> rd, int plen, const Hit& lh, const Hit& rh, Sink& sink) {
    int32_t clen = g_len(g, lh.ref_id);
    if (clen == 0) return;
    if (lh.left < 0) return;
    if (rh.right < plen) return;
    int disc = (rh.right - lh.left) - plen;
    if ((int64_t)lh.left + plen > clen) return;
    if (rh.right > clen) return;
    const PlanesT<WT> lg = Fetch<WT>::genome(g, lh.ref_id, lh.left);
    const PlanesT<WT> rg = Fetch<WT>::genome(g, lh.ref_id, (int64_t)rh.right - plen);
    const WT M = lowmask_t<WT>(plen);
    const WT mL = ((lg.lo ^ rd.lo) | (lg.hi ^ rd.hi) | lg.nm | rd.nm) & M;    // 'N' on either side is an error
    const WT mR = ((rg.lo ^ rd.lo) | (rg.hi ^ rd.hi) | rg.nm | rd.nm) & M;
    int min_err;
    int pos = split_bits(mL, mR, plen, min_err);
    if (pos < 0) return;
    int adj = (hit_rlen(lh) + hit_rlen(rh) >= plen) ? -1 : 0;
    if (min_err <= hit_ed(lh) + hit_ed(rh) + adj)
        sink.deletion(lh.ref_id, (uint32_t)(lh.left + pos - 1), (uint32_t)(lh.left + pos + disc));
}

// detect_small_insertion (segment_juncs.cpp:2470-2543).
template <class WT, class Sink>
THJ_HD void small_insertion(const Genome& g, PlanesT<WT> rd, int plen, const Hit& lh, const Hit& rh, u64 prio, Sink& sink) {
    int32_t clen = g_len(g, lh.ref_id);
    if (clen == 0) return;
    if (lh.left < 0) return;
    int disc = plen - (rh.right - lh.left);
    int64_t ge = rh.right;
    if (ge > clen) ge = clen;
    int glen = (int)(ge - lh.left);
    if (glen < 0) glen = 0;
    if (glen > plen || glen > (int)(8 * sizeof(WT))) return;
    const PlanesT<WT> gg = Fetch<WT>::genome(g, lh.ref_id, lh.left);          // DnaString: N -> A, mask ignored
    const WT M = lowmask_t<WT>(glen);
    // left_read = rd[0:glen], right_read = rd[plen-glen:plen]
    int sh = plen - glen;
    const WT mL = ((gg.lo ^ rd.lo) | (gg.hi ^ rd.hi) | rd.nm) & M;
    const WT mR = ((gg.lo ^ (rd.lo >> sh)) | (gg.hi ^ (rd.hi >> sh)) | (rd.nm >> sh)) & M;
    int min_err;
    int pos = split_bits(mL, mR, glen, min_err);
    if (pos < 0) return;
    int adj = (hit_rlen(lh) + hit_rlen(rh) >= plen) ? -1 : 0;
    if (min_err <= hit_ed(lh) + hit_ed(rh) + adj && pos + disc <= glen) {
        // inserted bases left_read[pos : pos+disc], 3 bits per base (A,C,G,T,N = 0..4)
        uint32_t seq = 0;
        for (int k = 0; k < disc; ++k) {
            int b = pos + k;
            uint32_t c = ((rd.nm >> b) & (WT)1) ? 4u : (uint32_t)(((rd.lo >> b) & (WT)1) | (((rd.hi >> b) & (WT)1) << 1));
            seq |= c << (3 * k);
        }
        sink.insertion(lh.ref_id, (uint32_t)(lh.left + pos - 1), disc, seq, prio);
    }
}

// ---- map_read_to_contig over the mate's flank (segment_juncs.cpp:2946-2973) ---
// Returns the first offset with the minimal Hamming distance (< 3), or -1.
THJ_HD int flank_scan(const Genome& g, uint32_t ref_id, int64_t left, int flen, Planes rd, int rlen) {
    // Bit-sliced over the offsets: for read base k, one 64-bit word says at which of the (up to 64 - rlen + 1) offsets
    // served by a 64-base fetch the genome differs from it; the words of the rlen bases are summed in a two-plane
    // counter with a sticky overflow.  About 26 word operations per read base instead of 25 per offset.
    int pos = -1, best = 3;
    int n_off = flen - rlen;                      // loop is i < contig_len - read_len
    int span = 64 - rlen + 1;                     // offsets served by one 64-base fetch
    for (int cs = 0; cs < n_off; cs += span) {
        Planes c = g_fetch(g, ref_id, left + cs);
        int lim = n_off - cs < span ? n_off - cs : span;
        u64 c0 = 0, c1 = 0, ov = 0;
        for (int k = 0; k < rlen; ++k) {
            const u64 ml = 0ull - ((rd.lo >> k) & 1ull), mh = 0ull - ((rd.hi >> k) & 1ull), mn = 0ull - ((rd.nm >> k) & 1ull);
            const u64 x = ((c.lo >> k) ^ ml) | ((c.hi >> k) ^ mh) | ((c.nm >> k) ^ mn);      // 'N'=='N' matches
            const u64 k0 = c0 & x;
            c0 ^= x;
            const u64 k1 = c1 & k0;
            c1 ^= k0;
            ov |= k1;
        }
        const u64 valid = lowmask(lim) & ~ov;
        const u64 z0 = ~c0 & ~c1 & valid, z1 = c0 & ~c1 & valid, z2 = ~c0 & c1 & valid;
        // the first offset with a strictly smaller distance replaces (scan order: ascending offsets)
        if (z0) { best = 0; pos = cs + ctz(z0); }
        else if (best > 1 && z1) { best = 1; pos = cs + ctz(z1); }
        else if (best > 2 && z2) { best = 2; pos = cs + ctz(z2); }
        if (best == 0) break;                     // nothing can replace a perfect match
    }
    return pos;
}

// The same scan for the two patterns of a rescue pair at once (the read's last bases and their reverse complement), offset by
// offset on 32-bit words.  flank_scan's bit-slicing pays for 64-bit operations on a 15-base pattern (~37 instructions per base,
// chunk and pattern: ~2200 per pair, 70 % of thj_k_segjuncs_rescue); here an offset costs two funnel shifts for the genome
// window plus, per pattern, two XORs, a masked OR, a popcount and a min over (distance << 16 | offset) keys -- the smallest key is
// the first offset with the smallest distance -- ~16 instructions, ~900 per pair, and the genome words of a 64-offset chunk are
// fetched together instead of one dependent fetch per 50 offsets and pattern.  Only for patterns of at most 16 bases without N
// and windows without N ('N' == 'N' is a match in the reference, which the planes-XOR of flank_scan handles): returns false
// otherwise and the caller takes flank_scan.
THJ_HD uint32_t funnel32(uint32_t w0, uint32_t w1, int s) { return (uint32_t)(((((u64)w1) << 32) | (u64)w0) >> s); }     // s in 0..31
THJ_HD bool flank_scan_pair(const Genome& g, uint32_t ref_id, int64_t left, int flen, const Planes& f, const Planes& r, int rlen, int& fpos, int& rpos) {
    const int n_off = flen - rlen;
    fpos = -1; rpos = -1;
    if (n_off <= 0) return true;
    if (rlen > 16 || n_off >= 65536 || (f.nm | r.nm) != 0) return false;
    const uint32_t m = (uint32_t)lowmask(rlen);
    const uint32_t fl = (uint32_t)f.lo, fh = (uint32_t)f.hi, rl = (uint32_t)r.lo, rh = (uint32_t)r.hi;
    uint32_t bf = 0xFFFFFFFFu, br = 0xFFFFFFFFu;
    for (int cs = 0; cs < n_off; cs += 64) {
        const int lim = n_off - cs < 64 ? n_off - cs : 64;          // offsets of this chunk; they read bases [0, lim + rlen - 1)
        const int need = lim + rlen - 1;
        const Planes a = g_fetch(g, ref_id, left + cs);
        Planes b; b.lo = 0; b.hi = 0; b.nm = 0;
        if (need > 64) b = g_fetch(g, ref_id, left + cs + 64);
        if (((need >= 64 ? a.nm : a.nm & lowmask(need)) | (need > 64 ? b.nm & lowmask(need - 64) : 0ull)) != 0) return false;
        const uint32_t wl[4] = {(uint32_t)a.lo, (uint32_t)(a.lo >> 32), (uint32_t)b.lo, (uint32_t)(b.lo >> 32)};
        const uint32_t wh[4] = {(uint32_t)a.hi, (uint32_t)(a.hi >> 32), (uint32_t)b.hi, (uint32_t)(b.hi >> 32)};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int smax = lim - 32 * j < 32 ? lim - 32 * j : 32;
#pragma unroll 4
            for (int sft = 0; sft < smax; ++sft) {
                const uint32_t gl = funnel32(wl[j], wl[j + 1], sft), gh = funnel32(wh[j], wh[j + 1], sft);
                const uint32_t o = (uint32_t)(cs + 32 * j + sft);
                const uint32_t kf = ((uint32_t)popc32(((gl ^ fl) | (gh ^ fh)) & m) << 16) | o;
                const uint32_t kr = ((uint32_t)popc32(((gl ^ rl) | (gh ^ rh)) & m) << 16) | o;
                bf = kf < bf ? kf : bf;
                br = kr < br ? kr : br;
            }
        }
        if ((bf >> 16) == 0 && (br >> 16) == 0) break;              // nothing replaces two perfect matches
    }
    if ((bf >> 16) < 3u) fpos = (int)(bf & 0xFFFFu);
    if ((br >> 16) < 3u) rpos = (int)(br & 0xFFFFu);
    return true;
}

enum { SLOT_NONE = -1, SLOT_BREAK = -2 };

// One (left hit, mate hit) pair of the mate-anchored rescue (segment_juncs.cpp:3406-3492).
// Writes the two pseudo-hit lefts (fwd, rev) or SLOT_NONE; fwd = SLOT_BREAK when the
// reference `break`s out of the mate loop at this pair.  rp = the read's planes.
// The part that does not look at the left hit: where the read's last bases (or their reverse complement) lie in the mate hit's
// flank.  A read with forty hits in its first segment and one mate hit asks forty times for the same answer: callers keep it.
THJ_HD bool rescue_scan(const Genome& g, const Params& p, const u64* rp, int W, int rl, const Hit& rh, int32_t& fwd_left, int32_t& rev_left);
THJ_HD bool rescue_pair(const Genome& g, const Params& p, const u64* rp, int W, int rl, const Hit& lh, const Hit& rh,
                        int32_t& fwd_left, int32_t& rev_left) {
    fwd_left = SLOT_NONE;
    rev_left = SLOT_NONE;
    if (lh.ref_id != rh.ref_id || hit_anti(lh) == hit_anti(rh)) return false;      // :3414
    return rescue_scan(g, p, rp, W, rl, rh, fwd_left, rev_left);
}
THJ_HD bool rescue_scan(const Genome& g, const Params& p, const u64* rp, int W, int rl, const Hit& rh, int32_t& fwd_left, int32_t& rev_left) {
    fwd_left = SLOT_NONE;
    rev_left = SLOT_NONE;
    int32_t clen = g_len(g, rh.ref_id);
    if (clen == 0) return false;
    int part = p.inner_dist_std_dev > p.inner_dist_mean ? p.inner_dist_std_dev - p.inner_dist_mean : 0;
    int flank = p.inner_dist_mean + p.inner_dist_std_dev;
    int64_t left;
    if (hit_anti(rh)) {
        if (flank <= rh.left) left = rh.left - flank; else { fwd_left = SLOT_BREAK; return false; }
    } else {
        if (part <= rh.right) left = rh.right - part; else { fwd_left = SLOT_BREAK; return false; }
    }
    int64_t fe = left + flank + part;
    if (fe > clen) fe = clen;
    int flen = (int)(fe - left);
    if (flen < 0) flen = 0;
    int cl = p.segment_length - p.segment_mismatches - 3;
    if (cl > 15) cl = 15;                                                      // :3451
    if (cl < 1 || cl > rl) return false;
    Planes fwd = r_fetch(rp, W, rl - cl, cl);     // last cl bases of the read
    Planes rev = rc_piece(fwd, cl);               // first cl bases of its reverse complement
    int fp, rvp;
    if (!flank_scan_pair(g, rh.ref_id, left, flen, fwd, rev, cl, fp, rvp)) {
        fp = flank_scan(g, rh.ref_id, left, flen, fwd, cl);
        rvp = flank_scan(g, rh.ref_id, left, flen, rev, cl);
    }
    if (fp >= 0) fwd_left = (int32_t)(left + fp);
    if (rvp >= 0) rev_left = (int32_t)(left + rvp);
    return true;                                  // the pair was scanned (what the rescue-pair statistic counts)
}

// ---- per-read view of hits_for_read -------------------------------------------
struct ReadView {
    const Hit* hits;          // batch hits
    const uint32_t* so;       // this read's nseg+1 CSR offsets
    int nseg;
    const u64* rp;            // read planes
    int W;
    int rl;                   // read length
    // mates / rescue
    const Hit* mate;          // this read's mate hits (may be null)
    int n_mate;
    const int32_t* slots;     // this read's rescue slots [n_left*n_mate*2]; null with `rescue` set = from mscan, or computed on the fly
    const int32_t* mscan;     // what rescue_scan left for each mate hit [n_mate*2] (fwd = SLOT_UNSCANNED where it returned false without a break);
                              // the scan does not look at the left hit, so a read with 40 x 80 pairs needs 80 scans, and the pair (l, m) is the
                              // contig / strand test of its two hits plus a look-up
    const Genome* lazy_g;     // genome / params for the on-the-fly rescue (only read when slots == null)
    const Params* lazy_p;
    const struct PHit* plist = nullptr;   // the rescue's pseudo-hit list itself, in the reference's order (rescue_pseudo_hits), n_plist entries:
    int n_plist = -1;                     // what the mate loops of :3406-3492 push into the last segment; -1: not built (slots / mscan / lazy)
    // derived by prepare()
    int size;                 // hits_for_read.size() after the trailing-empty trim
    bool rescue;              // segments 1.. replaced by the pseudo-hit list in `size-1`
    int check_len;
};

THJ_HD int rv_count_raw(const ReadView& v, int s) { return (int)(v.so[s + 1] - v.so[s]); }

// One pseudo-hit of the mate-anchored rescue: where the read's last bases (anti = 0) or their reverse complement (anti = 1) were
// found in a mate hit's flank.  8 bytes, so that a wave can keep a read's whole list in LDS.
struct PHit { int32_t left; uint32_t ref_anti; };          // ref_id << 1 | anti
// The pseudo-hits ONE left hit contributes, in the order the reference pushes them (segment_juncs.cpp:3409-3492): its mate loop over
// the mate hits on its contig and the opposite strand, ended by the first of them whose flank would start before the contig (`break`,
// :3431-3450); per mate hit the forward find, then the reverse one.  The scan of a mate hit's flank does not look at the left hit
// (rescue_scan), so `mscan` holds it once per mate hit.  Returns the count (and writes them when out != nullptr); `scanned` counts the
// pairs the reference maps (the rescue-pair statistic).  A read with k left hits and k mate hits has k x k pairs and up to 2 k x k
// pseudo-hits: a lane per left hit builds the list in k steps where a walk of the pairs by every caller of rv_foreach was k x k
// dependent loads -- per look at the list (round 6; a 40-copy read held its wave for a millisecond).
THJ_HD int rescue_pseudo_hits(const Hit& lh, const Hit* mate, int n_mate, const int32_t* mscan, PHit* out, int& scanned) {
    int n = 0;
    for (int m = 0; m < n_mate; ++m) {
        const Hit rh = mate[m];
        if (lh.ref_id != rh.ref_id || hit_anti(lh) == hit_anti(rh)) continue;      // :3414
        const int32_t a = mscan[2 * m], b = mscan[2 * m + 1];
        if (a == SLOT_BREAK) break;
        if (a != -3) ++scanned;                                                     // -3: SLOT_UNSCANNED
        if (a >= 0) { if (out) { out[n].left = a; out[n].ref_anti = rh.ref_id << 1; } ++n; }
        if (b >= 0) { if (out) { out[n].left = b; out[n].ref_anti = (rh.ref_id << 1) | 1u; } ++n; }
    }
    return n;
}

// Iterate the effective hit list of segment s; f(const Hit&) returns false to stop.
template <class F>
THJ_HD void rv_foreach(const ReadView& v, int s, F f) {
    if (!v.rescue || s == 0) {
        for (uint32_t k = v.so[s]; k < v.so[s + 1]; ++k) {
            Hit h = v.hits[k];
            if (!f(h)) return;
        }
        return;
    }
    if (s != v.size - 1) return;            // cleared (:3398-3401)
    if (v.n_plist >= 0) {                   // the list as built (rescue_pseudo_hits)
        for (int k = 0; k < v.n_plist; ++k) {
            const PHit ph = v.plist[k];
            Hit h; h.ref_id = ph.ref_anti >> 1; h.left = ph.left; h.right = ph.left + v.check_len;
            h.meta = 2u | (ph.ref_anti & 1u) | ((uint32_t)v.check_len << 24);
            if (!f(h)) return;
        }
        return;
    }
    int n_left = rv_count_raw(v, 0);
    for (int l = 0; l < n_left; ++l)
        for (int m = 0; m < v.n_mate; ++m) {
            int32_t a, b;
            if (v.slots) { a = v.slots[2 * (l * v.n_mate + m)]; b = v.slots[2 * (l * v.n_mate + m) + 1]; }
            else if (v.mscan) {
                const Hit lh = v.hits[v.so[0] + l], rh = v.mate[m];
                a = SLOT_NONE; b = SLOT_NONE;
                if (lh.ref_id == rh.ref_id && hit_anti(lh) != hit_anti(rh)) { a = v.mscan[2 * m]; b = v.mscan[2 * m + 1]; if (a == -3) a = SLOT_NONE; }      // -3: SLOT_UNSCANNED
            }
            else rescue_pair(*v.lazy_g, *v.lazy_p, v.rp, v.W, v.rl, v.hits[v.so[0] + l], v.mate[m], a, b);
            if (a == SLOT_BREAK) break;
            if (a >= 0) {
                Hit h; h.ref_id = v.mate[m].ref_id; h.left = a; h.right = a + v.check_len;
                h.meta = 2u | ((uint32_t)v.check_len << 24);
                if (!f(h)) return;
            }
            if (b >= 0) {
                Hit h; h.ref_id = v.mate[m].ref_id; h.left = b; h.right = b + v.check_len;
                h.meta = 3u | ((uint32_t)v.check_len << 24);
                if (!f(h)) return;
            }
        }
}

// every stride-th hit of the list, from the first-th on (a wave sharing one read's enumeration: lane, 64); f cannot stop it
template <class F>
THJ_HD void rv_foreach_strided(const ReadView& v, int s, int first, int stride, F f) {
    if (stride == 1) { rv_foreach(v, s, [&](const Hit& h) { f(h); return true; }); return; }
    if (!v.rescue || s == 0) {
        for (uint32_t k = v.so[s] + (uint32_t)first; k < v.so[s + 1]; k += (uint32_t)stride) { Hit h = v.hits[k]; f(h); }
        return;
    }
    int idx = 0;
    rv_foreach(v, s, [&](const Hit& h) { if (idx++ % stride == first) f(h); return true; });
}

THJ_HD int rv_count(const ReadView& v, int s) {
    if (v.rescue && s != 0 && v.n_plist >= 0) return s == v.size - 1 ? v.n_plist : 0;
    int n = 0;
    rv_foreach(v, s, [&](const Hit&) { ++n; return true; });
    return n;
}

// The head of find_gaps (segment_juncs.cpp:3304-3393): trailing-empty trim, the
// single-segment early return and the rescue decision.  Returns false when
// find_gaps returns before doing anything.  Does not need the rescue slots.
THJ_HD bool gaps_prepare(const Params& p, ReadView& v, bool& wants_rescue) {
    wants_rescue = false;
    v.rescue = false;
    v.check_len = p.segment_length - p.segment_mismatches - 3;
    if (v.check_len > 15) v.check_len = 15;
    if (v.nseg == 0) return false;
    int last = v.nseg - 1;
    while (last > 0 && rv_count_raw(v, last) == 0) --last;
    v.size = last + 1;
    if (last == 0) {
        if (rv_count_raw(v, 0) == 0) return false;
        Hit h0 = v.hits[v.so[0]];
        if (hit_end(h0)) return false;                                        // :3316-3318
    }
    bool check_partner = true;
    if (last != 0) {
        for (uint32_t i = v.so[0]; i < v.so[1] && check_partner; ++i) {
            Hit lh = v.hits[i];
            for (uint32_t j = v.so[last]; j < v.so[last + 1]; ++j) {
                Hit rh = v.hits[j];
                if (lh.ref_id == rh.ref_id && hit_anti(lh) == hit_anti(rh)) {
                    int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
                    if (dist >= p.min_segment_intron && dist < p.max_segment_intron) { check_partner = false; break; }
                }
            }
        }
    }
    wants_rescue = check_partner && v.n_mate > 0;
    return true;
}

// ... for a read that is known to take the rescue (it was listed because gaps_prepare said so): everything but the partner search
THJ_HD void gaps_prepare_listed(const Params& p, ReadView& v) {
    v.rescue = false;
    v.check_len = p.segment_length - p.segment_mismatches - 3;
    if (v.check_len > 15) v.check_len = 15;
    int last = v.nseg - 1;
    while (last > 0 && rv_count_raw(v, last) == 0) --last;
    v.size = last + 1;
}

// The same by a group of callers sharing one read (a wave: lane, 64): each takes every stride-th first-segment hit of the partner
// search, `any` (called once by all of them) tells whether any found one.
template <class Any>
THJ_HD bool gaps_prepare_shared(const Params& p, ReadView& v, bool& wants_rescue, int first, int stride, Any any) {
    wants_rescue = false;
    v.rescue = false;
    v.check_len = p.segment_length - p.segment_mismatches - 3;
    if (v.check_len > 15) v.check_len = 15;
    if (v.nseg == 0) return false;
    int last = v.nseg - 1;
    while (last > 0 && rv_count_raw(v, last) == 0) --last;
    v.size = last + 1;
    if (last == 0) {
        if (rv_count_raw(v, 0) == 0) return false;
        Hit h0 = v.hits[v.so[0]];
        if (hit_end(h0)) return false;                                        // :3316-3318
    }
    bool found = false;
    if (last != 0) {
        for (uint32_t i = v.so[0] + (uint32_t)first; i < v.so[1] && !found; i += (uint32_t)stride) {
            Hit lh = v.hits[i];
            for (uint32_t j = v.so[last]; j < v.so[last + 1]; ++j) {
                Hit rh = v.hits[j];
                if (lh.ref_id == rh.ref_id && hit_anti(lh) == hit_anti(rh)) {
                    int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
                    if (dist >= p.min_segment_intron && dist < p.max_segment_intron) { found = true; break; }
                }
            }
        }
    }
    wants_rescue = !any(found) && v.n_mate > 0;
    return true;
}

// Most reads are unspliced: one hit per segment, the hits abutting on one strand.  For those neither
// find_insertions_and_deletions nor find_gaps can produce a task, and unless the mate-anchored rescue applies there is
// nothing to do.  This is that test, made on a handful of hits before any of the general machinery runs; `true` is a
// promise that indels_enumerate + gaps_prepare/gaps_enumerate would emit nothing for this read.  (A read without
// any hit is trivial too.)  Anything else -- and any doubt -- returns false.
// The same decision for reads of at most NS segments as straight-line code: every test of the loops below becomes one
// term of `ok` (none of them has a side effect), so a wave runs it once instead of branching out test by test -- on the
// device the loop form cost 650 scalar instructions per wave of 64 reads, most of the main kernel's issue slots.
template <int NS>
THJ_HD bool read_is_trivial_flat(const Params& p, const ReadView& v) {
    const int nseg = v.nseg;
    const uint32_t first = v.so[0];
    if (v.so[nseg] == first) return true;                                     // no hit at all
    bool single = true;
#pragma unroll
    for (int s = 0; s < NS; ++s) single = single && (s >= nseg || v.so[s + 1] - v.so[s] == 1u);
    if (!single) return false;
    Hit h[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) if (s < nseg) h[s] = v.hits[first + s];
    const Hit h0 = h[0];
    const bool anti = hit_anti(h0);
    if (nseg == 1) return hit_end(h0) || v.n_mate == 0;
    const int L = p.segment_length;
    bool ok = true;
    Hit prev = h0;
#pragma unroll
    for (int s = 1; s < NS; ++s) {
        if (s < nseg) {
            const Hit c = h[s];
            ok = ok && c.ref_id == h0.ref_id && hit_anti(c) == anti;
            ok = ok && (anti ? c.right == prev.left : prev.right == c.left);
            if (s + 1 < nseg) {
                const int start = (s - 1) * L;
                const int plen = v.rl - start < 2 * L ? v.rl - start : 2 * L;
                const int apparent = anti ? prev.right - c.left : c.right - prev.left;
                ok = ok && start <= v.rl && apparent == plen;
            }
            prev = c;
        }
    }
    if (!ok) return false;
    const int dist = anti ? h0.left - prev.right : prev.left - h0.right;
    if (dist >= p.min_segment_intron && dist < p.max_segment_intron) return true;
    return v.n_mate == 0;
}

THJ_HD bool read_is_trivial(const Params& p, const ReadView& v) {
    if (v.nseg < 1) return true;
    if (v.nseg <= 4) return read_is_trivial_flat<4>(p, v);
    const uint32_t first = v.so[0];
    if (v.so[v.nseg] == first) return true;                                   // no hit at all
    for (int s = 0; s < v.nseg; ++s) if (v.so[s + 1] - v.so[s] != 1u) return false;
    const Hit h0 = v.hits[first];
    const bool anti = hit_anti(h0);
    if (v.nseg == 1) return hit_end(h0) || v.n_mate == 0;                     // find_gaps :3316-3318, else rescue decides
    const int L = p.segment_length;
    Hit prev = h0;
    for (int s = 1; s < v.nseg; ++s) {
        const Hit h = v.hits[first + s];
        if (h.ref_id != h0.ref_id || hit_anti(h) != anti) return false;
        if (!((anti && h.right == prev.left) || (!anti && prev.right == h.left))) return false;   // must abut (:3530-3545)
        if (s + 1 < v.nseg) {                                                 // the pair (s-1, s) of find_insertions_and_deletions
            const int start = (s - 1) * L;
            if (start > v.rl) return false;
            const int plen = v.rl - start < 2 * L ? v.rl - start : 2 * L;
            const int apparent = anti ? prev.right - h.left : h.right - prev.left;
            if (apparent != plen) return false;
        }
        prev = h;
    }
    // rescue decision of find_gaps (:3330-3393): first against last segment
    const int dist = anti ? h0.left - prev.right : prev.left - h0.right;
    if (dist >= p.min_segment_intron && dist < p.max_segment_intron) return true;
    return v.n_mate == 0;
}

// ---- reads with at most one hit per segment ("flat" reads) ------------------------------------------------------------------
// Every read of a uniquely mapping sample is one: its hit lists are registers, not lists, and find_insertions_and_deletions
// (:2807-2942), the head of find_gaps (:3304-3393) and its body (:3499-3617) become straight-line code in which every test of
// the loops is one term of a predicate -- no list walk, no divergent loop, a bounded number of tasks (NS-2 indel pairs, NS-1
// windows).  `emit.task(valid, a, b, c, d)` is called the same number of times by every caller (a wave ballots on `valid`);
// the task words are those of the kernels' queues (QueueSink in thj_segjuncs.hip).  Hit indices are batch indices: with one hit
// per segment the hit of segment s is so[s].  Must equal indels_enumerate + gaps_prepare + gaps_enumerate on such reads
// (tests/hostsim runs both).  The mate-anchored rescue (res.rescue) is left to the caller; nothing of find_gaps is emitted then.
// The first word of a queued task (the kernels' LDS queues and HBM task lists; words b..d: the contig / the two hit indices, the window's
// ends / li | ri << 16).  Reads of up to 16 segments and 512 bases: a window's support read starts below 512 (9 bits) and is at most
// L + 16 <= 80 bases long (7 bits); an indel pair is one of 14 (4 bits) and its read piece at most 2L <= 128 bases (8 bits).
THJ_HD uint32_t task_window_word(bool anti, int start, int slen) { return (anti ? 1u << 9 : 0u) | ((uint32_t)(start & 511) << 10) | ((uint32_t)(slen & 127) << 19); }
THJ_HD uint32_t task_indel_word(bool anti, bool is_del, int i, int plen) {
    return (1u << 8) | (anti ? 1u << 9 : 0u) | (is_del ? 1u << 10 : 0u) | ((uint32_t)(i & 15) << 11) | ((uint32_t)(plen & 255) << 15);
}
THJ_HD bool task_is_indel(uint32_t a) { return (a & (1u << 8)) != 0; }
THJ_HD bool task_anti(uint32_t a) { return ((a >> 9) & 1u) != 0; }
THJ_HD bool task_is_del(uint32_t a) { return ((a >> 10) & 1u) != 0; }
THJ_HD int task_indel_i(uint32_t a) { return (int)((a >> 11) & 15u); }
THJ_HD int task_indel_plen(uint32_t a) { return (int)((a >> 15) & 255u); }
THJ_HD int task_window_start(uint32_t a) { return (int)((a >> 10) & 511u); }
THJ_HD int task_window_slen(uint32_t a) { return (int)((a >> 19) & 127u); }

struct FlatResult { bool rescue; int size; int n_windows, n_indels; };

template <int NS, class Emit>
THJ_HD FlatResult flat_read(const Params& p, int nseg, const uint32_t (&so)[NS + 1], const Hit (&h)[NS], int rl, int n_mate, Emit& emit) {
    FlatResult res{false, 0, 0, 0};
    const int L = p.segment_length;
    uint32_t present = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) if (s < nseg && so[s + 1] != so[s]) present |= 1u << s;
    // find_insertions_and_deletions: the pairs (i, i+1), i + 2 < nseg; an empty list or a start past the read ends the function
    bool go = true;
#pragma unroll
    for (int i = 0; i + 2 < NS; ++i) {
        const int start = i * L;
        go = go && i + 2 < nseg && ((present >> i) & 3u) == 3u && start <= rl;
        const int plen = rl - start < 2 * L ? rl - start : 2 * L;
        const Hit lh = h[i], rh = h[i + 1];
        const bool anti = hit_anti(lh);
        const int apparent = anti ? lh.right - rh.left : rh.right - lh.left;
        const int disc = apparent - plen;
        const bool is_del = disc > 0 && disc <= p.max_deletion_length;
        const bool is_ins = disc < 0 && disc >= -p.max_insertion_length;
        const bool ok = go && lh.ref_id == rh.ref_id && anti == hit_anti(rh) && (is_del || is_ins);
        emit.task(ok, task_indel_word(anti, is_del, i, plen), anti ? so[i + 1] : so[i], anti ? so[i] : so[i + 1], 0u);
        res.n_indels += ok ? 1 : 0;
    }
    // the head of find_gaps: trailing-empty trim, the single-segment return, the partner test of first against last segment
    int last = 0;
    Hit hl = h[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) if ((present >> s) & 1u) { last = s; hl = h[s]; }
    res.size = last + 1;
    bool run = nseg > 0;
    if (last == 0) run = run && (present & 1u) && !hit_end(h[0]);
    bool check_partner = true;
    if (last != 0 && (present & 1u) && h[0].ref_id == hl.ref_id && hit_anti(h[0]) == hit_anti(hl)) {
        const int dist = hit_anti(h[0]) ? h[0].left - hl.right : hl.left - h[0].right;
        if (dist >= p.min_segment_intron && dist < p.max_segment_intron) check_partner = false;
    }
    res.rescue = run && check_partner && n_mate > 0;
    bool en = run && !res.rescue;
    if (p.bowtie2 && present != 0 && 1 > p.max_seg_multihits) en = false;        // :3499-3506
    // the body: segment s against s+1 (an abutting hit ends it, one at intron distance gives a 16-base window) and s+2 (L+16 bases)
#pragma unroll
    for (int s = 0; s + 1 < NS; ++s) {
        const bool has = en && s != last && ((present >> s) & 1u);
        const Hit bh = h[s];
        const bool banti = hit_anti(bh);
        const Hit rh = h[s + 1];
        const bool c1 = ((present >> (s + 1)) & 1u) && banti == hit_anti(rh) && bh.ref_id == rh.ref_id;
        const bool abut = banti ? rh.right == bh.left : bh.right == rh.left;
        const int d1 = banti ? bh.left - rh.right : rh.left - bh.right;
        const bool found = c1 && abut;
        const bool drs = c1 && !abut && d1 >= p.min_segment_intron && d1 < p.max_segment_intron;
        bool rrs = false;
        Hit d = rh;
        if (s + 2 < NS) {
            const Hit rrh = h[s + 2];
            const bool c2 = ((present >> (s + 2)) & 1u) && banti == hit_anti(rrh) && bh.ref_id == rrh.ref_id;
            const int d2 = banti ? bh.left - rrh.right : rrh.left - bh.right;
            rrs = !found && c2 && d2 >= p.min_segment_intron + L && d2 < p.max_segment_intron + L;
            if (rrs) d = rrh;
        }
        const int start = (s + 1) * L - 8;                                      // :3583-3586
        int slen = rrs ? L + 16 : 16;
        if (slen > rl - start) slen = rl - start;
        int32_t wl, wr;
        if (!banti) { wl = bh.right - 8; if (wl < 0) wl = 0; wr = d.left + 8; }   // :3589-3594
        else { wl = d.right - 8; wr = bh.left + 8; }                              // :3596-3604
        const bool ok = has && !found && (drs || rrs) && start >= 0 && slen >= 0;
        emit.task(ok, task_window_word(banti, start, slen), bh.ref_id, (uint32_t)wl, (uint32_t)wr);
        res.n_windows += ok ? 1 : 0;
    }
    return res;
}

// The mate-anchored rescue (:3330-3497) of a flat read whose mate has at most MAXM hits, after the flank scans: `sc` holds what
// rescue_scan left for each mate hit (fwd, rev; fwd == SLOT_UNSCANNED where it returned false without asking for the break).
// The pseudo-hits stand in the read's last kept segment (size - 1) and every segment between is cleared, so only the first
// segment's hit bh can see them: as its s+1 list (size 2, 16-base windows, an abutting pseudo-hit ends it) or as its s+2 list
// (size 3, L+16 bases).  Returns the number of scanned pairs (the rescue-pair statistic).  Must equal gaps_enumerate on the
// rescue view (rv_foreach) for such reads.
enum { SLOT_UNSCANNED = -3 };
template <int MAXM, class Emit>
THJ_HD int flat_rescue(const Params& p, bool has0, const Hit& bh, int size, int rl, const Hit (&mh)[MAXM], int n_mate,
                       const int32_t (&sc)[2 * MAXM], Emit& emit, int& n_windows) {
    const int L = p.segment_length;
    int cl = L - p.segment_mismatches - 3;
    if (cl > 15) cl = 15;
    const bool banti = hit_anti(bh);
    bool alive = has0;
    int pairs = 0, n_pseudo = 0;
    int32_t pl[2 * MAXM];
    bool pv[2 * MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        const bool compat = alive && m < n_mate && bh.ref_id == mh[m].ref_id && banti != hit_anti(mh[m]);     // :3414
        const int32_t f = compat ? sc[2 * m] : SLOT_NONE, rv = compat ? sc[2 * m + 1] : SLOT_NONE;
        if (compat && f == SLOT_BREAK) alive = false;                             // the reference leaves the mate loop (:3431-3450)
        pairs += (compat && f != SLOT_BREAK && f != SLOT_UNSCANNED) ? 1 : 0;
        pv[2 * m] = compat && alive && f >= 0; pl[2 * m] = f;
        pv[2 * m + 1] = compat && alive && rv >= 0; pl[2 * m + 1] = rv;
        n_pseudo += (pv[2 * m] ? 1 : 0) + (pv[2 * m + 1] ? 1 : 0);
    }
    bool en = has0 && (size == 2 || size == 3);
    if (p.bowtie2 && (1 > p.max_seg_multihits || n_pseudo > p.max_seg_multihits)) en = false;
    const bool far = size == 3;
    const int lo = p.min_segment_intron + (far ? L : 0), hi = p.max_segment_intron + (far ? L : 0);
    bool found = false, inr[2 * MAXM];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 2 * MAXM; ++k) {
        const bool panti = (k & 1) != 0;                                          // fwd pseudo-hit: sense, rev: antisense
        const bool c = pv[k] && banti == panti;
        const int32_t pleft = pl[k], pright = pl[k] + cl;
        const bool abut = banti ? pright == bh.left : bh.right == pleft;
        if (c && abut && !far) found = true;
        const int dist = banti ? bh.left - pright : pleft - bh.right;
        inr[k] = c && dist >= lo && dist < hi;
        n += inr[k] ? 1 : 0;
    }
    const int start = L - 8;
    int slen = far ? L + 16 : 16;
    if (slen > rl - start) slen = rl - start;
    const bool ok_all = en && !found && n > 0 && start >= 0 && slen >= 0;
#pragma unroll
    for (int k = 0; k < 2 * MAXM; ++k) {
        const int32_t pleft = pl[k], pright = pl[k] + cl;
        int32_t wl, wr;
        if (!banti) { wl = bh.right - 8; if (wl < 0) wl = 0; wr = pleft + 8; }
        else { wl = pright - 8; wr = bh.left + 8; }
        const bool ok = ok_all && inr[k];
        emit.task(ok, task_window_word(banti, start, slen), bh.ref_id, (uint32_t)wl, (uint32_t)wr);
        n_windows += ok ? 1 : 0;
    }
    return pairs;
}

// The body of find_gaps after the rescue (segment_juncs.cpp:3499-3617).
// Sink: window(ref, wl, wr, antisense, support_start, support_len).
template <class Sink>
THJ_HD void gaps_enumerate(const Params& p, const ReadView& v, Sink& sink, int first = 0, int stride = 1) {
    const int L = p.segment_length;
    if (p.bowtie2)                                                            // :3499-3506
        for (int s = 0; s < v.size; ++s) {
            int n = (!v.rescue || s == 0) ? rv_count_raw(v, s) : rv_count(v, s);
            if (n > p.max_seg_multihits) return;
        }
    // (first, stride): the hits `bh` this caller takes -- every hit is handled on its own, so a wave can share a read with many
    for (int s = 0; s < v.size; ++s) {
        rv_foreach_strided(v, s, first, stride, [&](const Hit& bh) {
            bool found = (s == v.size - 1);
            int n_drs = 0, n_rrs = 0;
            const bool banti = hit_anti(bh);
            if (s < v.size - 1) {
                rv_foreach(v, s + 1, [&](const Hit& rh) {
                    if (banti != hit_anti(rh) || bh.ref_id != rh.ref_id) return true;
                    if ((banti && rh.right == bh.left) || (!banti && bh.right == rh.left)) { found = true; return false; }
                    int dist = banti ? bh.left - rh.right : rh.left - bh.right;
                    if (dist >= p.min_segment_intron && dist < p.max_segment_intron) ++n_drs;
                    return true;
                });
            }
            if (!found && s < v.size - 2) {
                rv_foreach(v, s + 2, [&](const Hit& rrh) {
                    if (banti != hit_anti(rrh) || bh.ref_id != rrh.ref_id) return true;
                    int dist = banti ? bh.left - rrh.right : rrh.left - bh.right;
                    if (dist >= p.min_segment_intron + L && dist < p.max_segment_intron + L) ++n_rrs;
                    return true;
                });
            }
            if (!found && (n_drs > 0 || n_rrs > 0)) {
                const bool use_rrs = n_rrs > 0;
                const int lo = p.min_segment_intron + (use_rrs ? L : 0);
                const int hi = p.max_segment_intron + (use_rrs ? L : 0);
                const int start = (s + 1) * L - 8;                            // :3583-3586
                int slen = use_rrs ? L + 16 : 16;
                if (slen > v.rl - start) slen = v.rl - start;
                if (start >= 0 && slen >= 0) {
                    rv_foreach(v, use_rrs ? s + 2 : s + 1, [&](const Hit& d) {
                        if (banti != hit_anti(d) || bh.ref_id != d.ref_id) return true;
                        int dist = banti ? bh.left - d.right : d.left - bh.right;
                        if (dist < lo || dist >= hi) return true;
                        int32_t wl, wr;
                        if (!banti) { wl = bh.right - 8; if (wl < 0) wl = 0; wr = d.left + 8; }   // :3589-3594
                        else { wl = d.right - 8; wr = bh.left + 8; }                              // :3596-3604
                        sink.window(bh.ref_id, wl, wr, banti, start, slen);
                        return true;
                    });
                }
            }
        });
    }
}

// ---- a read with several hits a segment, by ONE WAVE with the hits in registers (round 6) -----------------------------------------
// The head of find_gaps (:3304-3393), find_insertions_and_deletions' pair enumeration (:2807-2942) and find_gaps' body (:3499-3617) for
// a read whose segments hold at most 64 hits each: lane j keeps hit j of the segment it enumerates FROM in registers, and every
// sweep over the hits of another segment is a loop of broadcasts (x.bcast: one v_readlane per word, the loop counter is uniform) --
// no memory on the way.  gaps_prepare_shared / indels_enumerate / gaps_enumerate with (lane, 64) do the same by walking the staged
// hit records: one dependent LDS round trip per step, eight sweeps a read, ~25 us for a read of 25 copies (0.25 ms per launch of
// thj_k_segjuncs_shared on the mix for 15 000 reads).  Same tasks, same counts; their order in the queue differs (it never mattered).
// X: lane, ballot(bool), bcast(uint32_t, int src).  v.hits / v.so: the read's hits (any memory; read once per lane and segment).
// Returns what gaps_prepare returns; with wants_rescue set nothing has been enumerated (the rescue kernels take the read, its
// indel pairs included).
template <class X>
THJ_HD Hit wave_bcast_hit(X& x, const Hit& h, int src) {
    Hit o;
    o.ref_id = x.bcast(h.ref_id, src); o.left = (int32_t)x.bcast((uint32_t)h.left, src);
    o.right = (int32_t)x.bcast((uint32_t)h.right, src); o.meta = x.bcast(h.meta, src);
    return o;
}
THJ_HD Hit wave_lane_hit(const ReadView& v, int s, int lane) {
    Hit h{0, 0, 0, 0};
    if (lane < rv_count_raw(v, s)) h = v.hits[v.so[s] + (uint32_t)lane];
    return h;
}
THJ_HD bool wave_read_fits(const ReadView& v) {
    for (int s = 0; s < v.nseg; ++s) if (rv_count_raw(v, s) > 64) return false;
    return true;
}
template <class X, class Sink>
THJ_HD bool wave_read_enumerate(X& x, const Params& p, ReadView& v, Sink& sink, bool& wants_rescue) {
    const int lane = x.lane;
    const int L = p.segment_length;
    wants_rescue = false;
    v.rescue = false;
    v.check_len = p.segment_length - p.segment_mismatches - 3;
    if (v.check_len > 15) v.check_len = 15;
    if (v.nseg == 0) return false;
    int last = v.nseg - 1;
    while (last > 0 && rv_count_raw(v, last) == 0) --last;
    v.size = last + 1;
    if (last == 0) {
        if (rv_count_raw(v, 0) == 0) return false;
        const Hit h0 = v.hits[v.so[0]];
        if (hit_end(h0)) return false;                                        // :3316-3318
    }
    // ---- the partner search (:3361-3390): a first-segment hit with a last-segment hit at intron distance
    {
        bool found = false;
        if (last != 0) {
            const Hit lh = wave_lane_hit(v, 0, lane);
            const bool have = lane < rv_count_raw(v, 0);
            const Hit D = wave_lane_hit(v, last, lane);
            const int nD = rv_count_raw(v, last);
            for (int j = 0; j < nD; ++j) {
                const Hit rh = wave_bcast_hit(x, D, j);
                if (have && lh.ref_id == rh.ref_id && hit_anti(lh) == hit_anti(rh)) {
                    const int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
                    if (dist >= p.min_segment_intron && dist < p.max_segment_intron) found = true;
                }
            }
        }
        wants_rescue = x.ballot(found) == 0ull && v.n_mate > 0;
    }
    if (wants_rescue) return true;
    if (THJ_EXPF(1 << 27)) return true;
    // ---- find_insertions_and_deletions: the pairs (hit of segment i, hit of segment i + 1), :2856-2940
    if (v.nseg >= 2 && !THJ_EXPF(1 << 17))
        for (int i = 0; i + 2 < v.nseg; ++i) {
            const uint32_t lb = v.so[i], le = v.so[i + 1], re = v.so[i + 2];
            if (lb == le || le == re) break;                                  // :2869-2870
            const int start = i * L;
            if (start > v.rl) break;
            const int plen = v.rl - start < 2 * L ? v.rl - start : 2 * L;
            const Hit lh = wave_lane_hit(v, i, lane);
            const bool have = (uint32_t)lane < le - lb;
            const Hit C = wave_lane_hit(v, i + 1, lane);
            const int nC = (int)(re - le);
            for (int ri = 0; ri < nC; ++ri) {
                const Hit rh = wave_bcast_hit(x, C, ri);
                if (!have || lh.ref_id != rh.ref_id) continue;
                const bool anti = hit_anti(lh);
                if (anti != hit_anti(rh)) continue;
                const int apparent = anti ? lh.right - rh.left : rh.right - lh.left;
                const int disc = apparent - plen;
                const bool is_del = disc > 0 && disc <= p.max_deletion_length;
                const bool is_ins = disc < 0 && disc >= -p.max_insertion_length;
                if (is_del || is_ins)
                    sink.indel(i, anti ? le + (uint32_t)ri : lb + (uint32_t)lane, anti ? lb + (uint32_t)lane : le + (uint32_t)ri, lane, ri, anti, plen, is_del);
            }
        }
    // ---- find_gaps' body (:3499-3617)
    if (THJ_EXPF(1 << 25)) return true;
    if (p.bowtie2)
        for (int s = 0; s < v.size; ++s) if (rv_count_raw(v, s) > p.max_seg_multihits) return true;      // :3499-3506
    for (int s = 0; s + 1 < v.size; ++s) {                  // (a hit of the last segment has nothing to its right: `found` from the start)
        const Hit bh = wave_lane_hit(v, s, lane);
        const bool have = lane < rv_count_raw(v, s);
        const bool banti = hit_anti(bh);
        const Hit B = wave_lane_hit(v, s + 1, lane);
        const int nB = rv_count_raw(v, s + 1);
        bool found = false;
        int n_drs = 0, n_rrs = 0;
        for (int c = 0; c < nB; ++c) {
            const Hit rh = wave_bcast_hit(x, B, c);
            if (!have || banti != hit_anti(rh) || bh.ref_id != rh.ref_id) continue;
            if ((banti && rh.right == bh.left) || (!banti && bh.right == rh.left)) found = true;
            const int dist = banti ? bh.left - rh.right : rh.left - bh.right;
            if (dist >= p.min_segment_intron && dist < p.max_segment_intron) ++n_drs;
        }
        Hit C{0, 0, 0, 0};
        int nC = 0;
        if (s < v.size - 2) {
            C = wave_lane_hit(v, s + 2, lane);
            nC = rv_count_raw(v, s + 2);
            if (x.ballot(have && !found) != 0ull)
                for (int c = 0; c < nC; ++c) {
                    const Hit rrh = wave_bcast_hit(x, C, c);
                    if (!have || found || banti != hit_anti(rrh) || bh.ref_id != rrh.ref_id) continue;
                    const int dist = banti ? bh.left - rrh.right : rrh.left - bh.right;
                    if (dist >= p.min_segment_intron + L && dist < p.max_segment_intron + L) ++n_rrs;
                }
        }
        const bool emits = have && !found && (n_drs > 0 || n_rrs > 0);
        const bool use_rrs = n_rrs > 0;
        const int lo = p.min_segment_intron + (use_rrs ? L : 0), hi = p.max_segment_intron + (use_rrs ? L : 0);
        const int start = (s + 1) * L - 8;                                    // :3583-3586
        int slen = use_rrs ? L + 16 : 16;
        if (slen > v.rl - start) slen = v.rl - start;
        const bool go = emits && start >= 0 && slen >= 0;
        // the windows: against the next segment's hits for the lanes that count there, the one after for the others
        for (int pass = 0; pass < 2; ++pass) {
            const bool mine = go && (use_rrs ? pass == 1 : pass == 0);
            if (x.ballot(mine) == 0ull) continue;
            const int nT = pass == 0 ? nB : nC;
            for (int c = 0; c < nT; ++c) {
                const Hit d = wave_bcast_hit(x, pass == 0 ? B : C, c);
                if (!mine || banti != hit_anti(d) || bh.ref_id != d.ref_id) continue;
                const int dist = banti ? bh.left - d.right : d.left - bh.right;
                if (dist < lo || dist >= hi) continue;
                int32_t wl, wr;
                if (!banti) { wl = bh.right - 8; if (wl < 0) wl = 0; wr = d.left + 8; }   // :3589-3594
                else { wl = d.right - 8; wr = bh.left + 8; }                              // :3596-3604
                sink.window(bh.ref_id, wl, wr, banti, start, slen);
            }
        }
    }
    return true;
}

// find_insertions_and_deletions (segment_juncs.cpp:2807-2942) pair enumeration.
// Sink: indel(i, left_idx, right_idx, li, ri, antisense, plen, is_deletion)
// with left/right already swapped for antisense pairs (:2914-2920).
template <class Sink>
THJ_HD void indels_enumerate(const Params& p, const ReadView& v, Sink& sink, int first = 0, int stride = 1) {
    const int L = p.segment_length;
    if (v.nseg < 2) return;
    for (int i = 0; i + 2 < v.nseg; ++i) {                                    // :2856
        uint32_t lb = v.so[i], le = v.so[i + 1], re = v.so[i + 2];
        if (lb == le || le == re) return;                                     // :2869-2870
        int start = i * L;
        if (start > v.rl) return;
        int plen = v.rl - start < 2 * L ? v.rl - start : 2 * L;
        for (uint32_t li = lb + (uint32_t)first; li < le; li += (uint32_t)stride) {
            Hit lh = v.hits[li];
            for (uint32_t ri = le; ri < re; ++ri) {
                Hit rh = v.hits[ri];
                if (lh.ref_id != rh.ref_id) continue;
                bool anti = hit_anti(lh);
                if (anti != hit_anti(rh)) continue;
                int apparent = anti ? lh.right - rh.left : rh.right - lh.left;
                int disc = apparent - plen;
                bool is_del = disc > 0 && disc <= p.max_deletion_length;
                bool is_ins = disc < 0 && disc >= -p.max_insertion_length;
                if (is_del || is_ins)
                    sink.indel(i, anti ? ri : li, anti ? li : ri, (int)(li - lb), (int)(ri - le), anti, plen, is_del);
            }
        }
    }
}

// Execute one indel task: fetch the 2L read piece, rc when antisense, run the detector.
// WIDE = false compiles the 64-bit path only (segment_length <= 32: the default, and what keeps the kernels lean)
template <bool WIDE = true, class Sink>
THJ_HD void indel_exec(const Genome& g, const Params& p, const ReadView& v, int i, uint32_t lidx, uint32_t ridx,
                       bool anti, int plen, bool is_del, u64 prio, Sink& sink) {
    const Hit lh = v.hits[lidx], rh = v.hits[ridx];
    if (!WIDE || plen <= 64) {
        if (plen > 64) return;
        PlanesT<u64> rd = Fetch<u64>::read(v.rp, v.W, i * p.segment_length, plen);
        if (anti) rd = Fetch<u64>::rc(rd, plen);
        if (is_del) small_deletion<u64>(g, rd, plen, lh, rh, sink);
        else small_insertion<u64>(g, rd, plen, lh, rh, prio, sink);
    } else if (WIDE) {                            // segment_length > 32
        PlanesT<u128> rd = Fetch<u128>::read(v.rp, v.W, i * p.segment_length, plen);
        if (anti) rd = Fetch<u128>::rc(rd, plen);
        if (is_del) small_deletion<u128>(g, rd, plen, lh, rh, sink);
        else small_insertion<u128>(g, rd, plen, lh, rh, prio, sink);
    }
}

// Execute one window task.
template <bool WIDE = true, class Sink>
THJ_HD void window_exec(const Genome& g, const Params& p, const ReadView& v, uint32_t ref_id, int32_t wl, int32_t wr,
                        bool anti, int start, int slen, Sink& sink) {
    if (slen < 2) return;
    if (!WIDE || slen <= 62) {
        if (slen > 62) return;
        PlanesT<u64> sup = Fetch<u64>::read(v.rp, v.W, start, slen);
        if (anti) sup = Fetch<u64>::rc(sup, slen);                             // :3598-3601
        window_scan<u64>(g, p, ref_id, wl, wr, anti, sup, slen, sink);
    } else if (WIDE) {                            // L + 16 support read with segment_length > 46
        PlanesT<u128> sup = Fetch<u128>::read(v.rp, v.W, start, slen);
        if (anti) sup = Fetch<u128>::rc(sup, slen);
        window_scan<u128>(g, p, ref_id, wl, wr, anti, sup, slen, sink);
    }
}

// insertion priority = visiting order inside one batch: read ordinal (< 2^29), segment pair (< 16), li, ri: 45 bits.  In the
// insertion table it sits above the inserted bases: value = prio << INS_SEQ_BITS | bases (3 bits each, at most 6 of them).
static constexpr int INS_SEQ_BITS = 18;
THJ_HD u64 ins_prio(uint32_t ordinal, int i, int li, int ri) {
    if (li > 63) li = 63;     // bowtie2 runs with -k 41 (tophat.py:2294): never reached in practice
    if (ri > 63) ri = 63;
    return ((u64)ordinal << 16) | ((u64)(i & 15) << 12) | ((u64)li << 6) | (u64)ri;
}


// ---- fusion search: find_fusions + detect_fusion (segment_juncs.cpp:2976-3291, :2629-2805) -------------------
enum { FUS_FF = 7, FUS_FR = 8, FUS_RF = 9, FUS_RR = 10 };

// `len` (<= 64) bases of the whole read (forward, or its reverse complement) starting at `off`
THJ_HD Planes read_piece(const u64* rp, int W, int rl, bool rc, int off, int len) {
    if (!rc) return r_fetch(rp, W, off, len);
    return rc_piece(r_fetch(rp, W, rl - off - len, len), len);
}
// genomic string of detect_fusion: ref[start, start+rl) forward, or reverse-complemented (then piece `off` of the
// string comes from the far end)
THJ_HD Planes genomic_piece(const Genome& g, uint32_t ref, int64_t start, int rl, bool rc, int off, int len) {
    if (!rc) return g_fetch(g, ref, start + off);
    Planes p = g_fetch(g, ref, start + rl - off - len);
    return rc_piece(p, len);
}

// detect_fusion with the whole-read simpleSplitAlignment (all tied best positions), in two steps so that a wave can reserve the
// room for all its lanes' events with one atomic: fusion_eval -> how many events the pair gives (0: none), fusion_emit -> the
// k-th of them to f(k, ref1, ref2, left, right, dir, ed).
// e(q) = errors of the left hit's string before q + errors of the right hit's string from q on, q = 1..rl-1; the events are the
// q with the smallest e, unless that exceeds the hits' edit distances or 2, or one of them leaves a side shorter than
// fusion_anchor_length (:2697-2713).  One walk over the two mismatch masks finds the minimum, how often it is reached and
// the minimum over the forbidden ends; a second walk, only for the pairs that pass, emits.
struct FusEval { u64 mL[4], mR[4]; int e1, min_err, total_ed; bool lrc, rrc; };
THJ_HD int fusion_eval(const Genome& g, const Params& p, const u64* rp, int W, int rl, bool read_rc, const Hit& lh, const Hit& rh, int dir, FusEval& ev) {
    const int32_t llen = g_len(g, lh.ref_id), rlen = g_len(g, rh.ref_id);
    if (llen == 0 || rlen == 0 || rl > 256) return 0;
    int64_t lstart, rstart;
    const bool lrc = !(dir == FUS_FF || dir == FUS_FR), rrc = !(dir == FUS_FF || dir == FUS_RF);
    ev.lrc = lrc; ev.rrc = rrc;
    if (!lrc) { if (lh.left + rl > llen || lh.left < 0) return 0; lstart = lh.left; }
    else { if (lh.right < rl || lh.right > llen) return 0; lstart = (int64_t)lh.right - rl; }
    if (!rrc) { if (rh.right < rl || rh.right > rlen) return 0; rstart = (int64_t)rh.right - rl; }
    else { if (rh.left + rl > rlen || rh.left < 0) return 0; rstart = rh.left; }
    int tot_r = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        ev.mL[w] = 0; ev.mR[w] = 0;
        int off = w * 64;
        if (off < rl) {
            int l = rl - off < 64 ? rl - off : 64;
            Planes rd = read_piece(rp, W, rl, read_rc, off, l);
            Planes a = genomic_piece(g, lh.ref_id, lstart, rl, lrc, off, l);
            Planes b = genomic_piece(g, rh.ref_id, rstart, rl, rrc, off, l);
            u64 M = lowmask(l);
            ev.mL[w] = ((a.lo ^ rd.lo) | (a.hi ^ rd.hi) | a.nm | rd.nm) & M;     // 'N' on either side is an error (:2416-2418)
            ev.mR[w] = ((b.lo ^ rd.lo) | (b.hi ^ rd.hi) | b.nm | rd.nm) & M;
            tot_r += popc(ev.mR[w]);
        }
    }
    // e(1) = bitL(0) + tot_r - bitR(0); e(q + 1) = e(q) + bitL(q) - bitR(q)
    ev.e1 = (int)(ev.mL[0] & 1ull) + tot_r - (int)(ev.mR[0] & 1ull);
    ev.total_ed = hit_ed(lh) + hit_ed(rh);
    const int A = p.fusion_anchor_length;
    int mn = rl + 1, cnt = 0, mn_ends = rl + 1, ee = ev.e1;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int q0 = w == 0 ? 1 : w * 64, q1 = rl < (w + 1) * 64 ? rl : (w + 1) * 64;
        u64 l = ev.mL[w] >> (q0 & 63), r = ev.mR[w] >> (q0 & 63);
        for (int q = q0; q < q1; ++q) {
            if (ee < mn) { mn = ee; cnt = 1; } else if (ee == mn) ++cnt;
            if ((q < A || rl - q < A) && ee < mn_ends) mn_ends = ee;
            ee += (int)(l & 1ull) - (int)(r & 1ull);
            l >>= 1; r >>= 1;
        }
    }
    ev.min_err = mn;
    if (mn > ev.total_ed || mn > 2 || mn_ends == mn) return 0;
    return cnt;
}
template <class F>
THJ_HD void fusion_emit(const FusEval& ev, int rl, const Hit& lh, const Hit& rh, int dir, F f) {
    int ee = ev.e1, k = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int q0 = w == 0 ? 1 : w * 64, q1 = rl < (w + 1) * 64 ? rl : (w + 1) * 64;
        u64 l = ev.mL[w] >> (q0 & 63), r = ev.mR[w] >> (q0 & 63);
        for (int q = q0; q < q1; ++q) {
            if (ee == ev.min_err) {
                uint32_t left = !ev.lrc ? (uint32_t)(lh.left + q - 1) : (uint32_t)(lh.right - q);
                uint32_t right = !ev.rrc ? (uint32_t)(rh.right - (rl - q)) : (uint32_t)(rh.left + (rl - q) - 1);
                uint32_t r1 = lh.ref_id, r2 = rh.ref_id; int tdir = dir;
                if (r2 < r1 || (r1 == r2 && left > right)) {                       // :2776-2789
                    uint32_t t = r1; r1 = r2; r2 = t;
                    t = left; left = right; right = t;
                    if (dir == FUS_FF) tdir = FUS_RR;
                }
                f(k++, r1, r2, left, right, (uint32_t)tdir, (uint32_t)ev.total_ed);
            }
            ee += (int)(l & 1ull) - (int)(r & 1ull);
            l >>= 1; r >>= 1;
        }
    }
}
template <class Sink>
THJ_HD void detect_fusion(const Genome& g, const Params& p, const u64* rp, int W, int rl, bool read_rc, const Hit& lh, const Hit& rh,
                          int dir, Sink& sink) {
    FusEval ev;
    if (fusion_eval(g, p, rp, W, rl, read_rc, lh, rh, dir, ev) == 0) return;
    fusion_emit(ev, rl, lh, rh, dir, [&](int, uint32_t r1, uint32_t r2, uint32_t left, uint32_t right, uint32_t tdir, uint32_t ed) { sink.fusion(r1, r2, left, right, tdir, ed); });
}

// DEFER: the pair is handed to sink.defer() instead of running detect_fusion here (the device kernel queues such pairs -- a few
// per hundred reads -- and runs them densely; see thj_k_fusion)
template <bool DEFER = false, class Sink>
THJ_HD void fusion_pair(const Genome& g, const Params& p, const u64* rp, int W, int rl, Hit lh, Hit rh, Sink& sink) {
    if (sink.ignored(lh.ref_id) || sink.ignored(rh.ref_id)) return;          // --fusion-ignore-chromosomes (:3214-3231)
    if (p.bowtie2 && hit_ed(lh) + hit_ed(rh) > (p.segment_mismatches << 1)) return;      // :3222-3226
    const int minus_dist = -p.max_insertion_length * 2;
    if (lh.ref_id == rh.ref_id && hit_anti(lh) == hit_anti(rh)) {
        int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
        if (dist > minus_dist && dist <= p.fusion_min_dist) return;
    }
    int dir = FUS_FF;
    bool rc = false;
    if (hit_anti(lh) == hit_anti(rh)) {
        if (hit_anti(lh)) { Hit t = lh; lh = rh; rh = t; rc = true; }           // :3266-3274
    } else if (!hit_anti(lh) && hit_anti(rh)) dir = FUS_FR;
    else dir = FUS_RF;
    if constexpr (DEFER) sink.defer(rc, lh, rh, dir);
    else detect_fusion(g, p, rp, W, rl, rc, lh, rh, dir, sink);
}

// find_fusions for one read (all visited reads, incl. top == 0), in two parts: the pairs of real hits -- returns whether the
// read also wants the mate-anchored part -- and that part.  (The device kernel runs the second part densely over a queue of
// such reads: it holds the flank scans.)
template <bool DEFER = false, class Sink>
THJ_HD bool fusion_read_pairs(const Genome& g, const Params& p, const ReadView& v, Sink& sink) {
    if (v.nseg == 0) return false;
    int last = v.nseg - 1;
    while (last > 0 && rv_count_raw(v, last) == 0) --last;
    const uint32_t l0 = v.so[0], l1 = v.so[1];
    if (last == 0 && (l0 == l1 || hit_end(v.hits[l0]))) return false;           // :3035-3037
    const uint32_t r0 = last != 0 ? v.so[last] : 0, r1 = last != 0 ? v.so[last + 1] : 0;   // right_segment_hits (:3075-3080)
    bool check_partner = true;
    if (last != 0) {
        for (uint32_t i = l0; i < l1 && check_partner; ++i) {
            Hit lh = v.hits[i];
            for (uint32_t j = r0; j < r1; ++j) {
                Hit rh = v.hits[j];
                if (lh.ref_id == rh.ref_id && hit_anti(lh) == hit_anti(rh)) {
                    int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
                    if (dist > -p.max_insertion_length && dist <= p.fusion_min_dist) { check_partner = false; break; }
                }
            }
        }
    }
    // pairs with the real hits of the last segment
    for (uint32_t i = l0; i < l1; ++i)
        for (uint32_t j = r0; j < r1; ++j) fusion_pair<DEFER>(g, p, v.rp, v.W, v.rl, v.hits[i], v.hits[j], sink);
    return check_partner && v.n_mate > 0 && l1 > l0;
}
// mate-anchored pseudo-hits (:3117-3202): every one of them is then paired with every left hit.  Where the read's last bases lie
// in a mate hit's flank does not depend on the left hit (rescue_scan, the same scan as find_gaps' rescue): kept per mate hit
// for the first two.
template <bool DEFER = false, class Sink>
THJ_HD void fusion_read_mates(const Genome& g, const Params& p, const ReadView& v, Sink& sink) {
    const uint32_t l0 = v.so[0], l1 = v.so[1];
    const int minus_dist = -p.max_insertion_length * 2;
    int cl = p.segment_length - p.segment_mismatches - 3; if (cl > 15) cl = 15;
    int32_t c0_f = SLOT_NONE, c0_r = SLOT_NONE, c1_f = SLOT_NONE, c1_r = SLOT_NONE;
    bool c0_ok = false, c1_ok = false;
    uint32_t have = 0;
    for (uint32_t l = l0; l < l1; ++l) {
        Hit lh = v.hits[l];
        for (int m = 0; m < v.n_mate; ++m) {
            Hit rh = v.mate[m];
            if (lh.ref_id == rh.ref_id && hit_anti(lh) != hit_anti(rh)) {
                int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
                if (dist > minus_dist && dist <= p.fusion_min_dist) continue;
            }
            int32_t f, rv; bool scanned;
            if (m == 0) { if (!(have & 1u)) { c0_ok = rescue_scan(g, p, v.rp, v.W, v.rl, rh, c0_f, c0_r); have |= 1u; } f = c0_f; rv = c0_r; scanned = c0_ok; }
            else if (m == 1) { if (!(have & 2u)) { c1_ok = rescue_scan(g, p, v.rp, v.W, v.rl, rh, c1_f, c1_r); have |= 2u; } f = c1_f; rv = c1_r; scanned = c1_ok; }
            else scanned = rescue_scan(g, p, v.rp, v.W, v.rl, rh, f, rv);
            if (f == SLOT_BREAK) break;                                        // the flank would start before the contig: the mate loop ends
            if (!scanned) continue;                                            // unknown contig, or a pattern longer than the read
            for (int k = 0; k < 2; ++k) {
                const int32_t pos = k == 0 ? f : rv;
                if (pos == SLOT_NONE) continue;
                Hit ph; ph.ref_id = rh.ref_id; ph.left = pos; ph.right = ph.left + cl;
                ph.meta = (k == 0 ? 2u : 3u) | ((uint32_t)cl << 24);
                for (uint32_t i = l0; i < l1; ++i) fusion_pair<DEFER>(g, p, v.rp, v.W, v.rl, v.hits[i], ph, sink);
            }
        }
    }
}
template <bool DEFER = false, class Sink>
THJ_HD void fusion_read(const Genome& g, const Params& p, const ReadView& v, Sink& sink) {
    if (fusion_read_pairs<DEFER>(g, p, v, sink)) fusion_read_mates<DEFER>(g, p, v, sink);
}

// ---- packed event keys ------------------------------------------------------
// junction/deletion: [gpos(left)+1 : 34][right-left : 29][antisense : 1]; sorts like
// Junction::operator< (junctions.h:39-57) because contigs are laid out in ref_id order.
THJ_HD u64 junc_key(const Genome& g, uint32_t ref_id, uint32_t left, uint32_t right, bool anti) {
    u64 gpos = (u64)g.contig_blk[ref_id - 1] * 64ull + (u64)(int64_t)(int32_t)left + 1ull;
    u64 len = (u64)(right - left) & ((1ull << 29) - 1);
    return (gpos << 30) | (len << 1) | (anti ? 1ull : 0ull);
}
// insertion key: [gpos(left)+1 : 34][len : 4]; value: [prio : 46][seq : 18] (atomicMin => first wins)
THJ_HD u64 ins_key(const Genome& g, uint32_t ref_id, uint32_t left, int len) {
    u64 gpos = (u64)g.contig_blk[ref_id - 1] * 64ull + (u64)(int64_t)(int32_t)left + 1ull;
    return (gpos << 4) | (u64)(len & 15);
}

}  // namespace thj
