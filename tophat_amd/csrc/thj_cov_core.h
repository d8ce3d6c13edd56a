// thj_cov_core.h -- per-thread logic of the coverage search (see thj_covsearch_impl.h for the kernels and the C ABI,
// tests/hostsim for the CPU build that checks it against the oracle without a GPU).
#ifndef THJ_COV_CORE_H
#define THJ_COV_CORE_H
#include "thj_core.h"

namespace thj {
namespace cov {

static constexpr int EXTEND = 45, REPEAT_TOL = 5;       // capture_island_ends :4346-4347
static constexpr int MAX_EXT_BP = 14;                   // MerExtension::MAX_EXTENSION_BP :148
static constexpr uint32_t N_KEYS = 1u << 20;            // 4^10 seeds

struct Layout { const uint32_t* contig_blk; const int32_t* contig_len; int32_t n_contigs; int64_t n_words; };

// contig (0-based) owning word w: last k with contig_blk[k] <= w
THJ_HD int contig_of(const Layout& L, int64_t w) {
    int lo = 0, hi = L.n_contigs;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)L.contig_blk[mid] <= w) lo = mid; else hi = mid; }
    return lo;
}
// word w of a bitmap with its neighbours inside the same contig (0 outside)
struct W3 { u64 a, b, c; };
THJ_HD W3 load3(const u64* bm, const Layout& L, int k, int64_t w) {
    W3 r;
    r.b = bm[w];
    r.a = w > (int64_t)L.contig_blk[k] ? bm[w - 1] : 0ull;
    r.c = w + 1 < (int64_t)L.contig_blk[k + 1] ? bm[w + 1] : 0ull;
    return r;
}
// bits [s, s + 64) of the 128-bit value hi:lo, s in [0, 63]
THJ_HD u64 shr2(u64 lo, u64 hi, int s) { return s ? (lo >> s) | (hi << (64 - s)) : lo; }
// (x << s) with the top s bits of `below` shifted in, s in [0, 63]
THJ_HD u64 shl2(u64 below, u64 x, int s) { return s ? (x << s) | (below >> (64 - s)) : x; }
THJ_HD u64 below_mask(int64_t n_bits) { return n_bits >= 64 ? ~0ull : (n_bits <= 0 ? 0ull : ((1ull << n_bits) - 1ull)); }

// ---- coverage from hits (build_coverage_map): bits [left, right) of every hit; per contig max(right) + 1
// or_word(word index, mask), max_size(contig, size): atomics on the device.  Both targets only ever grow, so the kernel
// reads them first and skips the atomic when it would change nothing -- deep coverage means thousands of hits per word,
// and one contig means every hit of the batch raising the same size (measured: 43 ms of same-address atomics per
// 3.8 M hits without the check).
template <class OrFn, class MaxFn>
THJ_HD void add_hit(const Layout& L, const Hit& h, OrFn or_word, MaxFn max_size) {
    if (h.ref_id == 0 || (int32_t)h.ref_id > L.n_contigs) return;
    const int k = (int)h.ref_id - 1;
    const int64_t len = L.contig_len[k];
    int64_t l = h.left, r = h.right;
    if (r < 0) return;
    max_size(k, (int32_t)(r > len ? len + 1 : r + 1));       // hits are inside their contig; clamp anyway
    if (l < 0) l = 0;
    if (r > len) r = len;
    const int64_t base = (int64_t)L.contig_blk[k];
    for (int64_t w = l >> 6; w <= (r - 1) >> 6 && l < r; ++w) {
        const int64_t lo = w << 6;
        const int64_t a = l > lo ? l - lo : 0, b = r < lo + 64 ? r - lo : 64;
        const u64 m = below_mask(b) & ~below_mask(a);
        if (m) or_word(base + w, m);
    }
}

// ---- long_enough (:4368-4394).  A maximal covered run [a, b] qualifies when b - a + 2 >= min_len (b + 1 >= min_len for
// a run starting at 0) and marks [max(a, 1), b + 1].  m = min_len - 1 >= 1.
THJ_HD void long_enough_word(const Layout& L, const u64* covbits, u64* le, int m, int64_t w) {
    const int k = contig_of(L, w);
    const W3 x = load3(covbits, L, k, w);
    u64 ea = ~0ull, eb = ~0ull;                      // erosion: a run of m covered bases starts here
    for (int s = 0; s < m; ++s) { ea &= shr2(x.a, x.b, s); eb &= shr2(x.b, x.c, s); }
    u64 d = 0, da_top = 0;                           // dilation back over the run
    for (int s = 0; s < m; ++s) { d |= shl2(ea, eb, s); da_top |= (ea >> (63 - s)) & 1ull; }
    u64 r = d | (d << 1) | da_top;                   // ... plus the base after it
    if (w == (int64_t)L.contig_blk[k]) {
        // position 0: never marked; a run starting at 0 needs one base more than the others
        const u64 nb = ~x.b;
        const int run0 = nb ? __builtin_ctzll(nb) : 64;
        if (run0 == m) r &= ~below_mask(m + 1);
        r &= ~1ull;
    }
    le[w] = r;
}

// ---- look-left / look-right flags (:4426-4455), clipped to the coverage vector's size
THJ_HD void look_word(const Layout& L, const u64* le, const int32_t* cov_size, u64* ll, u64* lr, int64_t w) {
    const int k = contig_of(L, w);
    const int64_t w0 = (int64_t)L.contig_blk[k], w1 = (int64_t)L.contig_blk[k + 1];
    // long_enough of words w-2 .. w+2 (0 outside the contig): starts / ends of w-1, w, w+1 need one word further back
    u64 x[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { const int64_t q = w - 2 + i; x[i] = (q >= w0 && q < w1) ? le[q] : 0ull; }
    u64 S[3], T[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const u64 prev = (x[i + 1] << 1) | (x[i] >> 63);          // long_enough[c - 1]
        S[i] = x[i + 1] & ~prev;                                   // island starts
        T[i] = ~x[i + 1] & prev;                                   // first base after an island
        const int64_t pos0 = (w - 1 + i - w0) * 64;                // contig position of bit 0 of this word
        // an island start at c < EXTEND gets no flags (r >= 0 fails at once), an end at c < REPEAT_TOL neither; c >= 1
        S[i] &= ~below_mask(EXTEND - pos0);
        T[i] &= ~below_mask(REPEAT_TOL - pos0) & ~below_mask(1 - pos0);
        // the scan stops at c == size - 1
        const int64_t n = cov_size[k];
        S[i] &= below_mask(n - pos0); T[i] &= below_mask(n - pos0);
        if (w - 1 + i < w0 || w - 1 + i >= w1) { S[i] = 0; T[i] = 0; }
    }
    u64 fl = 0, fr = 0;
    // LOOK_LEFT  on [c - 45, c + 5):  flag[p] = OR S[p + d], d in [-4, 45]
    for (int d = 0; d <= EXTEND; ++d) fl |= shr2(S[1], S[2], d);
    for (int d = 1; d < REPEAT_TOL; ++d) fl |= shl2(S[0], S[1], d);
    // LOOK_RIGHT on [c - 5, c + 45):  flag[p] = OR T[p + d], d in [-44, 5]
    for (int d = 0; d <= REPEAT_TOL; ++d) fr |= shr2(T[1], T[2], d);
    for (int d = 1; d < EXTEND; ++d) fr |= shl2(T[0], T[1], d);
    const int64_t n = cov_size[k], pos0 = (w - w0) * 64;
    ll[w] = fl & below_mask(n - pos0);
    lr[w] = fr & below_mask(n - pos0);
}

// ---- windows that the reference drops (:4457-4512, :2154): a flag run starting at position 0, and a run whose end
// reaches len - 1.  One thread per (contig, bitmap); walks are as long as the run (a few dozen bits).
THJ_HD void clear_run(u64* bm, int64_t wbase, int64_t pos, int64_t n) {         // clears the run containing `pos`
    for (int64_t p = pos; p >= 0 && ((bm[wbase + (p >> 6)] >> (p & 63)) & 1ull); --p) bm[wbase + (p >> 6)] &= ~(1ull << (p & 63));
    for (int64_t p = pos + 1; p < n && ((bm[wbase + (p >> 6)] >> (p & 63)) & 1ull); ++p) bm[wbase + (p >> 6)] &= ~(1ull << (p & 63));
}
THJ_HD void drop_windows(const Layout& L, const int32_t* cov_size, u64* ll, u64* lr, int i) {      // i < 2 * n_contigs
    const int k = i >> 1;
    u64* bm = (i & 1) ? lr : ll;
    const int64_t wbase = (int64_t)L.contig_blk[k], n = cov_size[k], len = L.contig_len[k];
    if (n <= 0) return;
    if (bm[wbase] & 1ull) clear_run(bm, wbase, 0, n);
    for (int64_t p = n - 1; p >= len - 2 && p >= 0; --p)                // window end = last flag + 1 >= len - 1
        if ((bm[wbase + (p >> 6)] >> (p & 63)) & 1ull) clear_run(bm, wbase, p, n);
}

// ---- sites (:2290-2318): a dinucleotide at (p, p + 1) counts when both bases are flagged; N reads as A
THJ_HD void site_word(const Genome& g, const Layout& L, const u64* ll, const u64* lr, u64* fd, u64* ra, u64* fa, u64* rd, int64_t w) {
    const int k = contig_of(L, w);
    const bool last = w + 1 >= (int64_t)L.contig_blk[k + 1];
    const u64* p0 = g.blocks + w * 4;
    const u64 nm = p0[2], lo = p0[0] & ~nm, hi = p0[1] & ~nm;
    u64 nlo = 0, nhi = 0, nll = 0, nlr = 0;
    if (!last) { const u64* p1 = p0 + 4; const u64 nm1 = p1[2]; nlo = p1[0] & ~nm1; nhi = p1[1] & ~nm1; nll = ll[w + 1]; nlr = lr[w + 1]; }
    const u64 lo1 = (lo >> 1) | (nlo << 63), hi1 = (hi >> 1) | (nhi << 63);          // the next base
    const u64 A = ~lo & ~hi, C = lo & ~hi, G = ~lo & hi;
    const u64 C1 = lo1 & ~hi1, G1 = ~lo1 & hi1, T1 = lo1 & hi1;
    const u64 l = ll[w], r = lr[w];
    const u64 l2 = l & ((l >> 1) | (nll << 63)), r2 = r & ((r >> 1) | (nlr << 63));
    fd[w] = r2 & G & T1;          // GT   fwd donor        (look right)
    ra[w] = r2 & C & T1;          // CT   rev acceptor     (look right)
    fa[w] = l2 & A & G1;          // AG   fwd acceptor     (look left)
    rd[w] = l2 & A & C1;          // AC   rev donor        (look left)
}

// 2-bit strings with the first base most significant (dna5str_to_idx :218-229)
THJ_HD u64 spread32(u64 x) {                          // bit i -> bit 2i
    x &= 0xFFFFFFFFull;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}
THJ_HD u64 planes_to_mer32(u64 lo, u64 hi) {          // base i (bit i of the planes) -> bits 2 * (31 - i) + {0, 1}
    return spread32(brev64(lo) >> 32) | (spread32(brev64(hi) >> 32) << 1);
}

// ---- the extension table (:240-360): one entry per 10-mer seed position of a read's first 32 bases
// value = left_str | left_len << 28 | right_str << 32 | right_len << 60
// What a context keeps -- and what ranks exchange -- is one record per read: its first 32 bases as a 2-bit string (first base most
// significant, N as A) and min(length, 32); the up to 23 entries of a read are made from that when the table is built (round 4: twelve
// bytes a read instead of 276, and the table itself is laid out by a counting sort over the 4^10 seeds -- no sorted copy of the keys, no
// sort buffers).
THJ_HD void read_record(const u64* planes, const uint16_t* lens, int W, uint32_t* rec_len, u64* rec_seq, int64_t base, int64_t r) {
    int len = lens[r]; if (len > 32) len = 32;
    const u64* rp = planes + (size_t)r * 3 * W;
    const u64 nm = rp[2 * W], lo = rp[0] & ~nm, hi = rp[W] & ~nm;          // charToDna5 & 3: N is 0
    u64 seq = planes_to_mer32(lo, hi);                                     // base i at bits 2 * (31 - i)
    if (len > 0 && len < 32) seq &= ~0ull << (2 * (32 - len));             // (a producer may hand more bases than the length kept here)
    rec_len[base + r] = (uint32_t)len; rec_seq[base + r] = len > 0 ? seq : 0ull;
}
// the entries of one read record: emit(seed, value) per seed position
template <class Emit>
THJ_HD void record_entries(uint32_t rec_len, u64 seq, Emit emit) {
    const int len = (int)rec_len;
    if (len < 10) return;                                                  // the reference reads past the string here: undefined
    for (int i = 0; i + 10 <= len; ++i) {
        const uint32_t key = (uint32_t)((seq >> (2 * (22 - i))) & 0xFFFFFu);
        int rl = len - 10 - i; if (rl > MAX_EXT_BP) rl = MAX_EXT_BP;
        const u64 right = rl ? (seq >> (2 * (32 - (i + 10 + rl)))) & ((1ull << (2 * rl)) - 1ull) : 0ull;
        const int ln = i < MAX_EXT_BP ? i : MAX_EXT_BP;
        const u64 left = ln ? (seq >> (2 * (32 - i))) & ((1ull << (2 * ln)) - 1ull) : 0ull;
        emit(key, left | ((u64)ln << 28) | (right << 32) | ((u64)rl << 60));
    }
}
THJ_HD void key_offset(const uint32_t* sorted_keys, int64_t n, uint32_t* off, uint32_t k) {      // off[k] = first entry with key >= k; k <= N_KEYS
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sorted_keys[mid] < k) lo = mid + 1; else hi = mid; }
    off[k] = (uint32_t)lo;
}

// The table is looked at once per (donor, acceptor) candidate and orientation, and a seed has about a hundred entries at
// the scale of a whole run (23 per unmapped read over 4^10 seeds); almost every candidate fails.  A one-hash Bloom filter
// over (seed, the 7 extension bases next to it, side) answers "no" with one probe: only candidates it lets through scan
// the seed's entries (measured on the config-5 shard: the pairing kernel went from 247 ms to the numbers in DESIGN.md).
struct ExtTable { const uint32_t* off; const u64* val; const u64* filter; u64 filter_mask; };      // filter_mask = bits - 1; filter may be null
THJ_HD u64 filter_bit(uint32_t key, uint32_t ext7, int side) {
    u64 x = ((u64)key << 15) | ((u64)ext7 << 1) | (u64)side;
    x *= 0x9E3779B97F4A7C15ull; x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    return x;
}
// the filter bits of one table entry: set(bit index) for its left side (if it has >= 7 bases there) and its right side
template <class SetFn>
THJ_HD void entry_filter_bits(uint32_t key, u64 v, u64 filter_mask, SetFn set) {
    const int ln = (int)((v >> 28) & 15), rl = (int)(v >> 60);
    if (ln >= 7) set(filter_bit(key, (uint32_t)(v & 0x3FFFull), 0) & filter_mask);                          // the 7 bases before the seed
    if (rl >= 7) set(filter_bit(key, (uint32_t)(((v >> 32) & 0x0FFFFFFFull) >> (2 * (rl - 7))), 1) & filter_mask);   // the 7 after it
}
// extendable_junction (:1520-1566), min_ext_len 7, extension_mismatches 0
THJ_HD bool extendable(const ExtTable& t, u64 up, u64 down) {
    const uint32_t key = ((uint32_t)(up & 0x3FFull) << 10) | (uint32_t)(down >> 54);
    up >>= 10; down <<= 10;
    if (t.filter) {
        const u64 bl = filter_bit(key, (uint32_t)(up & 0x3FFFull), 0) & t.filter_mask, br = filter_bit(key, (uint32_t)(down >> 50), 1) & t.filter_mask;
        if (!((t.filter[bl >> 6] >> (bl & 63)) & 1ull) && !((t.filter[br >> 6] >> (br & 63)) & 1ull)) return false;
    }
    for (uint32_t i = t.off[key]; i < t.off[key + 1]; ++i) {
        const u64 v = t.val[i];
        const int ln = (int)((v >> 28) & 15), rl = (int)(v >> 60);
        if (ln >= 7 && (uint32_t)(v & 0x0FFFFFFFull) == (uint32_t)(up & ((1ull << (2 * ln)) - 1ull))) return true;
        if (rl >= 7 && (uint32_t)((v >> 32) & 0x0FFFFFFFull) == (uint32_t)(down >> (2 * (32 - rl)))) return true;
    }
    return false;
}
// 32 bases of the contig starting at pos as a 2-bit string, first base most significant (dna5str_to_idx :218-229)
THJ_HD u64 mer32(const Genome& g, uint32_t ref_id, int64_t pos) {
    const Planes p = g_fetch(g, ref_id, pos);
    return planes_to_mer32(p.lo & ~p.nm, p.hi & ~p.nm);
}
THJ_HD u64 rc32(u64 s) {                              // rc_dna_str :650-661: complement, reverse the order of the 2-bit groups
    const u64 r = brev64(~s);
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}

// ---- RecordExtendableJuncs::record (:1568-1626) for one left site `lp` of contig k; -> junctions found.
// The words of the right-site bitmap within reach are shared out over `n_lanes` callers (the 64 lanes of a wave on the
// device, one caller on the CPU): in round i lane `lane` takes word first + i * n_lanes + lane.  Every junction goes to
// ev.cov_junction(ref, left, right, antisense, skip) with skip = R - curr_R (:1615), the rank of the acceptor among the
// right sites from lp + min_intron on -- what the max_cov_juncs cut orders by; `scan(v, total)` = exclusive prefix sum of
// v over the lanes and its total (ranks of the earlier lanes' sites).
struct OneLane { THJ_HD int operator()(int v, int& total) const { total = v; return 0; } };
template <class Sink, class Scan = OneLane>
THJ_HD unsigned int pair_site(const Genome& g, const Layout& L, const ExtTable& et, const u64* right_sites, int antisense,
                              int min_intron, int max_intron, int k, int64_t lp, Sink& ev, int lane = 0, int n_lanes = 1, Scan scan = Scan()) {
    unsigned int found = 0;
    const int64_t wbase = (int64_t)L.contig_blk[k], wend = (int64_t)L.contig_blk[k + 1], len = L.contig_len[k];
    // attach_upstream_mers (:741-784): (0, 0) when too close to a contig end
    u64 lf = 0, lrv = 0;
    if (lp > 32 && lp < len) { lf = mer32(g, (uint32_t)k + 1, lp - 32); lrv = rc32(lf); }
    const int64_t q0 = lp + min_intron, q1 = lp + max_intron;            // right sites in [q0, q1)
    int64_t w_end = wbase + ((q1 + 63) >> 6);                             // words with a position below q1
    if (w_end > wend) w_end = wend;
    uint32_t base = 0;                                                    // right sites in [q0, ..) seen in earlier rounds
    for (int64_t w0 = wbase + (q0 >> 6); w0 < w_end; w0 += n_lanes) {     // the same trip count for every lane
        const int64_t rw = w0 + lane;
        const int64_t p0 = (rw - wbase) * 64;
        u64 rb = rw < w_end ? (right_sites[rw] & ~below_mask(q0 - p0) & below_mask(q1 - p0)) : 0ull;
        int total = 0;
        uint32_t rank = base + (uint32_t)scan(__builtin_popcountll(rb), total);
        base += (uint32_t)total;
        while (rb) {
            const int c = __builtin_ctzll(rb);
            rb &= rb - 1;
            const int64_t rp = p0 + c;
            u64 rf = 0, rrv = 0;                                          // attach_downstream_mers (:786-832)
            if (rp + 2 + 32 < len) { rf = mer32(g, (uint32_t)k + 1, rp + 2); rrv = rc32(rf); }
            if (extendable(et, lf, rf) || extendable(et, rrv, lrv)) {
                ev.cov_junction((uint32_t)k + 1, (uint32_t)(lp - 1), (uint32_t)(rp + 2), antisense != 0, rank);
                ++found;
            }
            ++rank;
        }
    }
    return found;
}
// the left sites of bitmap word w (the CPU build walks words; the device first compacts the sites into a list)
template <class Sink>
THJ_HD unsigned int pair_word(const Genome& g, const Layout& L, const ExtTable& et, const u64* left_sites, const u64* right_sites, int antisense,
                              int min_intron, int max_intron, int64_t w, Sink& ev) {
    u64 bits = left_sites[w];
    if (!bits) return 0;
    unsigned int found = 0;
    const int k = contig_of(L, w);
    while (bits) {
        const int b = __builtin_ctzll(bits);
        bits &= bits - 1;
        found += pair_site(g, L, et, right_sites, antisense, min_intron, max_intron, k, (w - (int64_t)L.contig_blk[k]) * 64 + b, ev);
    }
    return found;
}


// ================================================================================================ butterfly search
// segment_juncs.cpp:4178-4249 (pair_covered_sites), :1698-2049 (ButterflyKey, RecordButterflyJuncs::record), :466-501 (the extension
// table pruned to butterfly_overhang = 6 bases a side and compacted).  Opt-in (--butterfly-search).
// Every island of the coverage map (a maximal covered run [s, e)) is one window [max(s - 45, 0), e + 45) searched for all four
// dinucleotides; a window reaching the contig's last base but one is dropped whole (:2154).  A position p is therefore a candidate
// site iff some kept island has a covered base within [p - 44, p + 45]: the dilation below.  The reference's windows of
// 2 * max_intron bases over the sorted sites only bound its working set -- every (donor, acceptor) pair within (min, max) intron
// shares a window -- so the pairing here is one join over the whole genome: all (site, extension) keys of the left sites and of
// the right sites sorted, equal keys within reach paired.
static constexpr int BF_OVERHANG = 6;
static constexpr u64 BF_POS_MASK = (1ull << 34) - 1ull;

// islands whose window the reference drops: those whose last covered base lies at len - 47 or beyond.  V = the coverage bitmap's
// copy; one call per contig.
THJ_HD void bf_drop_tail(const Layout& L, u64* V, int k) {
    const int64_t wbase = (int64_t)L.contig_blk[k], nwords = (int64_t)L.contig_blk[k + 1] - wbase, len = L.contig_len[k];
    int64_t T = len - (EXTEND + 2);
    if (T < 0) T = 0;
    const bool straddles = T > 0 && ((V[wbase + (T >> 6)] >> (T & 63)) & 1ull);
    for (int64_t w = T >> 6; w < nwords; ++w) V[wbase + w] &= below_mask(T - (w << 6));
    if (!straddles) return;
    int64_t p = T - 1;                                  // the island that reaches T: cleared back to its first base
    while (p >= 0) {
        const int64_t w = p >> 6; const int b = (int)(p & 63);
        const u64 m = below_mask(b + 1), zeros = ~V[wbase + w] & m;
        if (!zeros) { V[wbase + w] &= ~m; p = (w << 6) - 1; continue; }
        const int hz = 63 - __builtin_clzll(zeros);     // the highest uncovered base at or below p
        V[wbase + w] &= ~(m & ~below_mask(hz + 1));
        break;
    }
}
// candidate positions: E[p] = OR of V[p - 44 .. p + 45]
THJ_HD void bf_eligible_word(const Layout& L, const u64* V, u64* E, int64_t w) {
    const int k = contig_of(L, w);
    const W3 x = load3(V, L, k, w);
    u64 e = 0;
    for (int d = 0; d <= EXTEND; ++d) e |= shr2(x.b, x.c, d);
    for (int d = 1; d < EXTEND; ++d) e |= shl2(x.a, x.b, d);
    E[w] = e;
}
// sites: a dinucleotide whose first base is a candidate position (both scans of a window cover the same bases); N reads as A
THJ_HD void bf_site_word(const Genome& g, const Layout& L, const u64* E, u64* fd, u64* ra, u64* fa, u64* rd, int64_t w) {
    const int k = contig_of(L, w);
    const bool last = w + 1 >= (int64_t)L.contig_blk[k + 1];
    const u64* p0 = g.blocks + w * 4;
    const u64 nm = p0[2], lo = p0[0] & ~nm, hi = p0[1] & ~nm;
    u64 nlo = 0, nhi = 0;
    if (!last) { const u64* p1 = p0 + 4; const u64 nm1 = p1[2]; nlo = p1[0] & ~nm1; nhi = p1[1] & ~nm1; }
    const u64 lo1 = (lo >> 1) | (nlo << 63), hi1 = (hi >> 1) | (nhi << 63);
    const u64 A = ~lo & ~hi, C = lo & ~hi, G = ~lo & hi;
    const u64 C1 = lo1 & ~hi1, G1 = ~lo1 & hi1, T1 = lo1 & hi1;
    const u64 e = E[w];
    fd[w] = e & G & T1;           // GT   fwd donor
    ra[w] = e & C & T1;           // CT   rev acceptor
    fa[w] = e & A & G1;           // AG   fwd acceptor
    rd[w] = e & A & C1;           // AC   rev donor
}
// The keys of one listed site (entry = contig position | contig << 32 | antisense << 63, as k_list_sites writes it): one per entry of
// the two seeds' buckets that has six bases on the side wanted.  emit(antisense << 58 | key << 34 | global position); the 24-bit key
// = the six seed bases next to the splice site and the six bases the read shows beyond it (:1803-1870 left sites, :1874-1980 right
// sites; the table is read unpruned: six bases of a longer extension are what pruning leaves).
template <class Emit>
THJ_HD void bf_site_keys(const Genome& g, const Layout& L, const ExtTable& t, u64 entry, bool right_side, Emit emit) {
    const int k = (int)((entry >> 32) & 0x7FFFFFFFull);
    const int64_t pos = (int64_t)(entry & 0xFFFFFFFFull), len = L.contig_len[k];
    const u64 head = ((entry >> 63) << 58), gpos = (u64)L.contig_blk[k] * 64ull + (u64)pos;
    u64 f = 0, r = 0;
    if (!right_side) {
        if (pos > 32 && pos < len) { f = mer32(g, (uint32_t)k + 1, pos - 32); r = rc32(f); }          // attach_upstream_mers
        const uint32_t fwd_up = (uint32_t)(f & 0xFFFFFull), rev_up = (uint32_t)(r >> 44);
        const u64 top = (u64)(fwd_up & 0xFFFu) << 12;
        for (uint32_t i = t.off[fwd_up]; i < t.off[fwd_up + 1]; ++i) {
            const u64 v = t.val[i];
            const int rl = (int)(v >> 60);
            if (rl < BF_OVERHANG) continue;
            emit(head | ((((v >> 32) & 0x0FFFFFFFull) >> (2 * (rl - BF_OVERHANG))) | top) << 34 | gpos);
        }
        for (uint32_t i = t.off[rev_up]; i < t.off[rev_up + 1]; ++i) {
            const u64 v = t.val[i];
            if ((int)((v >> 28) & 15) < BF_OVERHANG) continue;
            emit(head | ((rc32(v & 0xFFFull) >> 52) | top) << 34 | gpos);                          // the forward seed's bases here too, as the reference has it
        }
    } else {
        if (pos + 2 + 32 < len) { f = mer32(g, (uint32_t)k + 1, pos + 2); r = rc32(f); }              // attach_downstream_mers
        const uint32_t fwd_down = (uint32_t)(f >> 44), rev_down = (uint32_t)(r & 0xFFFFFull);
        const u64 bottom = (u64)(fwd_down >> 8);
        for (uint32_t i = t.off[fwd_down]; i < t.off[fwd_down + 1]; ++i) {
            const u64 v = t.val[i];
            if ((int)((v >> 28) & 15) < BF_OVERHANG) continue;
            emit(head | (((v & 0xFFFull) << 12) | bottom) << 34 | gpos);
        }
        for (uint32_t i = t.off[rev_down]; i < t.off[rev_down + 1]; ++i) {
            const u64 v = t.val[i];
            const int rl = (int)(v >> 60);
            if (rl < BF_OVERHANG) continue;
            const u64 six = ((v >> 32) & 0x0FFFFFFFull) >> (2 * (rl - BF_OVERHANG));
            emit(head | (((rc32(six) >> 52) << 12) | bottom) << 34 | gpos);
        }
    }
}
// the right keys a left key pairs with: the same antisense and key, the same contig, acceptor - donor within (min_intron, max_intron)
// where donor = left position - 1 and acceptor = right position + 2 (:2013-2041).  rkeys sorted, distinct.
struct BfRange { int64_t lo, hi; int k; int64_t cstart; };
THJ_HD BfRange bf_match_range(const Layout& L, const u64* rkeys, int64_t n_r, u64 lkey, int min_intron, int max_intron) {
    BfRange o{0, 0, 0, 0};
    const u64 gk = lkey >> 34;
    const int64_t lg = (int64_t)(lkey & BF_POS_MASK);
    o.k = contig_of(L, lg >> 6);
    o.cstart = (int64_t)L.contig_blk[o.k] * 64;
    int64_t a = lg + min_intron - 2, b = lg + (int64_t)max_intron - 4;
    const int64_t cend = o.cstart + L.contig_len[o.k];
    if (a < o.cstart) a = o.cstart;
    if (b > cend - 1) b = cend - 1;
    if (a > b) return o;
    const u64 ka = (gk << 34) | (u64)a, kb = (gk << 34) | (u64)b;
    int64_t lo = 0, hi = n_r;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rkeys[mid] < ka) lo = mid + 1; else hi = mid; }
    o.lo = lo;
    hi = n_r;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (rkeys[mid] <= kb) lo = mid + 1; else hi = mid; }
    o.hi = lo;
    return o;
}
template <class Sink>
THJ_HD void bf_emit_pairs(const BfRange& m, const u64* rkeys, u64 lkey, Sink& ev) {
    const int64_t lpos = (int64_t)(lkey & BF_POS_MASK) - m.cstart;
    for (int64_t j = m.lo; j < m.hi; ++j) {
        const int64_t rpos = (int64_t)(rkeys[j] & BF_POS_MASK) - m.cstart;
        const int64_t donor = lpos - 1, acceptor = rpos + 2;
        ev.cov_junction((uint32_t)m.k + 1, (uint32_t)donor, (uint32_t)acceptor, ((lkey >> 58) & 1ull) != 0, (uint32_t)(acceptor - donor));      // skip count = the intron's length
    }
}


// ================================================================================================ microexon search
// segment_juncs.cpp:3880-3941 (window registration in look_for_hit_group), :3675-3735 (add_to_microexon_windows -- host code,
// csrc/host/thj_mx_host.h), :3737-3815 (align_microexon_segs).  Per merged window: an extension table made of the window's own
// strings, every GT / CT in it a left site, every AG / AC a right site, pairs within [min_coverage_intron, 2000) kept when a string
// extends across them -- the coverage search's pairing with a table per window.
static constexpr int MX_STRETCH = 2000;                    // max_microexon_stretch (:60)
// A merged window can be wide: its left end is the newest candidate's, its right end the furthest of the windows that started inside it,
// and such merges chain.  Its site bitmaps therefore live in global memory: words [win_off[w], win_off[w + 1]) of four arrays.
THJ_HD int64_t mx_window_words(int32_t left, int32_t right) { return (((int64_t)right - 1) >> 6) - ((int64_t)left >> 6) + 1; }

// One candidate window of a read (what look_for_hit_group hands to add_to_microexon_windows): the read's first segment_length bases as
// a 2-bit string, first base most significant (N reads as A, :24 of the oracle: charToDna5 & 3), reverse-complemented for antisense hits.
struct MxCand { uint32_t ordinal; uint16_t rank; uint8_t side, len; uint32_t ref_id; int32_t left, right; uint32_t pad; u64 str; };
static_assert(sizeof(MxCand) == 32, "candidate record");

// the read's candidates: its first segment has no hit, every other segment has some (:3893-3910); one per hit of the second segment
template <class Emit>
THJ_HD void mx_read_candidates(const Genome& g, const Hit* hits, const uint32_t* so, int nseg, const u64* rp, int W, int rl, int seg_len, int min_anchor,
                               Emit emit) {
    if (nseg < 2 || so[1] != so[0]) return;
    for (int h = 1; h < nseg; ++h) if (so[h + 1] == so[h]) return;
    const int n = rl < seg_len ? rl : seg_len;                   // substr(0, segment_length); n <= 32 (checked by the caller)
    const u64 lo = rp[0], hi = rp[W], nm = rp[2 * W];
    u64 fwd = 0, rev = 0;
    for (int i = 0; i < n; ++i) {
        const u64 isn = (nm >> i) & 1ull;
        const u64 code = isn ? 0ull : (((hi >> i) & 1ull) << 1 | ((lo >> i) & 1ull));
        fwd = (fwd << 2) | code;                                  // base i at bits 2 * (n - 1 - i)
        rev |= (isn ? 0ull : (3ull - code)) << (2 * i);           // reverse complement: base i lands at string position n - 1 - i
    }
    const int seq_key_len = min_anchor < 6 ? min_anchor : 6;
    for (uint32_t h = so[1]; h < so[2]; ++h) {
        const Hit& bh = hits[h];
        if (bh.ref_id == 0 || (int32_t)bh.ref_id > g.n_contigs) continue;
        const int ref_len = (int)g_len(g, bh.ref_id);
        if (ref_len <= 0) continue;                               // no FASTA record for the contig (rt.get_seq == NULL)
        int lb, rb;
        if (hit_anti(bh)) { lb = bh.right - min_anchor; if (lb < 0) lb = 0; rb = lb + MX_STRETCH; if (rb > ref_len - 2) rb = ref_len - 2; }
        else { rb = bh.left + min_anchor; if (rb > ref_len - 2) rb = ref_len - 2; lb = rb - MX_STRETCH; if (lb < 0) lb = 0; }
        if (rb - lb < 2 * seq_key_len) continue;
        emit((int)(h - so[1]), bh.ref_id, lb, rb, hit_anti(bh) ? rev : fwd, n);
    }
}

// the extension-table entries of one string (store_read_extensions(extensions, 5, 5, s, false), :240-360): seed i = bases [i, i + 10);
// value as in read_entries above.  key = window << 20 | seed.
template <class Emit>
THJ_HD void mx_string_entries(u64 str, int len, u64 window, Emit emit) {
    for (int i = 0; i + 10 <= len; ++i) {
        const auto base = [&](int k) -> u64 { return (str >> (2 * (len - 1 - k))) & 3ull; };
        u64 seed = 0;
        for (int k = 0; k < 10; ++k) seed = (seed << 2) | base(i + k);
        int rl = len - 10 - i; if (rl > 14) rl = 14;
        u64 r = 0;
        for (int k = 0; k < rl; ++k) r = (r << 2) | base(i + 10 + k);
        const int ll = i < 14 ? i : 14;
        u64 l = 0;
        for (int k = i - ll; k < i; ++k) l = (l << 2) | base(k);
        emit(window << 20 | seed, (l & 0x0FFFFFFFull) | ((u64)ll << 28) | ((r & 0x0FFFFFFFull) << 32) | ((u64)rl << 60));
    }
}

// the window's table: entries of all windows sorted by key; extendable_junction against the entries of `window` only
struct MxTable { const u64* keys; const u64* vals; int64_t n; };
THJ_HD bool mx_extendable(const MxTable& t, u64 window, u64 up, u64 down) {
    const u64 key = window << 20 | ((up & 0x3FFull) << 10) | (down >> 54);
    up >>= 10; down <<= 10;
    int64_t lo = 0, hi = t.n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (t.keys[mid] < key) lo = mid + 1; else hi = mid; }
    for (int64_t i = lo; i < t.n && t.keys[i] == key; ++i) {
        const u64 v = t.vals[i];
        const int ln = (int)((v >> 28) & 15), rl = (int)(v >> 60);
        if (ln >= 7 && (uint32_t)(v & 0x0FFFFFFFull) == (uint32_t)(up & ((1ull << (2 * ln)) - 1ull))) return true;
        if (rl >= 7 && (uint32_t)((v >> 32) & 0x0FFFFFFFull) == (uint32_t)(down >> (2 * (32 - rl)))) return true;
    }
    return false;
}

// word j of the window's four site bitmaps (bit b = contig position (w0 + j) * 64 + b, w0 = left / 64): dinucleotides starting at
// positions [left, right - 2] (i <= to = seg_len - 2, :2171-2175); N reads as A; library-type skips as for a sense window (:2112-2138)
struct MxSites { u64 fd, ra, fa, rd; };
THJ_HD MxSites mx_site_word(const Genome& g, uint32_t ref_id, int32_t left, int32_t right, int library_type, int side, int j) {
    const int64_t p0 = ((int64_t)(left >> 6) + j) * 64;
    const Planes a = g_fetch(g, ref_id, p0), b = g_fetch(g, ref_id, p0 + 1);
    const u64 lo0 = a.lo & ~a.nm, hi0 = a.hi & ~a.nm, lo1 = b.lo & ~b.nm, hi1 = b.hi & ~b.nm;
    u64 valid = ~0ull;
    if (p0 < left) valid &= ~below_mask((int64_t)left - p0);
    valid &= below_mask((int64_t)right - 1 - p0);                // positions <= right - 2
    bool skip_fwd = false, skip_rev = false;
    if (library_type == 2) { if (side == 1) skip_fwd = true; else if (side == 2) skip_rev = true; }
    if (library_type == 3) { if (side == 1) skip_rev = true; else if (side == 2) skip_fwd = true; }
    MxSites s;
    s.fd = skip_fwd ? 0ull : (~lo0 & hi0 & lo1 & hi1 & valid);   // GT
    s.ra = skip_rev ? 0ull : (lo0 & ~hi0 & lo1 & hi1 & valid);   // CT
    s.fa = skip_fwd ? 0ull : (~lo0 & ~hi0 & ~lo1 & hi1 & valid); // AG
    s.rd = skip_rev ? 0ull : (~lo0 & ~hi0 & lo1 & ~hi1 & valid); // AC
    return s;
}

// RecordExtendableJuncs::record for one left site lp of the window against its right-site words rs[0 .. n_words) (word 0 = contig word w0);
// the words within reach are shared out over the lanes as in pair_site above
template <class Sink, class Scan = OneLane>
THJ_HD unsigned int mx_pair_site(const Genome& g, const MxTable& t, u64 window, uint32_t ref_id, int64_t len, const u64* rs, int64_t w0, int n_words, int antisense,
                                 int min_intron, int64_t lp, Sink& ev, int lane = 0, int n_lanes = 1, Scan scan = Scan()) {
    unsigned int found = 0;
    u64 lf = 0, lrv = 0;
    if (lp > 32 && lp < len) { lf = mer32(g, ref_id, lp - 32); lrv = rc32(lf); }
    const int64_t q0 = lp + min_intron, q1 = lp + MX_STRETCH;
    int64_t w_end = (q1 + 63) >> 6;
    if (w_end > w0 + n_words) w_end = w0 + n_words;
    uint32_t base = 0;
    for (int64_t wa = q0 >> 6; wa < w_end; wa += n_lanes) {
        const int64_t rw = wa + lane;
        const int64_t p0 = rw * 64;
        u64 rb = (rw < w_end && rw >= w0) ? (rs[rw - w0] & ~below_mask(q0 - p0) & below_mask(q1 - p0)) : 0ull;
        int total = 0;
        uint32_t rank = base + (uint32_t)scan(__builtin_popcountll(rb), total);
        base += (uint32_t)total;
        while (rb) {
            const int c = __builtin_ctzll(rb);
            rb &= rb - 1;
            const int64_t rp = p0 + c;
            u64 rf = 0, rrv = 0;
            if (rp + 2 + 32 < len) { rf = mer32(g, ref_id, rp + 2); rrv = rc32(rf); }
            if (mx_extendable(t, window, lf, rf) || mx_extendable(t, window, rrv, lrv)) {
                ev.cov_junction(ref_id, (uint32_t)(lp - 1), (uint32_t)(rp + 2), antisense != 0, rank);
                ++found;
            }
            ++rank;
        }
    }
    return found;
}

}  // namespace cov
}  // namespace thj
#endif
