// thj_deflate_core.h -- one BGZF member's DEFLATE stream and CRC-32 by one workgroup of 16 waves (the device side of the BAM
// writer: long_spanning_reads' output, bam_write1 -> bgzf_write -> deflate_block, samtools-0.1.18 bgzf.c:287-349 via bam.c:207-236).
//
// The reference calls zlib; any valid DEFLATE stream of the same bytes is the same BAM file to every reader, so the work is laid
// out for the machine rather than after zlib:
//   * the member (<= 64 KiB) sits in LDS; 16 waves each own a 4 KiB slice and find matches for 64 consecutive positions at a
//     time: one 4-byte-hash probe per position into the wave's own 2 K-entry table (positions of earlier 64-byte steps, the
//     slice before it included by a warm-up), the byte before as the second candidate (runs), lengths by 8-byte compares;
//   * the greedy parse of the 64 positions is a scalar walk over ballots (runs of literals in one step, one readlane per match);
//     tokens go to HBM, symbol counts to the wave's histogram in LDS;
//   * one dynamic-Huffman block per member: exact Huffman lengths (sorted by rank counting, two-queue merge on one lane,
//     depths in parallel; the length limit enforced as thj_fastdeflate.h does), canonical codes from ballots;
//   * every wave then knows its slice's bit length (histogram x code lengths), so the slices' bit strings are written in
//     parallel at their final bit offsets through a small LDS staging buffer per wave.
//   * CRC-32: 64-byte pieces per thread, combined pairwise by multiplying with x^(8 * length) modulo the CRC polynomial.
//
// The code is written against an execution context X (lane / wave ids, ballot, shuffle, scan, barriers, LDS atomics) so that the
// same source is the HIP kernel (thj_bamout.hip) and, under tests/hostsim/simt.h, a CPU build the CPU test-suite inflates with
// zlib.  Wave-wide operations appear in wave-uniform control flow only.
#pragma once
#include <stdint.h>
#include <string.h>

#ifndef THJ_DFN
#define THJ_DFN inline
#endif

namespace dfl {

constexpr int NW = 16;                       // waves per member
constexpr int NT = NW * 64;
#ifndef DFL_TB
#define DFL_TB 11
#endif
#ifndef DFL_WARM
#define DFL_WARM 4096
#endif
constexpr int TB = DFL_TB;                   // log2 hash-table entries per wave
constexpr uint32_t MAXN = 65536;
constexpr uint32_t WARM = DFL_WARM;              // bytes of the previous slice whose positions a wave enters into its table first
constexpr uint32_t CAP = 20;                 // bytes of a match a lane measures by itself
constexpr uint32_t NSYM = 320;               // histogram row: literal/length symbols at 0..285, distance symbols at 288..317
constexpr uint32_t DSYM = 288;

// ---- LDS layout (bytes)
constexpr uint32_t L_DATA = 0;                               // the member's bytes + 64 bytes of zeros
constexpr uint32_t L_TAB = MAXN + 64;                        // u16 [NW << TB]; after the match phase: staging, CRC, Huffman scratch
constexpr uint32_t L_HIST = L_TAB + ((uint32_t)NW << TB) * 2;   // u32 [NW][NSYM]
constexpr uint32_t L_TOT = L_HIST + NW * NSYM * 4;           // u32 [NSYM] member totals
constexpr uint32_t L_LEN = L_TOT + NSYM * 4;                 // u8  [NSYM] code lengths
constexpr uint32_t L_CODE = L_LEN + NSYM;                    // u16 [NSYM] codes, bit-reversed
constexpr uint32_t L_CL = L_CODE + NSYM * 2;                 // code-length alphabet: u32 freq[32], u8 len[32], u16 code[32]
constexpr uint32_t L_MISC = L_CL + 32 * 4 + 32 + 32 * 2;     // u32 [64]
constexpr uint32_t L_CRCT = L_MISC + 64 * 4;                 // u32 [4][256] CRC tables (slicing by four)
constexpr uint32_t L_XP = L_CRCT + 4096;                    // u32 [64] x^(512 j) mod p, then u32 [16] x^(32768 j) mod p
constexpr uint32_t L_END = L_XP + 80 * 4;
#ifndef DFL_EXPERIMENT
static_assert(L_END <= 160 * 1024, "LDS of one CU");
#endif
// inside L_TAB once the tables are dead
constexpr uint32_t A_STAGE = 0;                              // u32 [NW][128]
constexpr uint32_t A_CRC = A_STAGE + NW * 512;               // u32 [NW] (room for NT)
constexpr uint32_t A_HUF = A_CRC + NT * 4;                   // two Huffman builds side by side, HUF_BYTES each
constexpr uint32_t HUF_BYTES = 576 * 4 + 576 * 2 + 288 * 2 + 288;
constexpr uint32_t A_HDR = A_HUF + 2 * HUF_BYTES;            // u32 [128] header staging (a bit sink like the slices')
static_assert(A_HDR + 512 <= ((uint32_t)NW << TB) * 2, "aliases fit the table area");
// L_MISC words
enum { M_NTOK = 0 /*[NW]*/, M_BITS = 16 /*[NW]*/, M_HDRBITS = 32, M_HLIT = 33, M_HDIST = 34, M_FAIL = 35, M_CRC = 36 };

enum { ST_OK = 0, ST_TOO_BIG = 1, ST_HUFF = 2 };

THJ_DFN uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
THJ_DFN uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
THJ_DFN int ctz64(uint64_t v) { return __builtin_ctzll(v); }
THJ_DFN int popc64(uint64_t v) { return __builtin_popcountll(v); }
THJ_DFN int flog2(uint32_t v) { return 31 - __builtin_clz(v); }

// length (3..258) -> literal/length symbol - 257, number of extra bits, their value
THJ_DFN void len_sym(uint32_t l, uint32_t& idx, uint32_t& nb, uint32_t& ext) {
    const uint32_t x = l - 3;
    if (x < 8) { idx = x; nb = 0; ext = 0; return; }
    if (l == 258) { idx = 28; nb = 0; ext = 0; return; }
    nb = (uint32_t)flog2(x) - 2;
    idx = 4 * nb + 4 + ((x >> nb) & 3u);
    ext = x & ((1u << nb) - 1u);
}
// distance (1..32768) -> distance symbol, number of extra bits, their value
THJ_DFN void dist_sym(uint32_t d, uint32_t& idx, uint32_t& nb, uint32_t& ext) {
    const uint32_t x = d - 1;
    if (x < 4) { idx = x; nb = 0; ext = 0; return; }
    nb = (uint32_t)flog2(x) - 1;
    idx = 2 * nb + 2 + ((x >> nb) & 1u);
    ext = x & ((1u << nb) - 1u);
}
THJ_DFN uint32_t sym_extra_bits(uint32_t s) {     // by histogram index
    if (s < 265) return 0;
    if (s < 285) return (s - 261) >> 2;
    if (s < DSYM + 4) return 0;
    if (s < DSYM + 30) return (s - DSYM - 2) >> 1;
    return 0;
}

// ---- CRC-32 (the gzip polynomial, reflected)
constexpr uint32_t CRC_POLY = 0xEDB88320u;
constexpr uint32_t crc_multmodp_c(uint32_t a, uint32_t b) {         // a(x) * b(x) mod p(x); a != 0
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
THJ_DFN uint32_t crc_multmodp(uint32_t a, uint32_t b) {
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
        m >>= 1;
        b = (b & 1u) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
// x^(2^k) mod p, at compile time
constexpr uint32_t crc_x2n_c(int k) { uint32_t v = 1u << 30; for (int i = 0; i < k; ++i) v = crc_multmodp_c(v, v); return v; }
// x^(2^(k + 9)) mod p: what level k of the CRC tree multiplies with (its right-hand operand is 2^(k+9) bits long)
THJ_DFN uint32_t crc_level_factor(int k) {
    constexpr uint32_t F0 = crc_x2n_c(9), F1 = crc_x2n_c(10), F2 = crc_x2n_c(11), F3 = crc_x2n_c(12), F4 = crc_x2n_c(13), F5 = crc_x2n_c(14), F6 = crc_x2n_c(15),
                       F7 = crc_x2n_c(16), F8 = crc_x2n_c(17), F9 = crc_x2n_c(18);
    switch (k) { case 0: return F0; case 1: return F1; case 2: return F2; case 3: return F3; case 4: return F4; case 5: return F5; case 6: return F6;
                 case 7: return F7; case 8: return F8; default: return F9; }
}

// ---- a wave's bit string on its way to HBM.  stg: 128 zeroed LDS words; word 0 holds global word w0's bits from `sb` on.
struct BitSink {
    uint32_t* stg; uint32_t* out; uint32_t w0, sb; bool first;
};
template <class X>
THJ_DFN void sink_open(X& x, BitSink& s, uint32_t* stg, uint32_t* out, uint32_t bit) {
    s.stg = stg; s.out = out; s.w0 = bit >> 5; s.sb = bit & 31u; s.first = true;
    stg[x.lane] = 0; stg[x.lane + 64] = 0;
    x.wsync();
}
// every lane appends nb (<= 48) bits, in lane order
template <class X>
THJ_DFN void sink_put(X& x, BitSink& s, uint64_t bits, uint32_t nb) {
    const uint32_t incl = x.incl_scan(nb), tot = x.bcast(incl, 63);
    if (tot == 0) return;
    if (nb) {
        const uint32_t pos = s.sb + incl - nb, sh = pos & 31u, wi = pos >> 5;
        const uint64_t lo = bits << sh;
        x.lds_or(s.stg + wi, (uint32_t)lo);
        if (sh + nb > 32) x.lds_or(s.stg + wi + 1, (uint32_t)(lo >> 32));
        if (sh + nb > 64) x.lds_or(s.stg + wi + 2, (uint32_t)(bits >> (64 - sh)));
    }
    x.wsync();
    const uint32_t endbit = s.sb + tot, nfull = endbit >> 5;
    for (uint32_t j = (uint32_t)x.lane; j < nfull; j += 64) {
        const uint32_t v = s.stg[j];
        if (s.first && j == 0) x.glb_or(s.out + s.w0, v); else s.out[s.w0 + j] = v;
    }
    const uint32_t carry = s.stg[nfull];
    x.wsync();
    if (nfull) {
        for (uint32_t j = (uint32_t)x.lane; j <= nfull; j += 64) s.stg[j] = 0;
        x.wsync();
        if (x.lane == 0) s.stg[0] = carry;
        x.wsync();
        s.w0 += nfull; s.first = false;
    }
    s.sb = endbit & 31u;
}
template <class X>
THJ_DFN void sink_close(X& x, BitSink& s) {
    if (x.lane == 0 && s.sb) x.glb_or(s.out + s.w0, s.stg[0]);
}

// ---- Huffman code lengths for n symbols (n <= 288), at most maxbits long, by ONE wave.  freq / len in LDS; scr: HUF_BYTES of
// LDS.  A symbol with count 0 gets length 0, a lone symbol length 1.  Returns false (wave-uniform) if the limit could not be met.
template <class X>
THJ_DFN bool huff_lengths(X& x, const uint32_t* freq, int n, int maxbits, uint8_t* len, uint8_t* scr) {
    uint32_t* w = (uint32_t*)scr;                     // [576] node weights: leaves 0..m-1 in rising order, then internal nodes
    uint16_t* parent = (uint16_t*)(scr + 576 * 4);    // [576]
    uint16_t* idx = (uint16_t*)(scr + 576 * 4 + 576 * 2);   // [288] symbol of leaf i
    uint8_t* depth = scr + 576 * 4 + 576 * 2 + 288 * 2;     // [288]
    // rank of every used symbol among the used ones, by (count, symbol)
    int m = 0;
    for (int base = 0; base < n; base += 64) m += popc64(x.ballot(base + x.lane < n && freq[base + x.lane] != 0));
    for (int i = x.lane; i < n; i += 64) {
        len[i] = 0;
        const uint32_t f = freq[i];
        if (!f) continue;
        int r = 0;
        for (int j = 0; j < n; ++j) { const uint32_t g = freq[j]; r += (g != 0 && (g < f || (g == f && j < i))) ? 1 : 0; }
        idx[r] = (uint16_t)i; w[r] = f;
    }
    x.wsync();
    if (m == 0) return true;
    if (m == 1) { if (x.lane == 0) len[idx[0]] = 1; x.wsync(); return true; }
    if (x.lane == 0) {                                // two-queue merge: internal nodes come out in non-decreasing weight
        int leaf = 0, inode = m, next = m;
        while (next < 2 * m - 1) {
            int pick[2];
            for (int k = 0; k < 2; ++k) {
                if (leaf < m && (inode >= next || w[leaf] <= w[inode])) pick[k] = leaf++; else pick[k] = inode++;
            }
            w[next] = w[pick[0]] + w[pick[1]];
            parent[pick[0]] = (uint16_t)next; parent[pick[1]] = (uint16_t)next;
            ++next;
        }
    }
    x.wsync();
    const int root = 2 * m - 2;
    uint32_t maxd = 0;
    for (int base = 0; base < m; base += 64) {
        const int i = base + x.lane;
        uint32_t d = 0;
        if (i < m) { for (int node = i; node != root; node = parent[node]) ++d; depth[i] = (uint8_t)(d > 255 ? 255 : d); }
        const uint32_t mx = x.wave_max(d);
        if (mx > maxd) maxd = mx;
    }
    x.wsync();
    bool ok = true;
    if ((int)maxd > maxbits) {
        // clamp, lengthen the rarest symbols that still can be until the Kraft sum fits, give the slack back to the commonest
        uint32_t good = 1;
        if (x.lane == 0) {
            uint32_t kraft = 0; const uint32_t one = 1u << maxbits;
            for (int i = 0; i < m; ++i) { if (depth[i] > maxbits) depth[i] = (uint8_t)maxbits; kraft += one >> depth[i]; }
            while (kraft > one) {
                bool moved = false;
                for (int i = 0; i < m && kraft > one; ++i)
                    if (depth[i] < maxbits) { kraft -= (one >> depth[i]) - (one >> (depth[i] + 1)); ++depth[i]; moved = true; }
                if (!moved) break;
            }
            for (int i = m - 1; i >= 0; --i)
                while (depth[i] > 1 && kraft + (one >> depth[i]) <= one) { kraft += one >> depth[i]; --depth[i]; }
            good = kraft == one ? 1u : 0u;
        }
        ok = x.bcast(good, 0) != 0;
    }
    x.wsync();
    for (int i = x.lane; i < m; i += 64) len[idx[i]] = depth[i];
    x.wsync();
    return ok;
}

// the order the code-length code's own lengths are sent in (RFC 1951, 3.2.7)
THJ_DFN int cl_order(int k) {
    const uint64_t lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) | (5ull << 45) | (11ull << 50) | (4ull << 55);
    const uint64_t hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
    return k < 12 ? (int)((lo >> (5 * k)) & 31u) : k < 19 ? (int)((hi >> (5 * (k - 12))) & 31u) : 0;
}
// The same contract, without the tree: Shannon lengths ceil(log2(total / count)) -- a prefix code by Kraft's inequality --, then the
// slack of the Kraft sum given back: first to the symbols whose count is largest for their length (a threshold on count << length,
// found by bisection), the rest class by class from the longest codes up.  Everything is wave-wide sums over five symbols a lane;
// the code is complete (Kraft sum exactly 1) and on BAM members within a fraction of a percent of Huffman's.
template <class X>
THJ_DFN bool huff_lengths_fast(X& x, const uint32_t* freq, int n, int maxbits, uint8_t* len) {
    uint32_t f[5], l[5];
    uint32_t sum = 0, cnt = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { const int i = x.lane + 64 * k; f[k] = i < n ? freq[i] : 0u; l[k] = 0; sum += f[k]; cnt += f[k] ? 1u : 0u; }
    const uint32_t T = x.bcast(x.incl_scan(sum), 63), m = x.bcast(x.incl_scan(cnt), 63);
    const uint32_t B = (uint32_t)maxbits, one = 1u << B;
    if (m >= 2) {
#pragma unroll
        for (int k = 0; k < 5; ++k) if (f[k]) {
            const uint32_t q = (T + f[k] - 1) / f[k];
            uint32_t v = q <= 1 ? 1u : 32u - (uint32_t)__builtin_clz(q - 1);
            l[k] = v < 1 ? 1u : v > B ? B : v;
        }
        uint32_t K;
        for (;;) {
            uint32_t kk = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) if (f[k]) kk += one >> l[k];
            K = x.bcast(x.incl_scan(kk), 63);
            if (K <= one) break;
#pragma unroll
            for (int k = 0; k < 5; ++k) if (f[k] && l[k] < B) ++l[k];     // the clamp at maxbits overfilled the code: everything else one longer
        }
        uint32_t S = one - K;
        if (S) {
            uint32_t mx = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) if (f[k] && l[k] > 1) { const uint32_t key = f[k] << l[k]; mx = key > mx ? key : mx; }
            uint32_t lo = 0, hi = x.wave_max(mx) + 1;                     // cost(hi) = 0 fits; the smallest threshold that still fits
            while (hi - lo > 1) {
                const uint32_t mid = lo + (hi - lo) / 2;
                uint32_t c = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) if (f[k] && l[k] > 1 && (f[k] << l[k]) >= mid) c += one >> l[k];
                if (x.bcast(x.incl_scan(c), 63) <= S) hi = mid; else lo = mid;
            }
            uint32_t c = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) if (f[k] && l[k] > 1 && (f[k] << l[k]) >= hi) { c += one >> l[k]; --l[k]; }
            S -= x.bcast(x.incl_scan(c), 63);
            for (uint32_t L = B; L >= 2 && S; --L) {
                const uint32_t unit = one >> L;
                uint32_t mine = 0;
#pragma unroll
                for (int k = 0; k < 5; ++k) mine += (f[k] && l[k] == L) ? 1u : 0u;
                const uint32_t incl = x.incl_scan(mine), tot = x.bcast(incl, 63);
                const uint32_t t = tot < S / unit ? tot : S / unit;
                if (!t) continue;
                const uint32_t before = incl - mine;
                uint32_t quota = t > before ? (t - before < mine ? t - before : mine) : 0u;
#pragma unroll
                for (int k = 0; k < 5; ++k) if (quota && f[k] && l[k] == L) { --l[k]; --quota; }
                S -= t * unit;
            }
        }
        if (S) return false;
    } else if (m == 1) {
#pragma unroll
        for (int k = 0; k < 5; ++k) if (f[k]) l[k] = 1;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) { const int i = x.lane + 64 * k; if (i < n) len[i] = (uint8_t)l[k]; }
    x.wsync();
    return true;
}

THJ_DFN uint32_t bitrev(uint32_t v, uint32_t l) {
    uint32_t r = 0;
    for (uint32_t k = 0; k < l; ++k) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}
// canonical codes (bit-reversed: DEFLATE packs Huffman codes MSB first into an LSB-first stream) for n <= 320 symbols, by one wave
template <class X>
THJ_DFN void huff_codes(X& x, const uint8_t* len, int n, uint16_t* code) {
    uint32_t cnt[16];
    for (int L = 0; L < 16; ++L) cnt[L] = 0;
    for (int base = 0; base < n; base += 64) {
        const int i = base + x.lane;
        const uint32_t l = i < n ? len[i] : 0;
#pragma unroll
        for (int L = 1; L < 16; ++L) cnt[L] += (uint32_t)popc64(x.ballot(l == (uint32_t)L));
    }
    uint32_t first[16], c = 0;
    first[0] = 0; cnt[0] = 0;
    for (int L = 1; L < 16; ++L) { c = (c + cnt[L - 1]) << 1; first[L] = c; }
    const uint64_t lt = x.lane ? (~0ull >> (64 - x.lane)) : 0ull;
    for (int base = 0; base < n; base += 64) {
        const int i = base + x.lane;
        const uint32_t l = i < n ? len[i] : 0;
        uint32_t v = 0;
#pragma unroll
        for (int L = 1; L < 16; ++L) {
            const uint64_t b = x.ballot(l == (uint32_t)L);
            if (l == (uint32_t)L) v = first[L] + (uint32_t)popc64(b & lt);
            first[L] += (uint32_t)popc64(b);
        }
        if (i < n) code[i] = (uint16_t)bitrev(v, l);
    }
    x.wsync();
}

// position p into slot h of the wave's table, for the lanes that have one.  Lanes with the same slot: the highest position stays,
// whichever store the hardware lets win (the output must not depend on it).  Ends with the table settled for every lane.
template <class X>
THJ_DFN void tab_insert(X& x, uint16_t* tab, uint32_t h, uint32_t p, bool active) {
    if (active) tab[h] = (uint16_t)p;
    for (;;) {
        x.wsync();
        const bool lost = active && tab[h] < (uint16_t)p;
        if (!x.ballot(lost)) break;
        if (lost) tab[h] = (uint16_t)p;
    }
}

#ifdef DFL_EXACT_HUFFMAN
#define DFL_LENGTHS(x, f, n, b, l, scr) huff_lengths(x, f, n, b, l, scr)
#else
#define DFL_LENGTHS(x, f, n, b, l, scr) huff_lengths_fast(x, f, n, b, l)
#endif

// ---- the member.  lds: L_END bytes; in: n (1..65536) bytes; tokens: 65536 words of scratch; out: 16384 words, zeroed before the
// launch; result[0..2] = compressed bytes, CRC-32, status (ST_*).
template <class X>
THJ_DFN void deflate_member(X& x, uint8_t* lds, const uint8_t* in, uint32_t n, uint32_t* tokens, uint32_t* out, uint32_t* result, unsigned long long* tm = nullptr) {
#define DFL_MARK(k) do { if (tm && x.tid == 0) tm[k] = x.clock(); } while (0)
    DFL_MARK(0);
    uint8_t* data = lds + L_DATA;
    uint16_t* tab = (uint16_t*)(lds + L_TAB) + ((size_t)x.wave << TB);
    uint32_t* hist = (uint32_t*)(lds + L_HIST) + (size_t)x.wave * NSYM;
    uint32_t* tot = (uint32_t*)(lds + L_TOT);
    uint8_t* clen_all = lds + L_LEN;
    uint16_t* code_all = (uint16_t*)(lds + L_CODE);
    uint32_t* cl_freq = (uint32_t*)(lds + L_CL);
    uint8_t* cl_len = lds + L_CL + 128;
    uint16_t* cl_code = (uint16_t*)(lds + L_CL + 160);
    uint32_t* misc = (uint32_t*)(lds + L_MISC);
    uint32_t* crct = (uint32_t*)(lds + L_CRCT);
    uint32_t* xp = (uint32_t*)(lds + L_XP);
    uint8_t* alias = lds + L_TAB;

    // ---- phase 0: the member into LDS, tables empty, histograms zero, the CRC table
    for (uint32_t o = (uint32_t)x.tid * 16; o < MAXN + 64; o += NT * 16) {
        uint8_t v[16];
        if (o + 16 <= n) memcpy(v, in + o, 16);
        else for (int k = 0; k < 16; ++k) v[k] = o + (uint32_t)k < n ? in[o + (uint32_t)k] : 0;
        memcpy(data + o, v, 16);
    }
    for (uint32_t i = (uint32_t)x.lane; i < (1u << TB); i += 64) tab[i] = 0xFFFFu;
    for (uint32_t i = (uint32_t)x.lane; i < NSYM; i += 64) hist[i] = 0;
    if (x.tid < 256) {
        uint32_t c = (uint32_t)x.tid, t0[4];
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1;
        t0[0] = c;
        // table j: the CRC of byte tid followed by j zero bytes (bitwise here: the other tables' entries are not written yet)
        for (int j = 1; j < 4; ++j) { for (int k = 0; k < 8; ++k) c = (c & 1u) ? (c >> 1) ^ CRC_POLY : c >> 1; t0[j] = c; }
        for (int j = 0; j < 4; ++j) crct[j * 256 + x.tid] = t0[j];
    }
    if (x.tid < 40) misc[x.tid] = 0;
    if (x.wave == 2 || (x.wave == 3 && x.lane < 16)) {
        // what a piece's (a wave's) CRC is multiplied with to stand j pieces (waves) further left: products of the level factors
        uint32_t v = 1u << 31;
        const int base = x.wave == 2 ? 0 : 6, nb = x.wave == 2 ? 6 : 4;
        for (int k = 0; k < nb; ++k) if ((x.lane >> k) & 1) v = crc_multmodp(crc_level_factor(base + k), v);
        xp[(x.wave == 2 ? 0 : 64) + x.lane] = v;
    }
    x.sync();

    DFL_MARK(1);
    // ---- phase 1: matches and the greedy parse, one slice per wave
    const uint32_t SL = (((n + NW - 1) / NW) + 63u) & ~63u;
    const uint32_t sbeg = (uint32_t)x.wave * SL < n ? (uint32_t)x.wave * SL : n;
    const uint32_t send = sbeg + SL < n ? sbeg + SL : n;
    uint32_t* tk = tokens + sbeg;                      // at most one token per byte
    uint32_t ntok = 0;
    if (sbeg < send) {
        for (uint32_t p0 = sbeg > WARM ? sbeg - WARM : 0; p0 < sbeg; p0 += 64) {
            const uint32_t p = p0 + (uint32_t)x.lane;
            tab_insert(x, tab, (ld32(data + p) * 2654435761u) >> (32 - TB), p, p < sbeg);
        }
        uint32_t e = 0;                                // positions of this step a token that began earlier already covers
        for (uint32_t p0 = sbeg; p0 < send; p0 += 64) {
            if (e >= 64) { e -= 64; continue; }
            const uint32_t p = p0 + (uint32_t)x.lane;
            const uint32_t maxl = p < send ? (send - p < 258u ? send - p : 258u) : 0u;
            const bool can = maxl >= 4;
            const uint32_t v = ld32(data + p);
            const uint32_t h = (v * 2654435761u) >> (32 - TB);
            uint32_t cand = tab[h];
            x.wsync();
            tab_insert(x, tab, h, p, can);
            bool ok = can && cand != 0xFFFFu && cand < p && p - cand <= 32768u && ld32(data + cand) == v;
            if (!ok && can && p >= 1 && ld32(data + p - 1) == v) { cand = p - 1; ok = true; }
            // every lane looks CAP bytes far on its own; a longer match is measured by the whole wave, and only if the parse takes it
            // (in a run of equal bytes every lane has one: sixty-four private loops of thirty rounds each otherwise)
            uint32_t l = 1;
            bool sat = false;
            if (ok) {
                l = 4;
                const uint64_t d0 = ld64(data + cand + 4) ^ ld64(data + p + 4), d1 = ld64(data + cand + 12) ^ ld64(data + p + 12);
                if (d0) l += (uint32_t)ctz64(d0) >> 3;
                else if (d1) l += 8 + ((uint32_t)ctz64(d1) >> 3);
                else l = CAP;
                if (l >= maxl) l = maxl; else sat = l == CAP;
            }
            const uint32_t nvalid = send - p0 < 64u ? send - p0 : 64u;
            const uint64_t litmask = x.ballot(l == 1);
            uint64_t start = 0;
            uint32_t f = e;
            while (f < nvalid) {
                if ((litmask >> f) & 1ull) {
                    const uint64_t rest = ~(litmask >> f);
                    uint32_t run = rest ? (uint32_t)ctz64(rest) : 64u;
                    if (run > nvalid - f) run = nvalid - f;
                    start |= (run >= 64 ? ~0ull : ((1ull << run) - 1ull)) << f;
                    f += run;
                } else {
                    start |= 1ull << f;
                    uint32_t lf = x.bcast(l, (int)f);
                    if (x.bcast(sat ? 1u : 0u, (int)f)) {
                        const uint32_t cf = x.bcast(cand, (int)f), pf = p0 + f, mf = x.bcast(maxl, (int)f), off = CAP + 8u * (uint32_t)x.lane;
                        uint64_t d = 0;
                        if (off < mf) d = ld64(data + cf + off) ^ ld64(data + pf + off);
                        const uint64_t mism = x.ballot(d != 0);
                        if (mism) { const int fl = ctz64(mism); lf = CAP + 8u * (uint32_t)fl + x.bcast((uint32_t)(d ? ctz64(d) >> 3 : 0), fl); } else lf = mf;
                        if (lf > mf) lf = mf;
                        if ((uint32_t)x.lane == f) l = lf;
                    }
                    f += lf;
                }
            }
            e = f - nvalid;                            // (only a step that is not the slice's last can leave a remainder)
            const bool mine = (start >> x.lane) & 1ull;
            const uint32_t rank = (uint32_t)popc64(start & (x.lane ? (~0ull >> (64 - x.lane)) : 0ull));
            if (mine) {
                if (l == 1) { const uint32_t b = data[p]; tk[ntok + rank] = b; x.lds_add(hist + b, 1); }
                else {
                    const uint32_t dist = p - cand;
                    tk[ntok + rank] = 0x80000000u | ((l - 3) << 16) | (dist - 1);
                    uint32_t li, ln, le, di, dn, de;
                    len_sym(l, li, ln, le); dist_sym(dist, di, dn, de);
                    x.lds_add(hist + 257 + li, 1); x.lds_add(hist + DSYM + di, 1);
                }
            }
            ntok += (uint32_t)popc64(start);
        }
    }
    if (x.lane == 0) misc[M_NTOK + x.wave] = ntok;
    x.sync();

    DFL_MARK(2);
    // ---- phase 2: member totals; CRC pieces (the tables are dead from here on)
    for (uint32_t i = (uint32_t)x.tid; i < NSYM; i += NT) {
        uint32_t s = 0;
        for (int w = 0; w < NW; ++w) s += ((uint32_t*)(lds + L_HIST))[(size_t)w * NSYM + i];
        if (i == 256) s = 1;
        tot[i] = s; clen_all[i] = 0;
    }
    uint32_t* crcp = (uint32_t*)(alias + A_CRC);
    {   // pieces aligned to the END of the member, so that every right-hand operand of a combine is a whole piece: level k joins
        // groups of 2^k pieces, the right one 2^(k+9) bits long (crc_level_factor).  Here in one step per wave, the waves' results in phase 4.
        const int64_t pe = (int64_t)n - (int64_t)(NT - 1 - x.tid) * 64, pb = pe - 64;
        uint32_t c = 0;
        if (pe > 0) {
            c = 0xFFFFFFFFu;
            if (pb >= 0) {                            // a whole piece: four bytes a step, four independent look-ups
                const uint8_t* q = data + pb;
                for (int k = 0; k < 16; ++k) {
                    c ^= ld32(q + 4 * k);
                    c = crct[768 + (c & 0xFFu)] ^ crct[512 + ((c >> 8) & 0xFFu)] ^ crct[256 + ((c >> 16) & 0xFFu)] ^ crct[c >> 24];
                }
            } else for (int64_t k = 0; k < pe; ++k) c = crct[(c ^ data[k]) & 0xFFu] ^ (c >> 8);
            c ^= 0xFFFFFFFFu;
        }
        // crc(A || B) = crc(A) * x^(bits of B) ^ crc(B): every piece moved to the end of its wave's range, then XORed together
        c = x.wave_xor(crc_multmodp(xp[63 - x.lane], c));
        if (x.lane == 0) crcp[x.wave] = c;
    }
    if (x.tid < 128) ((uint32_t*)(alias + A_HDR))[x.tid] = 0;
    x.sync();

    DFL_MARK(3);
    // ---- phase 3: code lengths (wave 0: literal/length, wave 1: distance)
    if (x.wave == 0) { if (!DFL_LENGTHS(x, tot, 286, 15, clen_all, alias + A_HUF) && x.lane == 0) misc[M_FAIL] = ST_HUFF; }
    else if (x.wave == 1) {
        if (!DFL_LENGTHS(x, tot + DSYM, 30, 15, clen_all + DSYM, alias + A_HUF + HUF_BYTES) && x.lane == 0) misc[M_FAIL] = ST_HUFF;
        // no distance symbol at all: one unused 1-bit code (an inflater accepts that; an empty distance alphabet not every one does)
        bool any = false;
        for (int i = 0; i < 30; ++i) any = any || clen_all[DSYM + i] != 0;
        if (!any && x.lane == 0) clen_all[DSYM] = 1;
    }
    x.sync();
    DFL_MARK(4);
    DFL_MARK(5);
    // ---- phase 4: the codes (waves 1, 2), and on wave 0 the code-length code and the block header
    if (x.wave == 1) huff_codes(x, clen_all, 286, code_all);
    else if (x.wave == 2) huff_codes(x, clen_all + DSYM, 30, code_all + DSYM);
    else if (x.wave == 3) {
        uint32_t c = x.wave_xor(x.lane < NW ? crc_multmodp(xp[64 + NW - 1 - x.lane], crcp[x.lane]) : 0u);
        if (x.lane == 0) misc[M_CRC] = c;
    }
    else if (x.wave == 0) {
        int hlit = 257, hdist = 1;
        {
            const uint64_t bl = x.ballot(x.lane < 29 && clen_all[257 + x.lane] != 0);
            const uint64_t bd = x.ballot(x.lane < 30 && clen_all[DSYM + x.lane] != 0);
            if (bl) hlit = 257 + 64 - __builtin_clzll(bl);
            if (bd) hdist = 64 - __builtin_clzll(bd);
        }
        // code lengths as they are (no run-length symbols 16-18: one code per length)
        if (x.lane < 32) cl_freq[x.lane] = 0;
        x.wsync();
        for (int base = 0; base < hlit + hdist; base += 64) {
            const int i = base + x.lane;
            if (i < hlit + hdist) x.lds_add(cl_freq + (i < hlit ? clen_all[i] : clen_all[DSYM + i - hlit]), 1);
        }
        x.wsync();
        {   // the code-length code must be complete even when one symbol does all the work: a second, unused code
            const uint64_t used = x.ballot(x.lane < 19 && cl_freq[x.lane] != 0);
            x.wsync();
            if (popc64(used) == 1 && x.lane == 0) cl_freq[(used & 1ull) ? 1 : 0] = 1;
            x.wsync();
        }
        if (!DFL_LENGTHS(x, cl_freq, 19, 7, cl_len, alias + A_HUF) && x.lane == 0) misc[M_FAIL] = ST_HUFF;
        huff_codes(x, cl_len, 19, cl_code);
        int hclen = 4;
        {
            const uint64_t b = x.ballot(x.lane < 19 && cl_len[cl_order(x.lane)] != 0);
            if (b) hclen = 64 - __builtin_clzll(b);
            if (hclen < 4) hclen = 4;
            BitSink hs;
            sink_open(x, hs, (uint32_t*)(alias + A_HDR), out, 0);
            // BFINAL = 1, BTYPE = 2, HLIT, HDIST, HCLEN, then the code-length code's lengths in `order`
            uint64_t bits = 0; uint32_t nb = 0;
            if (x.lane == 0) { bits = 1u | (2u << 1) | ((uint32_t)(hlit - 257) << 3) | ((uint32_t)(hdist - 1) << 8) | ((uint32_t)(hclen - 4) << 13); nb = 17; }
            else if (x.lane <= hclen) { bits = cl_len[cl_order(x.lane - 1)]; nb = 3; }
            sink_put(x, hs, bits, nb);
            uint32_t hb = 17 + 3 * (uint32_t)hclen;
            for (int base = 0; base < hlit + hdist; base += 64) {
                const int i = base + x.lane;
                bits = 0; nb = 0;
                if (i < hlit + hdist) { const uint32_t l = i < hlit ? clen_all[i] : clen_all[DSYM + i - hlit]; bits = cl_code[l]; nb = cl_len[l]; }
                hb += x.bcast(x.incl_scan(nb), 63);
                sink_put(x, hs, bits, nb);
            }
            sink_close(x, hs);
            if (x.lane == 0) { misc[M_HDRBITS] = hb; misc[M_HLIT] = (uint32_t)hlit; misc[M_HDIST] = (uint32_t)hdist; }
        }
    }
    x.sync();

    DFL_MARK(6);
    // ---- phase 5: every slice's bit length
    {
        uint32_t bits = 0;
        for (uint32_t i = (uint32_t)x.lane; i < NSYM; i += 64) bits += hist[i] * ((uint32_t)clen_all[i] + sym_extra_bits(i));
        const uint32_t t = x.bcast(x.incl_scan(bits), 63);
        if (x.lane == 0) misc[M_BITS + x.wave] = t;
    }
    x.sync();
    uint32_t bit0 = misc[M_HDRBITS];
    for (int w = 0; w < x.wave; ++w) bit0 += misc[M_BITS + w];
    uint32_t total_bits = misc[M_HDRBITS];
    for (int w = 0; w < NW; ++w) total_bits += misc[M_BITS + w];
    total_bits += clen_all[256];
    const uint32_t cbytes = (total_bits + 7) >> 3;
    const bool fits = cbytes <= MAXN - 26 && misc[M_FAIL] == 0;      // bgzf.c: a member is at most 64 KiB with its 26 bytes of envelope

    DFL_MARK(7);
    // ---- phase 6: the slices' bits, in parallel at their final places
    if (fits) {
        BitSink s;
        sink_open(x, s, (uint32_t*)(alias + A_STAGE) + (size_t)x.wave * 128, out, bit0);
        for (uint32_t base = 0; base < ntok; base += 64) {
            uint64_t bits = 0; uint32_t nb = 0;
            if (base + (uint32_t)x.lane < ntok) {
                const uint32_t t = tk[base + (uint32_t)x.lane];
                if (!(t >> 31)) { bits = code_all[t]; nb = clen_all[t]; }
                else {
                    uint32_t li, ln, le, di, dn, de;
                    len_sym(((t >> 16) & 0xFFu) + 3u, li, ln, le); dist_sym((t & 0x7FFFu) + 1u, di, dn, de);
                    const uint32_t ll = clen_all[257 + li], dl = clen_all[DSYM + di];
                    bits = (uint64_t)code_all[257 + li] | ((uint64_t)le << ll);
                    nb = ll + ln;
                    bits |= ((uint64_t)code_all[DSYM + di] | ((uint64_t)de << dl)) << nb;
                    nb += dl + dn;
                }
            }
            sink_put(x, s, bits, nb);
        }
        if (x.wave == NW - 1) sink_put(x, s, x.lane == 0 ? code_all[256] : 0, x.lane == 0 ? clen_all[256] : 0);
        sink_close(x, s);
    }
    x.sync();
    DFL_MARK(8);
    if (x.tid == 0) { result[0] = cbytes; result[1] = misc[M_CRC]; result[2] = misc[M_FAIL] ? misc[M_FAIL] : (uint32_t)(fits ? ST_OK : ST_TOO_BIG); }
#undef DFL_MARK
}

}  // namespace dfl
