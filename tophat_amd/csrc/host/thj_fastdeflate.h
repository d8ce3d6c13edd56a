// thj_fastdeflate.h -- a small, fast DEFLATE (RFC 1951) compressor for the BGZF members of the BAM writer.
//
// Why it exists: with ingest and the stitch kernels on the GPU, long_spanning_reads' wall clock on 16 CPUs is zlib: level 1 runs at
// ~100 MB/s per core on BAM records and a run writes ~1 GB of them per side (THJ_TRACE: 8.7 of ~11 core-seconds).  BGZF needs a
// valid raw DEFLATE stream per member, not zlib's: this one is a greedy single-probe LZ77 (4-byte hash, 32 K-entry table, matches
// extended eight bytes at a time) followed by ONE dynamic-Huffman block per member (length-limited codes from the member's own
// symbol counts), ~4x zlib-1's speed at about its ratio on this data.  Any inflater reads the result (samtools, zlib, the device
// inflate kernel); tests/test_hostio_cpu.py round-trips it through zlib.  THJ_BGZF_LEVEL set to anything selects zlib instead.
// Nothing here follows a reference file: TopHat links samtools' bgzf.c, which calls zlib.
#pragma once
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>

namespace thjh { namespace fdz {

struct BitWriter {
    uint8_t* p; uint8_t* end; uint64_t acc = 0; int n = 0; bool ovf = false;
    inline void put(uint32_t v, int bits) {                     // bits <= 32, LSB first
        acc |= (uint64_t)v << n; n += bits;
        if (n >= 32) {
            if (p + 4 <= end) { uint32_t w = (uint32_t)acc; memcpy(p, &w, 4); p += 4; } else ovf = true;
            acc >>= 32; n -= 32;
        }
    }
    inline void finish() { while (n > 0) { if (p < end) *p++ = (uint8_t)acc; else ovf = true; acc >>= 8; n -= 8; } n = 0; }
};

// Code lengths (<= maxbits) for n symbols from their counts.  A symbol with count 0 gets length 0; a lone symbol gets length 1
// (an incomplete code, which inflaters accept for the literal/length and distance alphabets only -- the caller pads the
// code-length alphabet).  Returns false if the length-limited code could not be made complete (the caller then leaves the member to zlib).
inline bool huff_lengths(const uint32_t* freq, int n, int maxbits, uint8_t* len) {
    int idx[288]; int m = 0;
    for (int i = 0; i < n; ++i) { len[i] = 0; if (freq[i]) idx[m++] = i; }
    if (m == 0) return true;
    if (m == 1) { len[idx[0]] = 1; return true; }
    std::sort(idx, idx + m, [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
    // two-queue Huffman: leaves 0..m-1 (sorted), internal nodes m..2m-2 in creation order (their weights are non-decreasing)
    uint64_t w[576]; int parent[576];
    for (int i = 0; i < m; ++i) w[i] = freq[idx[i]];
    int leaf = 0, inode = m, next = m;
    auto pick = [&]() { if (leaf < m && (inode >= next || w[leaf] <= w[inode])) return leaf++; return inode++; };
    while (next < 2 * m - 1) {
        const int a = pick(), b = pick();
        w[next] = w[a] + w[b]; parent[a] = next; parent[b] = next; ++next;
    }
    int depth[576];
    depth[2 * m - 2] = 0;
    for (int i = 2 * m - 3; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
    int maxd = 0;
    for (int i = 0; i < m; ++i) maxd = std::max(maxd, depth[i]);
    if (maxd > maxbits) {
        // clamp, then lengthen the rarest symbols that still can be until the Kraft sum fits, then give slack back to the commonest
        uint64_t kraft = 0; const uint64_t one = 1ull << maxbits;
        for (int i = 0; i < m; ++i) { if (depth[i] > maxbits) depth[i] = maxbits; kraft += one >> depth[i]; }
        while (kraft > one) {
            bool moved = false;
            for (int i = 0; i < m && kraft > one; ++i)                     // rarest first
                if (depth[i] < maxbits) { kraft -= (one >> depth[i]) - (one >> (depth[i] + 1)); ++depth[i]; moved = true; if (kraft <= one) break; }
            if (!moved) break;
        }
        for (int i = m - 1; i >= 0; --i)                                     // commonest first
            while (depth[i] > 1 && kraft + (one >> depth[i]) <= one) { kraft += one >> depth[i]; --depth[i]; }
        if (kraft != one) return false;
    }
    for (int i = 0; i < m; ++i) len[idx[i]] = (uint8_t)depth[i];
    return true;
}

// canonical codes, bit-reversed (DEFLATE packs Huffman codes MSB first into an LSB-first bit stream)
inline void huff_codes(const uint8_t* len, int n, uint16_t* code) {
    int cnt[16] = {0}, nextc[16];
    for (int i = 0; i < n; ++i) cnt[len[i]]++;
    cnt[0] = 0;
    int c = 0;
    for (int b = 1; b < 16; ++b) { c = (c + cnt[b - 1]) << 1; nextc[b] = c; }
    for (int i = 0; i < n; ++i) {
        const int l = len[i];
        if (!l) { code[i] = 0; continue; }
        uint32_t v = (uint32_t)nextc[l]++, r = 0;
        for (int k = 0; k < l; ++k) { r = (r << 1) | (v & 1u); v >>= 1; }
        code[i] = (uint16_t)r;
    }
}

struct Tables {
    uint8_t len_code[256]; uint8_t len_extra[256]; uint16_t len_base[256];     // by match length - 3: code - 257, extra bit count, base
    uint8_t dist_code_lo[256], dist_code_hi[256];                             // zlib's d_code split: (d-1) < 256 / (d-1) >> 7
    uint8_t dist_extra[30]; uint16_t dist_base[30];
    Tables() {
        static const uint16_t LB[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t LE[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        static const uint16_t DB[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t DE[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        for (int l = 3; l <= 258; ++l) {
            int c = 28;
            while (LB[c] > l) --c;
            if (l == 258) c = 28;
            len_code[l - 3] = (uint8_t)c; len_extra[l - 3] = LE[c]; len_base[l - 3] = LB[c];
        }
        for (int k = 0; k < 30; ++k) { dist_extra[k] = DE[k]; dist_base[k] = DB[k]; }
        for (int d = 1; d <= 32768; ++d) {
            int c = 29;
            while (DB[c] > d) --c;
            if (d - 1 < 256) dist_code_lo[d - 1] = (uint8_t)c;
            dist_code_hi[(d - 1) >> 7] = (uint8_t)c;               // constant within each 128-wide bucket for d - 1 >= 256
        }
    }
    inline int dcode(uint32_t d) const { const uint32_t x = d - 1; return x < 256 ? dist_code_lo[x] : dist_code_hi[x >> 7]; }
};
inline const Tables& tables() { static const Tables t; return t; }

static inline uint32_t load32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// One raw DEFLATE stream (a single final dynamic-Huffman block) for in[0..n), n <= 65536.  False when it does not fit `cap` bytes.
inline bool deflate_fast(const uint8_t* in, size_t n, uint8_t* out, size_t cap, size_t* out_len) {
    if (n > 65536) return false;
    const Tables& T = tables();
    static constexpr int HB = 15;
    thread_local std::vector<uint16_t> tab_v(1u << HB);
    thread_local std::vector<uint32_t> sym_v(65536 + 8);
    uint16_t* tab = tab_v.data();
    uint32_t* sym = sym_v.data();
    memset(tab, 0, sizeof(uint16_t) << HB);
    uint32_t lfreq[286] = {0}, dfreq[30] = {0};
    size_t ns = 0, i = 0;
    // ---- greedy LZ77, one probe per position
    if (n >= 8) {
        const size_t last = n - 8;                 // every 4-byte load and the first 8-byte compare stay inside the input
        while (i <= last) {
            const uint32_t v = load32(in + i);
            const uint32_t h = (v * 2654435761u) >> (32 - HB);
            const size_t cand = tab[h];
            tab[h] = (uint16_t)i;
            const size_t dist = i - cand;
            if (dist - 1 < 32768 && load32(in + cand) == v) {
                const size_t maxl = std::min<size_t>(258, n - i);
                size_t l = 4;
                while (l + 8 <= maxl) {
                    const uint64_t x = load64(in + cand + l) ^ load64(in + i + l);
                    if (x) { l += (size_t)(__builtin_ctzll(x) >> 3); goto matched; }
                    l += 8;
                }
                while (l < maxl && in[cand + l] == in[i + l]) ++l;
            matched:
                sym[ns++] = 0x80000000u | ((uint32_t)(l - 3) << 16) | (uint32_t)(dist - 1);
                lfreq[257 + T.len_code[l - 3]]++;
                dfreq[T.dcode((uint32_t)dist)]++;
                // keep the table warm inside the match: two more positions (enough for records that repeat field by field)
                if (i + 2 <= last) {
                    tab[(load32(in + i + 1) * 2654435761u) >> (32 - HB)] = (uint16_t)(i + 1);
                    tab[(load32(in + i + 2) * 2654435761u) >> (32 - HB)] = (uint16_t)(i + 2);
                }
                i += l;
            } else {
                sym[ns++] = in[i]; lfreq[in[i]]++; ++i;
            }
        }
    }
    for (; i < n; ++i) { sym[ns++] = in[i]; lfreq[in[i]]++; }
    lfreq[256] = 1;
    // ---- codes
    uint8_t llen[286], dlen[30];
    uint16_t lcode[286], dcode[30];
    if (!huff_lengths(lfreq, 286, 15, llen) || !huff_lengths(dfreq, 30, 15, dlen)) return false;
    int hlit = 286; while (hlit > 257 && llen[hlit - 1] == 0) --hlit;
    int hdist = 30; while (hdist > 1 && dlen[hdist - 1] == 0) --hdist;
    huff_codes(llen, 286, lcode);
    huff_codes(dlen, 30, dcode);
    // code lengths, run-length coded with the code-length alphabet (RFC 1951, 3.2.7)
    uint8_t seq[316]; int nseq = 0;
    for (int k = 0; k < hlit; ++k) seq[nseq++] = llen[k];
    for (int k = 0; k < hdist; ++k) seq[nseq++] = dlen[k];
    uint8_t cl_sym[320], cl_ext[320]; int ncl = 0;
    uint32_t cfreq[19] = {0};
    for (int k = 0; k < nseq;) {
        const int v = seq[k]; int run = 1;
        while (k + run < nseq && seq[k + run] == v) ++run;
        int left = run;
        if (v == 0) {
            while (left >= 11) { const int r = std::min(left, 138); cl_sym[ncl] = 18; cl_ext[ncl++] = (uint8_t)(r - 11); cfreq[18]++; left -= r; }
            if (left >= 3) { cl_sym[ncl] = 17; cl_ext[ncl++] = (uint8_t)(left - 3); cfreq[17]++; left = 0; }
            while (left-- > 0) { cl_sym[ncl] = 0; cl_ext[ncl++] = 0; cfreq[0]++; }
        } else {
            cl_sym[ncl] = (uint8_t)v; cl_ext[ncl++] = 0; cfreq[v]++; --left;
            while (left >= 3) { const int r = std::min(left, 6); cl_sym[ncl] = 16; cl_ext[ncl++] = (uint8_t)(r - 3); cfreq[16]++; left -= r; }
            while (left-- > 0) { cl_sym[ncl] = (uint8_t)v; cl_ext[ncl++] = 0; cfreq[v]++; }
        }
        k += run;
    }
    uint8_t clen[19]; uint16_t ccode[19];
    {   // the code-length code must be complete even when one symbol does all the work: a second, unused one-bit code
        int used = 0, lone = 0;
        for (int k = 0; k < 19; ++k) if (cfreq[k]) { ++used; lone = k; }
        if (used == 1) cfreq[lone == 0 ? 1 : 0] = 1;
    }
    if (!huff_lengths(cfreq, 19, 7, clen)) return false;
    huff_codes(clen, 19, ccode);
    static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    int hclen = 19; while (hclen > 4 && clen[CLORD[hclen - 1]] == 0) --hclen;
    // ---- the block
    BitWriter bw; bw.p = out; bw.end = out + cap;
    bw.put(1, 1); bw.put(2, 2);                                  // BFINAL, BTYPE = dynamic
    bw.put((uint32_t)(hlit - 257), 5); bw.put((uint32_t)(hdist - 1), 5); bw.put((uint32_t)(hclen - 4), 4);
    for (int k = 0; k < hclen; ++k) bw.put(clen[CLORD[k]], 3);
    for (int k = 0; k < ncl; ++k) {
        const int s = cl_sym[k];
        bw.put(ccode[s], clen[s]);
        if (s == 16) bw.put(cl_ext[k], 2); else if (s == 17) bw.put(cl_ext[k], 3); else if (s == 18) bw.put(cl_ext[k], 7);
    }
    for (size_t k = 0; k < ns; ++k) {
        const uint32_t s = sym[k];
        if (!(s & 0x80000000u)) { bw.put(lcode[s], llen[s]); continue; }
        const uint32_t l3 = (s >> 16) & 0xFFu, d = (s & 0x7FFFu) + 1u;
        const int lc = T.len_code[l3], dc = T.dcode(d);
        // length code + its extra bits in one go (<= 15 + 5 bits), then distance code + extra (<= 15 + 13)
        bw.put((uint32_t)lcode[257 + lc] | ((l3 + 3u - T.len_base[l3]) << llen[257 + lc]), llen[257 + lc] + T.len_extra[l3]);
        bw.put((uint32_t)dcode[dc] | ((d - T.dist_base[dc]) << dlen[dc]), dlen[dc] + T.dist_extra[dc]);
        if (bw.ovf) return false;
    }
    bw.put(lcode[256], llen[256]);
    bw.finish();
    if (bw.ovf) return false;
    *out_len = (size_t)(bw.p - out);
    return true;
}

}}  // namespace thjh::fdz
