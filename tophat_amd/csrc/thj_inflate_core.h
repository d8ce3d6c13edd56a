// thj_inflate_core.h -- the per-member logic of the two-kernel BGZF inflater (replaces what
// samtools-0.1.18/bgzf.c:inflate_block hands to zlib).  DEFLATE (RFC 1951) splits cleanly in two:
//
//   1. entropy decoding -- a chain of dependent table look-ups, serial inside a member, that never looks at the bytes it
//      produced: thj_k_huff runs ONE MEMBER PER LANE (64 members per wave, every lane busy: the round-2 kernel decoded on one lane
//      of 64 and was bound by the issue latency of a single wave) and writes a stream of 32-bit tokens, literal or (length, distance);
//   2. LZ77 resolution -- copies inside the member's own output, parallel except where a match reads what an earlier one of the
//      same batch writes: thj_k_lz runs ONE WAVE PER MEMBER over 64 tokens at a time (prefix sum of the lengths = output
//      positions, literals scattered, matches copied in rounds behind a high-water mark) with the last 32 KiB of output in LDS.
//
// This header is the lane logic of kernel 1, written so that tests/hostsim can compile it for the CPU (one lane at a time) and
// check it against zlib before it ever runs on a GPU.  Everything a member needs sits in a private LDS slice:
//
//   lit[852]   u16  two-level literal/length table: 9-bit root + sub-tables (zlib's inftrees.h bound ENOUGH_LENS for a 9-bit root)
//   A[320]     u8   code lengths while a header is parsed; afterwards the 8-bit distance root table in A[0..256)
//   B[64]           distance codes longer than 8 bits: per length {first code, count, list base} + the symbols in canonical order
//   C[64]      u16  count / next-code per length while tables are built
//   ring[256]  u8   compressed input, topped up 16 bytes at a time at uniform points of the loop (loads land one top-up later)
//   stage[8]   u32  the newest tokens; they leave for HBM four at a time (16-byte stores) at the same uniform points
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define THJ_IHD __host__ __device__ __forceinline__
#else
#define THJ_IHD inline
#endif

namespace inf2 {

constexpr int ROOT = 9, ROOT_SIZE = 1 << ROOT, LIT_ENTRIES = 852;
constexpr int DROOT = 8, DROOT_SIZE = 1 << DROOT;
constexpr int RING = 256;
constexpr int STAGE = 8;                                   // tokens staged per member before they leave 16 bytes at a time
constexpr int OFF_A = LIT_ENTRIES * 2, OFF_B = OFF_A + 320, OFF_C = OFF_B + 64, OFF_RING = OFF_C + 64, OFF_STAGE = OFF_RING + RING, LANE_BYTES = OFF_STAGE + STAGE * 4;   // 2440
constexpr int STRIDE_WORDS = 611;                          // >= LANE_BYTES / 4, odd: the lanes' slices start in different banks
static_assert(STRIDE_WORDS * 4 >= LANE_BYTES && (STRIDE_WORDS & 1), "LDS slice");
constexpr uint32_t TOKCAP = 20480;                         // tokens kept per member (zlib closes a block at 16383 symbols; typical BAM members: 11-14 k)
constexpr uint32_t NTOK_FALLBACK = 0xFFFFFFFFu;            // ntok[]: this member goes to the one-lane kernel (stored blocks, too many tokens, corrupt streams)

// token: literal = byte; match = 1 << 31 | (len - 3) << 15 | (dist - 1)
THJ_IHD uint32_t tok_match(uint32_t len, uint32_t dist) { return 0x80000000u | ((len - 3u) << 15) | (dist - 1u); }

// lit[] entry: bits 0..3 = n (code bits this level consumes; 0 = no such code), bits 4..15 = payload:
//   < 256 literal | 256 end of block | 512 + i: sub-table at lit[i], n = its index bits | 0x800 | extra << 8 | (base - 3): a length code
enum { P_EOB = 256, P_SUB = 512, P_LEN = 0x800 };
// A[] (distance root) entry: symbol | (code length - 1) << 5; 0xFF = longer than 8 bits or no such code
enum { ST_HEADER = 0, ST_DECODE = 1, ST_DONE = 2, ST_FALLBACK = 3 };

struct Lane {
    uint64_t buf; int cnt;                 // bit buffer (next bit = bit 0)
    uint32_t nextw;                        // the ring word in front of rd, not yet in buf
    uint32_t rd;                           // read position in the aligned stream (bytes, multiple of 4)
    uint32_t ld;                           // bytes of the aligned stream landed in the ring (multiple of 16)
    uint32_t total;                        // skew + compressed length: where the member's bytes end in the aligned stream
    uint32_t outp, ntok, nflushed;         // tokens made / tokens already in HBM (multiple of 4 until the end)
    int state, last;
    bool inflight; uint32_t pend[4];       // a 16-byte piece on its way
    const uint8_t* src;                    // the aligned stream (16-byte aligned address at or before the member's first byte)
    uint16_t* lit; uint8_t* A; uint8_t* B; uint16_t* C; uint32_t* ring; uint32_t* stage;
    uint32_t* tok;
};

THJ_IHD uint32_t rev_bits(uint32_t v, int n) {             // the low n bits of v reversed, 1 <= n <= 15
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32 - n);
#else
    uint32_t r = 0; for (int i = 0; i < n; ++i) r |= ((v >> i) & 1u) << (n - 1 - i); return r;
#endif
}

// ---- input
THJ_IHD void topup(Lane& L) {                               // uniform points only: lands the piece asked for at the previous point, asks for the next
    if (L.inflight) {
        uint32_t* r = L.ring + ((L.ld & (RING - 1)) >> 2);
        r[0] = L.pend[0]; r[1] = L.pend[1]; r[2] = L.pend[2]; r[3] = L.pend[3];
        L.ld += 16; L.inflight = false;
    }
    if (L.ld < L.total && L.ld + 16 - (L.rd - 4) <= (uint32_t)RING) {       // the word at rd - 4 (nextw) has been read already, but keep it simple and safe
        uint32_t v[4];
        memcpy(v, L.src + L.ld, 16);
        L.pend[0] = v[0]; L.pend[1] = v[1]; L.pend[2] = v[2]; L.pend[3] = v[3];
        L.inflight = true;
    }
}
THJ_IHD void flush_tokens(Lane& L) {                        // uniform points only, before topup(): the store is older than the load topup() issues
    if (L.ntok - L.nflushed >= 4u) {
        const uint32_t* st = L.stage + (L.nflushed & (STAGE - 1));
        uint32_t v[4] = {st[0], st[1], st[2], st[3]};
        memcpy(L.tok + L.nflushed, v, 16);
        L.nflushed += 4;
    }
}
THJ_IHD void flush_tokens_end(Lane& L) {
    flush_tokens(L);
    for (; L.nflushed < L.ntok; ++L.nflushed) L.tok[L.nflushed] = L.stage[L.nflushed & (STAGE - 1)];
}
THJ_IHD bool input_ok(const Lane& L) { return L.rd + 8 <= L.ld || L.ld >= L.total; }       // two more ring words may be read (or all there is has landed)
THJ_IHD void refill(Lane& L) {
    if (L.cnt <= 32) { L.buf |= (uint64_t)L.nextw << L.cnt; L.cnt += 32; L.nextw = L.ring[(L.rd & (RING - 1)) >> 2]; L.rd += 4; }
}
// the same without a branch (the decode loop): lanes that are not live keep their state; the ring read always happens, its address is always valid
THJ_IHD void refill_bf(Lane& L, bool live) {
    const bool need = live && L.cnt <= 32;
    const uint32_t w = L.ring[(L.rd & (RING - 1)) >> 2];
    L.buf |= need ? (uint64_t)L.nextw << (L.cnt & 63) : 0ull;
    L.cnt += need ? 32 : 0; L.rd += need ? 4u : 0u;
    L.nextw = need ? w : L.nextw;
}
THJ_IHD uint32_t bfe(uint32_t x, uint32_t off, uint32_t width) { return (x >> off) & ((1u << width) - 1u); }
THJ_IHD uint32_t take(Lane& L, int n) { const uint32_t v = (uint32_t)L.buf & ((1u << n) - 1u); L.buf >>= n; L.cnt -= n; return v; }
THJ_IHD void lane_start(Lane& L, uint32_t skew) {          // after the ring's first pieces have landed
    L.rd = skew & ~3u;
    const uint32_t w = L.ring[(L.rd & (RING - 1)) >> 2]; L.rd += 4;
    L.buf = (uint64_t)(w >> (8 * (skew & 3u))); L.cnt = 32 - 8 * (int)(skew & 3u);
    L.nextw = L.ring[(L.rd & (RING - 1)) >> 2]; L.rd += 4;
}
// bits of the member consumed so far must not exceed what it has (a stream that runs off its end decodes the zero / stale padding)
THJ_IHD bool overrun(const Lane& L) { return (int32_t)((L.rd - 4u) * 8u) - L.cnt > (int32_t)(L.total * 8u); }      // members are at most 64 KiB: 32 bits do

// ---- length / distance arithmetic (RFC 1951, 3.2.5) without tables
THJ_IHD uint32_t len_payload(uint32_t k) {                  // k = symbol - 257; 0 = no such code (286, 287)
    if (k >= 29u) return 0;
    uint32_t ext, base;
    if (k < 8u) { ext = 0; base = 3u + k; } else if (k == 28u) { ext = 0; base = 258u; } else { ext = (k - 4u) >> 2; base = 3u + ((4u + (k & 3u)) << ext); }
    return (uint32_t)P_LEN | (ext << 8) | (base - 3u);
}
THJ_IHD void dist_base_ext(uint32_t ds, uint32_t& base, uint32_t& ext) {
    ext = ds < 2u ? 0u : (ds >> 1) - 1u;
    base = ds < 2u ? 1u + ds : 1u + ((2u + (ds & 1u)) << ext);
}

// ---- tables.  Every loop below runs in lock step over the lanes of a wave: trip counts are uniform or bounded by a small maximum.
// counts per length -> C[0..16); returns false when over-subscribed
THJ_IHD bool count_lengths(Lane& L, const uint8_t* lens, int n) {
    for (int l = 0; l < 16; ++l) L.C[l] = 0;
    for (int i = 0; i < n; ++i) L.C[lens[i]]++;
    int left = 1; bool ok = true;
    for (int l = 1; l < 16; ++l) { left = (left << 1) - (int)L.C[l]; ok = ok && left >= 0; }
    return ok;
}
// first canonical code per length -> C[16..32)
THJ_IHD void first_codes(Lane& L) {
    uint32_t code = 0;
    for (int l = 1; l < 16; ++l) { L.C[16 + l] = (uint16_t)code; code = (code + L.C[l]) << 1; }
}

// literal/length table from lens[0..n), n <= 288 (symbols 286 / 287 of the fixed code get codes but no entries: no stream may use them)
template <class W>
THJ_IHD bool build_lit(Lane& L, const uint8_t* lens, int n, bool live, const W& wave) {
    bool ok = count_lengths(L, lens, live ? n : 0);
    first_codes(L);
    for (int i = 0; i < LIT_ENTRIES; ++i) L.lit[i] = 0;
    // sub-tables: the codes longer than 9 bits in canonical order; codes that share their first 9 bits are neighbours there, and
    // the sub-table of such a run is indexed by as many bits as its longest code has beyond the root
    uint32_t used = ROOT_SIZE; int cur_prefix = -1, cur_max = 0;
    for (int l = ROOT + 1; l < 16; ++l) {
        const uint32_t first = L.C[16 + l], c = live ? L.C[l] : 0u;
        for (uint32_t k = 0; wave.any(k < c); ++k) {
            if (k < c) {
                const int prefix = (int)((first + k) >> (l - ROOT));
                if (prefix != cur_prefix) {
                    if (cur_prefix >= 0) { const int b = cur_max - ROOT; if (used + (1u << b) > (uint32_t)LIT_ENTRIES) ok = false; else { L.lit[rev_bits((uint32_t)cur_prefix, ROOT)] = (uint16_t)(((uint32_t)P_SUB + used) << 4 | (uint32_t)b); used += 1u << b; } }
                    cur_prefix = prefix;
                }
                cur_max = l;
            }
        }
    }
    if (cur_prefix >= 0) { const int b = cur_max - ROOT; if (used + (1u << b) > (uint32_t)LIT_ENTRIES) ok = false; else { L.lit[rev_bits((uint32_t)cur_prefix, ROOT)] = (uint16_t)(((uint32_t)P_SUB + used) << 4 | (uint32_t)b); used += 1u << b; } }
    // the symbols in order: canonical code = next code of its length
    for (int s = 0; wave.any(s < n); ++s) {
        const int l = (live && s < n) ? lens[s] : 0;
        uint32_t f = 0, end = 0, step = 1, e = 0; uint16_t* t = L.lit;
        if (l) {
            const uint32_t code = L.C[16 + l]; L.C[16 + l] = (uint16_t)(code + 1);
            const uint32_t payload = s < 256 ? (uint32_t)s : s == 256 ? (uint32_t)P_EOB : len_payload((uint32_t)s - 257u);
            if (l <= ROOT) { f = rev_bits(code, l); end = ROOT_SIZE; step = 1u << l; e = payload << 4 | (uint32_t)l; }
            else {
                const uint32_t r = L.lit[rev_bits(code >> (l - ROOT), ROOT)];
                const int j = l - ROOT, b = (int)(r & 15u);
                if ((r >> 4) >= (uint32_t)P_SUB && (r >> 4) < (uint32_t)P_LEN && j <= b) { t = L.lit + ((r >> 4) - (uint32_t)P_SUB); f = rev_bits(code & ((1u << j) - 1u), j); end = 1u << b; step = 1u << j; e = payload << 4 | (uint32_t)j; }
                else ok = false;
            }
            if (payload == 0 && s > 256) end = 0;                            // 286 / 287: leave the entries empty
        }
        for (; wave.any(f < end); f += step) if (f < end) t[f] = (uint16_t)e;
    }
    return ok;
}

// distance tables from dl[0..n), n <= 32 (the fixed code has 32 five-bit codes; symbols 30 / 31 are looked up and refused when used)
template <class W>
THJ_IHD bool build_dist(Lane& L, const uint8_t* dl, int n, bool live, const W& wave) {
    // dl = A + hlit with hlit >= 257: the lengths sit beyond the 256 bytes of the root table built below
    for (int l = 0; l < 16; ++l) L.C[l] = 0;
    for (int i = 0; i < 32; ++i) L.C[(live && i < n) ? dl[i] : 0]++;
    int left = 1; bool ok = true;
    for (int l = 1; l < 16; ++l) { left = (left << 1) - (int)L.C[l]; ok = ok && left >= 0; }
    first_codes(L);
    for (int i = 0; i < DROOT_SIZE; ++i) L.A[i] = 0xFF;
    // long codes: B as u32[7] {first code : 15 | count : 6 << 15 | list base : 6 << 21} for lengths 9..15, then u8[32] symbols
    uint32_t* bl = (uint32_t*)L.B; uint8_t* bsym = L.B + 28;
    uint32_t lbase[16]; { uint32_t acc = 0; for (int l = DROOT + 1; l < 16; ++l) { lbase[l] = acc; bl[l - DROOT - 1] = (uint32_t)L.C[16 + l] | (uint32_t)L.C[l] << 15 | acc << 21; acc += L.C[l]; } }
    for (int s = 0; s < 32; ++s) {
        const int l = (live && s < n) ? dl[s] : 0;
        uint32_t f = 0, end = 0, step = 1, e = 0;
        if (l) {
            const uint32_t code = L.C[16 + l]; L.C[16 + l] = (uint16_t)(code + 1);
            if (l <= DROOT) { f = rev_bits(code, l); end = DROOT_SIZE; step = 1u << l; e = (uint32_t)s | (uint32_t)(l - 1) << 5; }
            else { uint32_t lb = 0;
#pragma unroll
                   for (int q = DROOT + 1; q < 16; ++q) lb = q == l ? lbase[q] : lb;
                   const uint32_t firstc = bl[l - DROOT - 1] & 0x7FFFu; bsym[lb + (code - firstc)] = (uint8_t)s; }
        }
        for (; wave.any(f < end); f += step) if (f < end) L.A[f] = (uint8_t)e;
    }
    return ok;
}

// a distance code of more than 8 bits (the root said 0xFF): canonical decode over lengths 9..15.  Returns the symbol or -1; n = its length.
// Straight-line: the seven per-length words are read together, the shortest length whose code range holds the next bits is selected.
THJ_IHD int dist_long_raw(const uint8_t* Bt, uint32_t lo, int& n) {
    const uint32_t* bl = (const uint32_t*)Bt; const uint8_t* bsym = Bt + 28;
    uint32_t w[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) w[q] = bl[q];
    const uint32_t c15 = rev_bits(lo & 0x7FFFu, 15);                  // the next 15 bits, first bit most significant
    uint32_t idx = 0; n = 0;
#pragma unroll
    for (int q = 0; q < 7; ++q) {
        const int l = DROOT + 1 + q;
        const uint32_t d = (c15 >> (15 - l)) - (w[q] & 0x7FFFu);
        const bool hit = n == 0 && d < ((w[q] >> 15) & 63u);
        idx = hit ? (w[q] >> 21) + d : idx; n = hit ? l : n;
    }
    const int sym = (int)bsym[idx & 31u];
    return n ? sym : -1;
}
THJ_IHD int dist_long(const Lane& L, int& n) { return dist_long_raw(L.B, (uint32_t)L.buf, n); }

// ---- one block header (lane in ST_HEADER): BFINAL, BTYPE, code lengths, tables.  Lock step: lanes not in ST_HEADER idle through it.
// what the first half of a header leaves for the table builders: the code lengths in A[0 .. hlit + hdist), `build` = there are tables to build
struct HeaderInfo { int hlit, hdist; bool build, ok; };
template <class W>
THJ_IHD HeaderInfo parse_header_lengths(Lane& L, const W& wave) {
    const bool live = L.state == ST_HEADER;
    int type = -1;
    if (live) { refill(L); L.last = (int)take(L, 1); type = (int)take(L, 2); }
    bool dyn = live && type == 2, fixed = live && type == 1;
    if (live && !dyn && !fixed) L.state = ST_FALLBACK;                       // stored blocks (and BTYPE 3) go to the one-lane kernel
    int hlit = 288, hdist = 32, hclen = 0;
    bool ok = true;
    if (dyn) { refill(L); hlit = (int)take(L, 5) + 257; hdist = (int)take(L, 5) + 1; hclen = (int)take(L, 4) + 4; if (hlit > 286 || hdist > 30) { ok = false; dyn = false; } }
    // the code-length code: 19 symbols of up to 7 bits, a 7-bit direct table in lit[0..128) (the real table is built afterwards)
    uint8_t* cl = L.B;                                                       // 19 lengths, scratch
    for (int i = 0; i < 19; ++i) cl[i] = 0;
    {
        static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int i = 0; i < 19; ++i) if (dyn && i < hclen) { refill(L); cl[CLORD[i]] = (uint8_t)take(L, 3); }
    }
    {
        for (int l = 0; l < 8; ++l) L.C[l] = 0;
        for (int i = 0; i < 19; ++i) L.C[cl[i]]++;
        int left = 1; for (int l = 1; l < 8; ++l) { left = (left << 1) - (int)L.C[l]; ok = ok && left >= 0; }
        uint32_t code = 0; for (int l = 1; l < 8; ++l) { L.C[16 + l] = (uint16_t)code; code = (code + L.C[l]) << 1; }
        for (int i = 0; i < 128; ++i) L.lit[i] = 0;
        for (int s = 0; s < 19; ++s) {
            const int l = cl[s]; uint32_t f = 0, end = 0, step = 1;
            if (l) { const uint32_t c2 = L.C[16 + l]; L.C[16 + l] = (uint16_t)(c2 + 1); f = rev_bits(c2, l); end = 128; step = 1u << l; }
            for (; wave.any(f < end); f += step) if (f < end) L.lit[f] = (uint16_t)((uint32_t)s << 4 | (uint32_t)l);
        }
    }
    // the hlit + hdist code lengths -> A
    {
        const int n = hlit + hdist; int i = 0, it = 0; int prev = 0;
        bool run = dyn && ok;
        while (wave.any(run && i < n)) {
            if ((it++ & 7) == 0) topup(L);
            if (run && i < n && input_ok(L)) {
                refill(L);
                const uint32_t e = L.lit[(uint32_t)L.buf & 127u];
                const int l = (int)(e & 15u), sym = (int)(e >> 4);
                if (!l) { ok = false; run = false; }
                else {
                    L.buf >>= l; L.cnt -= l;
                    if (sym < 16) { L.A[i++] = (uint8_t)sym; prev = sym; }
                    else {
                        int rep, val = 0;
                        if (sym == 16) { if (i == 0) { ok = false; run = false; } val = prev; rep = 3 + (int)take(L, 2); }
                        else if (sym == 17) { rep = 3 + (int)take(L, 3); prev = 0; }
                        else { rep = 11 + (int)take(L, 7); prev = 0; }
                        if (i + rep > n) { ok = false; run = false; rep = 0; }
                        for (int k = 0; k < rep; ++k) L.A[i + k] = (uint8_t)val;      // at most 138: divergent but short, once per run
                        i += rep;
                    }
                }
            }
        }
        if (dyn && ok && L.A[256] == 0) ok = false;
    }
    if (fixed) {
        for (int i = 0; i < 144; ++i) L.A[i] = 8;
        for (int i = 144; i < 256; ++i) L.A[i] = 9;
        for (int i = 256; i < 280; ++i) L.A[i] = 7;
        for (int i = 280; i < 288; ++i) L.A[i] = 8;
        for (int i = 0; i < 32; ++i) L.A[288 + i] = 5;
    }
    HeaderInfo hi;
    hi.hlit = hlit; hi.hdist = hdist; hi.build = (dyn && ok) || fixed; hi.ok = ok;
    return hi;
}
template <class W>
THJ_IHD void parse_header(Lane& L, const W& wave) {
    const bool live = L.state == ST_HEADER;
    const HeaderInfo hi = parse_header_lengths(L, wave);
    bool ok = hi.ok;
    ok = build_lit(L, L.A, hi.hlit, hi.build, wave) && ok;    // the fixed code counts its symbols 286 / 287 (they shape the canonical codes) and enters neither
    ok = build_dist(L, L.A + hi.hlit, hi.hdist, hi.build, wave) && ok;
    if (live && L.state == ST_HEADER) L.state = (hi.build && ok && !overrun(L)) ? ST_DECODE : ST_FALLBACK;
}

// ---- the same tables built by a WAVE for one member (thj_k_huffp).  Lane 0 has read the code lengths (parse_header_lengths); on one
// lane the tables after them -- 852 + 256 entries cleared, 286 + 30 symbols entered one after the other, each fill a loop of LDS
// stores -- were 24-42 % of a member's time (profiles/r05_c_huffp_lz_phase_clocks.txt).  Here the lanes take a symbol each: a
// symbol's canonical code is the first code of its length plus the number of earlier symbols of that length -- a ballot per length
// over 64 symbols at a time; counts, first codes and running codes are the same on every lane.  Only the sub-table layout (the few
// codes longer than the root's 9 bits, in canonical order) stays with lane 0.
// X: lane, ballot(bool) -> 64-bit mask, sync() (LDS writes of the wave visible to the wave).
template <class X>
THJ_IHD bool build_lit_wave(uint16_t* lit, uint16_t* C, const uint8_t* lens, int n, bool build, X& x) {
    const int lane = x.lane;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (!build) n = 0;
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = 0;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int sy = c0 + lane;
        const int l = sy < n ? (int)lens[sy] : 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) cnt[q] += (uint32_t)__builtin_popcountll(x.ballot(l == q));
    }
    bool ok = true;
    {
        int left = 1;
#pragma unroll
        for (int l = 1; l < 16; ++l) { left = (left << 1) - (int)cnt[l]; ok = ok && left >= 0; }
    }
    uint32_t nxt[16];
    {
        uint32_t code = 0;
        nxt[0] = 0;
#pragma unroll
        for (int l = 1; l < 16; ++l) { nxt[l] = code; code = (code + cnt[l]) << 1; }
    }
    for (int i = lane; i < LIT_ENTRIES / 2; i += 64) ((uint32_t*)lit)[i] = 0u;
    if (lane < 16) {                                   // counts and first codes where the sub-table walk below reads them
        uint32_t c = 0, f = 0;
#pragma unroll
        for (int l = 0; l < 16; ++l) { c = lane == l ? cnt[l] : c; f = lane == l ? nxt[l] : f; }
        C[lane] = (uint16_t)c; C[16 + lane] = (uint16_t)f;
    }
    x.sync();
    // sub-tables (lane 0): the codes longer than 9 bits in canonical order; codes that share their first 9 bits are neighbours there
    bool ok0 = true;
    if (lane == 0) {
        uint32_t used = ROOT_SIZE; int cur_prefix = -1, cur_max = 0;
        for (int l = ROOT + 1; l < 16; ++l) {
            const uint32_t first = C[16 + l], c = C[l];
            for (uint32_t k = 0; k < c; ++k) {
                const int prefix = (int)((first + k) >> (l - ROOT));
                if (prefix != cur_prefix) {
                    if (cur_prefix >= 0) { const int b = cur_max - ROOT; if (used + (1u << b) > (uint32_t)LIT_ENTRIES) ok0 = false; else { lit[rev_bits((uint32_t)cur_prefix, ROOT)] = (uint16_t)(((uint32_t)P_SUB + used) << 4 | (uint32_t)b); used += 1u << b; } }
                    cur_prefix = prefix;
                }
                cur_max = l;
            }
        }
        if (cur_prefix >= 0) { const int b = cur_max - ROOT; if (used + (1u << b) > (uint32_t)LIT_ENTRIES) ok0 = false; else { lit[rev_bits((uint32_t)cur_prefix, ROOT)] = (uint16_t)(((uint32_t)P_SUB + used) << 4 | (uint32_t)b); used += 1u << b; } }
    }
    x.sync();
    // the symbols, 64 at a time
    bool okl = true;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int sy = c0 + lane;
        const int l = sy < n ? (int)lens[sy] : 0;
        uint32_t code = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) {
            const uint64_t m = x.ballot(l == q);
            code = l == q ? nxt[q] + (uint32_t)__builtin_popcountll(m & below) : code;
            nxt[q] += (uint32_t)__builtin_popcountll(m);
        }
        uint32_t f = 0, end = 0, step = 1, e = 0; uint16_t* t = lit;
        if (l) {
            const uint32_t payload = sy < 256 ? (uint32_t)sy : sy == 256 ? (uint32_t)P_EOB : len_payload((uint32_t)sy - 257u);
            if (l <= ROOT) { f = rev_bits(code, l); end = ROOT_SIZE; step = 1u << l; e = payload << 4 | (uint32_t)l; }
            else {
                const uint32_t r = lit[rev_bits(code >> (l - ROOT), ROOT)];
                const int j = l - ROOT, b = (int)(r & 15u);
                if ((r >> 4) >= (uint32_t)P_SUB && (r >> 4) < (uint32_t)P_LEN && j <= b) { t = lit + ((r >> 4) - (uint32_t)P_SUB); f = rev_bits(code & ((1u << j) - 1u), j); end = 1u << b; step = 1u << j; e = payload << 4 | (uint32_t)j; }
                else okl = false;
            }
            if (payload == 0 && sy > 256) end = 0;                            // 286 / 287: leave the entries empty
        }
        for (; x.ballot(f < end); f += step) if (f < end) t[f] = (uint16_t)e;
    }
    x.sync();
    return ok && !x.ballot(!ok0 || !okl);
}
template <class X>
THJ_IHD bool build_dist_wave(uint8_t* A, uint8_t* Bt, const uint8_t* dl, int n, bool build, X& x) {
    // dl = A + hlit with hlit >= 257: the lengths sit beyond the 256 bytes of the root table built here
    const int lane = x.lane;
    const uint64_t below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (!build) n = 0;
    const int l = lane < n && lane < 32 ? (int)dl[lane] : 0;
    uint32_t cnt[16], first[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) cnt[q] = 0;
    uint32_t code = 0, rank = 0;
#pragma unroll
    for (int q = 1; q < 16; ++q) {
        const uint64_t m = x.ballot(l == q);
        cnt[q] = (uint32_t)__builtin_popcountll(m);
        rank = l == q ? (uint32_t)__builtin_popcountll(m & below) : rank;
    }
    bool ok = true;
    {
        int left = 1; uint32_t c2 = 0;
        first[0] = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) { left = (left << 1) - (int)cnt[q]; ok = ok && left >= 0; first[q] = c2; c2 = (c2 + cnt[q]) << 1; }
    }
#pragma unroll
    for (int q = 1; q < 16; ++q) code = l == q ? first[q] + rank : code;
    for (int i = lane; i < DROOT_SIZE / 4; i += 64) ((uint32_t*)A)[i] = 0xFFFFFFFFu;
    // long codes: B as u32[7] {first code : 15 | count : 6 << 15 | list base : 6 << 21} for lengths 9..15, then u8[32] symbols
    uint32_t* bl = (uint32_t*)Bt; uint8_t* bsym = Bt + 28;
    uint32_t lbase[16];
    {
        uint32_t acc = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) lbase[q] = 0;
#pragma unroll
        for (int q = DROOT + 1; q < 16; ++q) { lbase[q] = acc; if (lane == q) bl[q - DROOT - 1] = (first[q] & 0xFFFFu) | cnt[q] << 15 | acc << 21;   /* (the first code as the one-lane builder's u16 holds it) */ acc += cnt[q]; }
    }
    x.sync();
    uint32_t f = 0, end = 0, step = 1, e = 0;
    if (l) {
        if (l <= DROOT) { f = rev_bits(code, l); end = DROOT_SIZE; step = 1u << l; e = (uint32_t)lane | (uint32_t)(l - 1) << 5; }
        else {
            uint32_t lb = 0, fc = 0;
#pragma unroll
            for (int q = DROOT + 1; q < 16; ++q) { lb = q == l ? lbase[q] : lb; fc = q == l ? (first[q] & 0x7FFFu) : fc; }
            bsym[lb + (code - fc)] = (uint8_t)lane;
        }
    }
    for (; x.ballot(f < end); f += step) if (f < end) A[f] = (uint8_t)e;
    x.sync();
    return ok;
}

// ---- one symbol (live = lane in ST_DECODE with input_ok).  Straight-line: every lane computes the match path (a literal lane's
// distance look-up reads some entry of its own table and is thrown away), results are selected at the end -- no divergent branches
// for the compiler to serialise, and the look-ups of a step overlap.  Only the rare paths (second table level, distance codes
// beyond the root table) are branches, taken when some lane of the wave needs them.
template <class W>
THJ_IHD void decode_one(Lane& L, bool live, const W& wave) {
    refill_bf(L, live);
    const uint32_t lo = (uint32_t)L.buf;
    const uint32_t e = L.lit[lo & (ROOT_SIZE - 1)];
    uint32_t n = e & 15u, p = e >> 4, nl = n;
    const bool sub = (p - (uint32_t)P_SUB) < (uint32_t)(P_LEN - P_SUB);
    if (wave.any(live && sub)) {
        const uint32_t idx = sub ? (p - (uint32_t)P_SUB) + bfe(lo >> ROOT, 0, n) : 0u;
        const uint32_t e2 = L.lit[idx];
        const uint32_t n2 = e2 & 15u, p2 = e2 >> 4;
        const bool ok2 = !((p2 - (uint32_t)P_SUB) < (uint32_t)(P_LEN - P_SUB)) && n2 != 0u;      // a sub-table never points on
        n = sub ? (ok2 ? n2 : 0u) : n;
        nl = sub ? (uint32_t)ROOT + n2 : nl;
        p = sub ? p2 : p;
    }
    const bool is_lit = p < 256u, is_eob = p == (uint32_t)P_EOB, is_len = p >= (uint32_t)P_LEN;
    const uint32_t ext = (p >> 8) & 7u;
    const uint32_t len = (p & 255u) + 3u + bfe(lo, nl, ext);                 // nl + ext <= 20: inside the low word
    const uint32_t used1 = live ? nl + (is_len ? ext : 0u) : 0u;
    L.buf >>= used1; L.cnt -= (int)used1;
    refill_bf(L, live && is_len);
    const uint32_t lo2 = (uint32_t)L.buf;
    const uint32_t d = L.A[lo2 & (DROOT_SIZE - 1)];
    int dn = (int)(d >> 5) + 1, ds = (int)(d & 31u);
    if (wave.any(live && is_len && d == 0xFFu)) { int n2; const int s2 = dist_long(L, n2); if (d == 0xFFu) { ds = s2; dn = n2; } }
    uint32_t dbase, dext; dist_base_ext((uint32_t)(ds < 0 ? 0 : ds), dbase, dext);
    const uint32_t dist = dbase + bfe(lo2, (uint32_t)dn, dext);              // dn + dext <= 28
    const uint32_t used2 = (live && is_len) ? (uint32_t)dn + dext : 0u;
    L.buf >>= used2; L.cnt -= (int)used2;
    const uint32_t adv = is_lit ? 1u : len;
    const bool bad = n == 0u || (is_len && (ds < 0 || ds >= 30 || dist > L.outp)) || (!is_eob && (L.outp + adv > 65536u || L.ntok + 1u > TOKCAP));
    const bool emit = live && !bad && !is_eob;
    if (emit) L.stage[L.ntok & (STAGE - 1)] = is_lit ? p : tok_match(len, dist);
    L.ntok += emit ? 1u : 0u; L.outp += emit ? adv : 0u;
    if (live) L.state = bad ? ST_FALLBACK : is_eob ? ((L.last && !overrun(L)) ? ST_DONE : (L.last ? ST_FALLBACK : ST_HEADER)) : ST_DECODE;
}

// the whole member: the kernel's loop (and the CPU check's).  Sixteen uniform points (tokens out, input in), four symbols after each,
// between two looks at the lanes' states.
template <class W>
THJ_IHD void run_member(Lane& L, bool present, uint32_t skew, const W& wave) {
    // the ring's first 64 bytes before anything is read
    for (int k = 0; k < 5; ++k) topup(L);
    if (present) lane_start(L, skew);
    L.state = present ? ST_HEADER : ST_DONE;
    for (;;) {
        if (wave.any(L.state == ST_HEADER)) parse_header(L, wave);
        if (!wave.any(L.state == ST_DECODE)) break;
        for (int it = 0; it < 16; ++it) {
            flush_tokens(L);
            topup(L);
#pragma unroll
            for (int q = 0; q < 4; ++q) decode_one(L, L.state == ST_DECODE && input_ok(L), wave);
        }
    }
    flush_tokens_end(L);
}


// ================================================================================================ one member per WAVE (round 3, second design)
// The lane-per-member kernel above is bound by the latency of one member's decode chain: ~11 000 symbols at ~1300 cycles each are 7 ms
// however few members a launch holds, and the launches of the executables hold 1 500 - 10 000.  A Huffman stream can be entered
// anywhere: a decoder started at a wrong bit falls into step with the right one after a few dozen symbols (the codes are
// self-synchronising).  So the block's bits are cut into 64 equal segments, lane i decodes the symbols that START in segment i,
// and the lanes agree on the starts by iteration: lane i's start is where lane i - 1 stopped (the first symbol at or beyond the
// segment border); after a first pass from the borders themselves, passes repeat for the lanes whose start changed, until none
// does -- by induction from lane 0, whose start is exact, every start is then the true one.  Usually two passes, because a lane
// that started wrong has long fallen into step when it crosses its far border.  A last pass stores the tokens at the offsets a
// prefix sum over the lanes' counts gives.  ~4 passes over 1/64 of the symbols instead of one over all: ~0.4 ms a member.
//
// Positions are bit offsets in the aligned stream (Lane::src, 32-bit words).  The header is parsed by lane 0 with the Lane code above.
constexpr uint32_t MARK = 0xFFFFFF00u;                     // segment results >= MARK: no position
constexpr uint32_t MARK_EOB = 0xFFFFFF01u, MARK_ERR = 0xFFFFFF02u, MARK_NONE = 0xFFFFFF03u;
struct Seg { uint32_t e, nt, ob, eob_pos; };              // e: where the next lane starts (or a mark); tokens, bytes; the bit after an end-of-block code

struct PIn { uint64_t buf; int cnt; uint32_t nextw, widx; const uint32_t* w; };
THJ_IHD void pin_start(PIn& I, const uint32_t* w, uint32_t s) {
    const uint32_t k = s >> 5, b = s & 31u;
    I.w = w; I.buf = (uint64_t)(w[k] >> b); I.cnt = 32 - (int)b; I.nextw = w[k + 1]; I.widx = k + 2;
}
THJ_IHD uint32_t pin_pos(const PIn& I) { return (I.widx - 1u) * 32u - (uint32_t)I.cnt; }
THJ_IHD void pin_refill(PIn& I, bool want) { if (want && I.cnt <= 32) { I.buf |= (uint64_t)I.nextw << I.cnt; I.cnt += 32; I.nextw = I.w[I.widx++]; } }

// the symbols that start in [s, bnext), from tables lit / A / B.  limit = bits the member has.  MODE:
//   SEG_COUNT  nothing is stored (the warm-up pass);
//   SEG_SLOT   tokens go to tok[] while they fit (outp0 = the slot's capacity; the count runs on): the lane's own slot of the member's
//              scratch, from which compact_segment moves them to their place once the lanes agree and the counts before it are known;
//   SEG_FINAL  tokens go to tok[] = their final place, outp0 = the output position of the first, distances are checked against it
//              (a pass of its own after the agreement: only when a lane's tokens did not fit its slot).
constexpr int SEG_COUNT = 0, SEG_FINAL = 1, SEG_SLOT = 2;
constexpr uint32_t SLOT_TOKENS = 2u * TOKCAP / 64u;       // a lane's slot: twice an even share of the most a member can hold
template <int MODE, class W>
THJ_IHD Seg decode_segment(const uint16_t* lit, const uint8_t* A, const uint8_t* Bt, const uint32_t* w, uint32_t limit, uint32_t s, uint32_t bnext,
                           uint32_t* tok, uint32_t outp0, const W& wave) {
    Seg r{s, 0, 0, 0};
    if (s >= MARK) return r;                               // the lane before ended the block (or failed): nothing starts here
    if (s >= limit) { r.e = MARK_ERR; return r; }          // a start past the member's bits (a member of a few bytes cut into 64 segments): nothing to read there
    PIn I; pin_start(I, w, s);
    uint32_t pos = s, nt = 0, ob = 0;
    r.e = MARK_NONE;
    while (pos < bnext) {
        pin_refill(I, true);
        const uint32_t lo = (uint32_t)I.buf;
        const uint32_t e = lit[lo & (ROOT_SIZE - 1)];
        uint32_t n = e & 15u, p = e >> 4, nl = n;
        const bool sub = (p - (uint32_t)P_SUB) < (uint32_t)(P_LEN - P_SUB);
        if (wave.any(sub)) {
            const uint32_t idx = sub ? (p - (uint32_t)P_SUB) + bfe(lo >> ROOT, 0, n) : 0u;
            const uint32_t e2 = lit[idx];
            const uint32_t n2 = e2 & 15u, p2 = e2 >> 4;
            const bool ok2 = !((p2 - (uint32_t)P_SUB) < (uint32_t)(P_LEN - P_SUB)) && n2 != 0u;
            n = sub ? (ok2 ? n2 : 0u) : n; nl = sub ? (uint32_t)ROOT + n2 : nl; p = sub ? p2 : p;
        }
        const bool is_lit = p < 256u, is_eob = p == (uint32_t)P_EOB, is_len = p >= (uint32_t)P_LEN;
        const uint32_t ext = (p >> 8) & 7u;
        const uint32_t len = (p & 255u) + 3u + bfe(lo, nl, ext);
        const uint32_t used1 = nl + (is_len ? ext : 0u);
        I.buf >>= used1; I.cnt -= (int)used1;
        pin_refill(I, is_len);
        const uint32_t lo2 = (uint32_t)I.buf;
        const uint32_t d = A[lo2 & (DROOT_SIZE - 1)];
        int dn = (int)(d >> 5) + 1, ds = (int)(d & 31u);
        if (wave.any(is_len && d == 0xFFu)) { int n2; const int s2 = dist_long_raw(Bt, lo2, n2); if (d == 0xFFu) { ds = s2; dn = n2; } }
        uint32_t dbase, dext; dist_base_ext((uint32_t)(ds < 0 ? 0 : ds), dbase, dext);
        const uint32_t dist = dbase + bfe(lo2, (uint32_t)dn, dext);
        const uint32_t used2 = is_len ? (uint32_t)dn + dext : 0u;
        I.buf >>= used2; I.cnt -= (int)used2;
        pos = pin_pos(I);
        const uint32_t adv = is_lit ? 1u : len;
        bool bad = n == 0u || (is_len && (ds < 0 || ds >= 30)) || pos > limit;
        if (MODE == SEG_FINAL) bad = bad || (is_len && dist > outp0 + ob) || (!is_eob && outp0 + ob + adv > 65536u);
        if (bad) { r.e = MARK_ERR; break; }
        if (is_eob) { r.e = MARK_EOB; r.eob_pos = pos; break; }
        if (MODE == SEG_FINAL || (MODE == SEG_SLOT && nt < outp0)) tok[nt] = is_lit ? p : tok_match(len, dist);
        ++nt; ob += adv;
    }
    if (r.e == MARK_NONE) r.e = pos;                       // the first symbol at or beyond the border
    r.nt = nt; r.ob = ob;
    return r;
}

// a lane's n tokens from its slot to their place in the member's token list, with the checks the decode could not make before the
// output position of its first token (outp0) was known: no match reaches back beyond the member's first byte, the member ends within
// 64 KiB.  Returns false when one fails.
THJ_IHD bool compact_segment(const uint32_t* slot, uint32_t n, uint32_t* dst, uint32_t outp0) {
    uint32_t outp = outp0; bool ok = true;
    uint32_t t = 0;
    for (; t + 4u <= n; t += 4u) {                          // (slots are 16-byte aligned)
        struct alignas(16) Q { uint32_t x, y, z, w; };
        const Q q = *(const Q*)(slot + t);
        const uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool m = (v[k] >> 31) != 0u;
            ok = ok && !(m && (v[k] & 0x7FFFu) + 1u > outp);
            outp += m ? ((v[k] >> 15) & 255u) + 3u : 1u;
            dst[t + k] = v[k];
        }
    }
    for (; t < n; ++t) {
        const uint32_t v = slot[t];
        const bool m = (v >> 31) != 0u;
        ok = ok && !(m && (v & 0x7FFFu) + 1u > outp);
        outp += m ? ((v >> 15) & 255u) + 3u : 1u;
        dst[t] = v;
    }
    return ok && outp <= 65536u;
}

// lane 0's Lane at a bit position of the aligned stream (the first block: skew * 8; later blocks: the bit after the end-of-block code)
THJ_IHD void lane_seek(Lane& L, uint32_t bitpos) {
    const uint32_t byte = bitpos >> 3;
    L.ld = byte & ~15u; L.rd = L.ld + 4; L.inflight = false;
    for (int k = 0; k < 5; ++k) topup(L);
    L.rd = byte & ~3u;
    const uint32_t wd = L.ring[(L.rd & (RING - 1)) >> 2]; L.rd += 4;
    const uint32_t sh = 8u * (byte & 3u) + (bitpos & 7u);
    L.buf = (uint64_t)(wd >> sh); L.cnt = 32 - (int)sh;
    L.nextw = L.ring[(L.rd & (RING - 1)) >> 2]; L.rd += 4;
}
THJ_IHD uint32_t lane_bitpos(const Lane& L) { return (L.rd - 4u) * 8u - (uint32_t)L.cnt; }

// ---- the first half of a block header read from the member's bytes where thj_k_huffp has them anyway: in LDS, as 32-bit words `w`
// (parse_header_lengths reads through the Lane's ring, topped up from global memory 16 bytes at a time -- right for a lane that owns
// a member, a detour for lane 0 of a wave whose member is already staged: round trips to HBM in front of a serial decode of some
// three hundred code lengths).  Same rules, same outputs (A, and lit / B / C as scratch); bitpos / limit: bit offsets in `w`.
struct HeaderW { HeaderInfo hi; int last; uint32_t end_bit; bool fallback; };
THJ_IHD uint32_t pin_take(PIn& I, int n) { const uint32_t v = (uint32_t)I.buf & ((1u << n) - 1u); I.buf >>= n; I.cnt -= n; return v; }
THJ_IHD HeaderW parse_header_lengths_w(const uint32_t* w, uint32_t bitpos, uint32_t limit, uint16_t* lit, uint8_t* A, uint8_t* Bt, uint16_t* C) {
    HeaderW r;
    PIn I; pin_start(I, w, bitpos);
    pin_refill(I, true);
    r.last = (int)pin_take(I, 1);
    const int type = (int)pin_take(I, 2);
    bool dyn = type == 2; const bool fixed = type == 1;
    r.fallback = !dyn && !fixed;                                           // stored blocks (and BTYPE 3) go to the one-lane kernel
    int hlit = 288, hdist = 32, hclen = 0;
    bool ok = true;
    if (dyn) { pin_refill(I, true); hlit = (int)pin_take(I, 5) + 257; hdist = (int)pin_take(I, 5) + 1; hclen = (int)pin_take(I, 4) + 4; if (hlit > 286 || hdist > 30) { ok = false; dyn = false; } }
    uint8_t* cl = Bt;                                                      // 19 lengths, scratch
    for (int i = 0; i < 19; ++i) cl[i] = 0;
    {
        static const uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        for (int i = 0; i < 19; ++i) if (dyn && i < hclen) { pin_refill(I, true); cl[CLORD[i]] = (uint8_t)pin_take(I, 3); }
    }
    {
        for (int l = 0; l < 8; ++l) C[l] = 0;
        for (int i = 0; i < 19; ++i) C[cl[i]]++;
        int left = 1; for (int l = 1; l < 8; ++l) { left = (left << 1) - (int)C[l]; ok = ok && left >= 0; }
        uint32_t code = 0; for (int l = 1; l < 8; ++l) { C[16 + l] = (uint16_t)code; code = (code + C[l]) << 1; }
        for (int i = 0; i < 128; ++i) lit[i] = 0;
        for (int s = 0; s < 19; ++s) {
            const int l = cl[s];
            if (l) {
                const uint32_t c2 = C[16 + l]; C[16 + l] = (uint16_t)(c2 + 1);
                for (uint32_t f = rev_bits(c2, l); f < 128u; f += 1u << l) lit[f] = (uint16_t)((uint32_t)s << 4 | (uint32_t)l);
            }
        }
    }
    {
        const int n = hlit + hdist; int i = 0; int prev = 0;
        bool run = dyn && ok;
        while (run && i < n) {
            pin_refill(I, true);
            const uint32_t e = lit[(uint32_t)I.buf & 127u];
            const int l = (int)(e & 15u), sym = (int)(e >> 4);
            if (!l) { ok = false; run = false; }
            else {
                I.buf >>= l; I.cnt -= l;
                if (sym < 16) { A[i++] = (uint8_t)sym; prev = sym; }
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (i == 0) { ok = false; run = false; } val = prev; rep = 3 + (int)pin_take(I, 2); }
                    else if (sym == 17) { rep = 3 + (int)pin_take(I, 3); prev = 0; }
                    else { rep = 11 + (int)pin_take(I, 7); prev = 0; }
                    if (i + rep > n) { ok = false; run = false; rep = 0; }
                    for (int k = 0; k < rep; ++k) A[i + k] = (uint8_t)val;
                    i += rep;
                }
            }
            if (pin_pos(I) > limit) { ok = false; run = false; }             // ran off the member's bits: whatever follows is someone else's
        }
        if (dyn && ok && A[256] == 0) ok = false;
    }
    if (fixed) {
        for (int i = 0; i < 144; ++i) A[i] = 8;
        for (int i = 144; i < 256; ++i) A[i] = 9;
        for (int i = 256; i < 280; ++i) A[i] = 7;
        for (int i = 280; i < 288; ++i) A[i] = 8;
        for (int i = 0; i < 32; ++i) A[288 + i] = 5;
    }
    r.hi.hlit = hlit; r.hi.hdist = hdist; r.hi.build = (dyn && ok) || fixed; r.hi.ok = ok;
    r.end_bit = pin_pos(I);
    if (r.end_bit > limit) r.fallback = true;
    return r;
}

}  // namespace inf2
