// thj_ingest.hip -- device-side ingest: BGZF inflate and BAM record parsing on the GPU, so that
// the host of the drop-in executables only moves compressed bytes.
//
// Why: a tophat run hands segment_juncs / long_spanning_reads ~2.8 KB of (inflated) BAM per read pair; zlib inflates
// ~0.35 GB/s per core, so 2 M pairs/s would take every core of a 16-CPU container for inflating alone (measured: the CPU
// path stops at ~0.65 M pairs/s there).  BGZF (samtools-0.1.18 bgzf.c) is a chain of independent <= 64 KiB DEFLATE members and
// bam_write1 never lets a record straddle two of them (bgzf_flush_try, bam.c:225), so both steps are parallel over blocks.
//
// thj_k_inflate: one 64-lane workgroup per BGZF block.  DEFLATE is serial inside a block, so one lane decodes; what makes
// it fast enough is where its working set lives: Huffman tables (10-bit / 8-bit direct lookup + canonical fallback) and a
// 32 KiB output window in LDS -- every table lookup and every LZ77 copy is an LDS access -- while the other 63 lanes do the
// memory work: they stage the compressed bytes into an LDS ring ahead of the decoder and flush finished 16 KiB halves of
// the window to HBM with coalesced 16-byte stores.  The LDS window is 8 KiB (far matches read the flushed output): 14 KB of LDS
// per workgroup = 11 blocks in flight per CU, 2800 per GPU (with the full 32 KiB window: 4 and 1024, measured 2x slower).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>
#include <mutex>

#include "../../include/thj.h"
#include "thj_ctx.h"
#include "thj_scan.h"

extern "C" void* thj_pinned_alloc(size_t bytes);
extern "C" void thj_pinned_free(void* p);

namespace ing {

// Sizing (round 2, measured on 8 M pairs' maps, tools/inflate_ab.sh): the decoder is one lane of a wave, so the rate of the
// kernel is the number of members in flight, and that is set by LDS and registers per workgroup.  32 KiB window: 4 per CU,
// 6.8 GB/s of inflated bytes.  8 KiB: 8 per CU (the kernel then held 176 VGPRs -- the header parser's arrays -- so two waves per
// SIMD, not the 11 its LDS allowed), 12.5 GB/s.  With the header arrays in LDS and build_table out of line the decode loop
// needs 78 VGPRs (6 waves per SIMD), and: 4 KiB window + 2 KiB input ring 16 per CU 22 GB/s; 2 KiB + 2 KiB 20 per CU 21.8;
// 2 KiB + 1 KiB 23 per CU 23.4; 1 KiB + 1 KiB 24 per CU 24.3.  A match that reaches behind the LDS window reads the member's own
// output in HBM (flushed by then, see out_limit) -- half of all matches with a 2 KiB window on level-1 BAM, and still the better
// trade.
#ifndef THJ_INFLATE_INRING
#define THJ_INFLATE_INRING 1024
#endif
#ifndef THJ_INFLATE_WAVES
#define THJ_INFLATE_WAVES 6     /* waves per SIMD the register budget is set for */
#endif
#ifndef THJ_INFLATE_WIN
#define THJ_INFLATE_WIN 2048
#endif
static constexpr int WIN = THJ_INFLATE_WIN, WIN_MASK = WIN - 1, HALF = WIN / 2;
static constexpr int INRING = THJ_INFLATE_INRING, IN_MASK = INRING - 1;
static constexpr int LIT_BITS = 10, DIST_BITS = 8;
static constexpr int NEAR_MAX = WIN - 40;             // longest match distance served from the LDS piece of the window (copies overshoot by < 32 bytes)

struct Shared {
    uint8_t win[WIN];
    uint8_t in[INRING];
    uint16_t lit_fast[1 << LIT_BITS];
    uint16_t dist_fast[1 << DIST_BITS];
    uint16_t lit_sym[288], dist_sym[32];
    uint16_t lit_cnt[16], dist_cnt[16];
    uint8_t lens[320];
    uint8_t cl[20], dl[32];                           // code-length code lengths, distance code lengths (header parsing)
    uint16_t offs[16];                                // build_table's per-length offsets
    uint16_t lbase[32], dbase[32];                    // length / distance bases and extra-bit counts (RFC 1951, 3.2.5): in LDS, the
    uint8_t lext[32], dext[32];                       // decoder looks one up per match
    // decoder <-> helpers
    uint32_t in_staged;        // compressed bytes staged so far (absolute)
    uint32_t in_used;          // compressed bytes the decoder no longer needs (absolute, rounded down to 4)
    uint32_t out_total;        // bytes produced
    uint32_t out_flushed;      // bytes already in HBM
    uint32_t done, error;
};

__constant__ uint16_t LBASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
__constant__ uint8_t LEXT[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
__constant__ uint16_t DBASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
__constant__ uint8_t DEXT[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
__constant__ uint8_t CLORD[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

__device__ __forceinline__ uint32_t rev_bits(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// canonical Huffman tables from code lengths: direct lookup for codes <= FAST bits (entry = symbol | length << 12, 0 = longer
// code), count / symbol arrays for the rest (decoded bit by bit like puff.c)
__device__ __noinline__ bool build_table(const uint8_t* lens, int n, uint16_t* fast, int fast_bits, uint16_t* sym, uint16_t* cnt, uint16_t* offs) {
    for (int i = 0; i < 16; ++i) cnt[i] = 0;
    for (int i = 0; i < n; ++i) cnt[lens[i]]++;
    for (int i = 0; i < (1 << fast_bits); ++i) fast[i] = 0;
    if (cnt[0] == n) return true;                    // no codes at all (e.g. a block without distances)
    int left = 1;
    for (int l = 1; l < 16; ++l) { left <<= 1; left -= cnt[l]; if (left < 0) return false; }
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + cnt[l];
    for (int i = 0; i < n; ++i) if (lens[i]) sym[offs[lens[i]]++] = (uint16_t)i;
    // canonical codes in symbol order per length
    int code = 0, idx = 0;
    for (int l = 1; l <= fast_bits; ++l) {
        for (int k = 0; k < cnt[l]; ++k, ++idx, ++code) {
            const uint32_t r = rev_bits((uint32_t)code, l);
            const uint16_t e = (uint16_t)(sym[idx] | (l << 12));
            for (uint32_t f = r; f < (1u << fast_bits); f += (1u << l)) fast[f] = e;
        }
        code <<= 1;
    }
    return true;
}

struct Bits {
    uint64_t buf; int cnt; uint32_t pos;            // pos = absolute compressed byte index of the next byte to load
};

// the ring is filled in whole 32-bit words (zero-padded past the end of the stream) and position 0 of the stream sits at ring
// offset 0, so the bit buffer takes one aligned LDS word at a time
__device__ __forceinline__ void refill(Bits& b, const Shared& s, uint32_t staged) {
    if (b.cnt <= 32 && b.pos < staged) { b.buf |= (uint64_t)((const uint32_t*)s.in)[(b.pos & IN_MASK) >> 2] << b.cnt; b.cnt += 32; b.pos += 4; }
}
__device__ __forceinline__ uint32_t take(Bits& b, int n) { const uint32_t v = (uint32_t)(b.buf & ((1ull << n) - 1)); b.buf >>= n; b.cnt -= n; return v; }

__device__ __forceinline__ int decode_sym(Bits& b, const uint16_t* fast, int fast_bits, const uint16_t* sym, const uint16_t* cnt) {
    const uint16_t e = fast[b.buf & ((1u << fast_bits) - 1)];
    if (e) { const int l = e >> 12; b.buf >>= l; b.cnt -= l; return e & 0xFFF; }
    // longer code: canonical decode, one bit at a time, starting past the lengths the direct table covers
    int code = 0, first = 0, index = 0;
    for (int l = 1; l < 16; ++l) {
        code |= (int)((b.buf >> (l - 1)) & 1);
        const int c = cnt[l];
        if (l > fast_bits && code - c < first) { b.buf >>= l; b.cnt -= l; return sym[index + (code - first)]; }
        index += c; first += c; first <<= 1; code <<= 1;
    }
    return -1;
}

enum { ST_HEADER = 0, ST_STORED, ST_CODES, ST_DONE };

}  // namespace ing


// out: 65536 bytes per block (block b at b << 16); out_len[b] = inflated bytes (0xFFFFFFFF on a corrupt stream)
// list / list_n (optional): the members to do are list[0 .. *list_n) -- the ones the two-kernel inflater below handed back
__global__ __launch_bounds__(64, THJ_INFLATE_WAVES) void thj_k_inflate(const uint8_t* __restrict__ comp, const thj_bgzf_block* __restrict__ blocks, int n_blocks,
                                                     uint8_t* __restrict__ out, uint32_t* __restrict__ out_len,
                                                     const uint32_t* __restrict__ list, const uint32_t* __restrict__ list_n) {
    using namespace ing;
    __shared__ Shared s;
    const int lane = threadIdx.x;
    if (lane < 29) { s.lbase[lane] = LBASE[lane]; s.lext[lane] = LEXT[lane]; }
    if (lane < 30) { s.dbase[lane] = DBASE[lane]; s.dext[lane] = DEXT[lane]; }
    __syncthreads();
    const int n_todo = list ? (int)*list_n : n_blocks;
    for (int idx = blockIdx.x; idx < n_todo; idx += gridDim.x) {
        const int blk = list ? (int)list[idx] : idx;
        const uint8_t* in = comp + blocks[blk].in_off;
        const uint32_t in_len = blocks[blk].in_len;
        uint8_t* dst = out + ((size_t)blk << 16);
        if (lane == 0) { s.in_staged = 0; s.in_used = 0; s.out_total = 0; s.out_flushed = 0; s.done = 0; s.error = 0; }
        __syncthreads();
        // decoder state (lane 0)
        Bits b{0, 0, 0};
        int state = ST_HEADER, last = 0;
        uint32_t stored_left = 0;
        for (;;) {
            // ---- helpers: stage input ahead of the decoder (the ring holds INRING bytes; keep what is not used yet)
            {
                uint32_t staged = s.in_staged;
                const uint32_t used = s.in_used;
                while (staged < in_len && staged + 256 <= used + INRING) {
                    const uint32_t p = staged + (uint32_t)lane * 4;
                    if (p < in_len) {
                        // one ring word per lane; the source address is only byte-aligned, the tail is zero-padded
                        uint32_t w = 0;
                        const uint32_t nb = in_len - p < 4 ? in_len - p : 4;
                        for (uint32_t k = 0; k < nb; ++k) w |= (uint32_t)in[p + k] << (8 * k);
                        ((uint32_t*)s.in)[(p & IN_MASK) >> 2] = w;
                    }
                    staged += 256;
                }
                if (staged > in_len) staged = in_len;
                // ---- helpers: flush finished halves of the window
                const uint32_t total = s.out_total;
                uint32_t flushed = s.out_flushed;
                const bool fin = s.done || s.error;
                while (flushed + HALF <= total || (fin && flushed < total)) {
                    const uint32_t n = total - flushed < (uint32_t)HALF ? total - flushed : (uint32_t)HALF;
                    for (uint32_t o = (uint32_t)lane * 16; o < n; o += 64 * 16) {
                        if (o + 16 <= n) {
                            const uint4 v = *(const uint4*)&s.win[(flushed + o) & WIN_MASK];
                            *(uint4*)(dst + flushed + o) = v;
                        } else for (uint32_t k = o; k < n; ++k) dst[flushed + k] = s.win[(flushed + k) & WIN_MASK];
                    }
                    flushed += n;
                }
                __syncthreads();
                if (lane == 0) { s.in_staged = staged; s.out_flushed = flushed; }
                __syncthreads();
                if (fin) break;
            }
            // ---- decoder
            if (lane == 0) {
                const uint32_t staged0 = s.in_staged;                            // fixed while the decoder runs
                const bool all_in = staged0 >= in_len;
                uint32_t outp = s.out_total;
                const uint32_t out_limit = s.out_flushed + WIN - 300;           // never overwrite window bytes not yet in HBM
                bool err = false, fin = false;
                // work while enough input is staged for the largest thing the state needs
                for (;;) {
                    refill(b, s, staged0);
                    if (b.cnt < 0) { err = true; break; }                        // ran past the end of a truncated stream
                    const int avail_i = (int)staged0 - (int)b.pos + (b.cnt >> 3);
                    const uint32_t avail = avail_i > 0 ? (uint32_t)avail_i : 0u;
                    if (state == ST_HEADER) {
                        if (!all_in && avail < 400) break;
                        last = (int)take(b, 1);
                        const int type = (int)take(b, 2);
                        if (type == 0) {
                            take(b, b.cnt & 7);                                  // to a byte boundary
                            refill(b, s, staged0);
                            const uint32_t len = take(b, 16), nlen = take(b, 16);
                            if ((len ^ 0xFFFFu) != nlen) { err = true; break; }
                            stored_left = len; state = ST_STORED;
                        } else if (type == 1) {
                            for (int i = 0; i < 144; ++i) s.lens[i] = 8;
                            for (int i = 144; i < 256; ++i) s.lens[i] = 9;
                            for (int i = 256; i < 280; ++i) s.lens[i] = 7;
                            for (int i = 280; i < 288; ++i) s.lens[i] = 8;
                            build_table(s.lens, 288, s.lit_fast, LIT_BITS, s.lit_sym, s.lit_cnt, s.offs);
                            for (int i = 0; i < 30; ++i) s.lens[i] = 5;
                            build_table(s.lens, 30, s.dist_fast, DIST_BITS, s.dist_sym, s.dist_cnt, s.offs);
                            state = ST_CODES;
                        } else if (type == 2) {
                            const int hlit = (int)take(b, 5) + 257, hdist = (int)take(b, 5) + 1, hclen = (int)take(b, 4) + 4;
                            if (hlit > 286 || hdist > 30) { err = true; break; }
                            uint8_t* cl = s.cl;
                            for (int i = 0; i < 19; ++i) cl[i] = 0;
                            for (int i = 0; i < hclen; ++i) { refill(b, s, staged0); cl[CLORD[i]] = (uint8_t)take(b, 3); }
                            // the code-length code uses the distance table's storage (7-bit direct lookup fits its 8 bits)
                            if (!build_table(cl, 19, s.dist_fast, 7, s.dist_sym, s.dist_cnt, s.offs)) { err = true; break; }
                            int i = 0;
                            while (i < hlit + hdist) {
                                refill(b, s, staged0);
                                const int sym = decode_sym(b, s.dist_fast, 7, s.dist_sym, s.dist_cnt);
                                if (sym < 0) { err = true; break; }
                                if (sym < 16) s.lens[i++] = (uint8_t)sym;
                                else {
                                    int rep, val = 0;
                                    if (sym == 16) { if (i == 0) { err = true; break; } val = s.lens[i - 1]; rep = 3 + (int)take(b, 2); }
                                    else if (sym == 17) rep = 3 + (int)take(b, 3);
                                    else rep = 11 + (int)take(b, 7);
                                    if (i + rep > hlit + hdist) { err = true; break; }
                                    while (rep--) s.lens[i++] = (uint8_t)val;
                                }
                            }
                            if (err) break;
                            if (s.lens[256] == 0) { err = true; break; }
                            // distance lengths first (they sit behind the literal lengths and the literal table build does not touch them)
                            uint8_t* dl = s.dl;
                            for (int k = 0; k < hdist; ++k) dl[k] = s.lens[hlit + k];
                            if (!build_table(s.lens, hlit, s.lit_fast, LIT_BITS, s.lit_sym, s.lit_cnt, s.offs)) { err = true; break; }
                            if (!build_table(dl, hdist, s.dist_fast, DIST_BITS, s.dist_sym, s.dist_cnt, s.offs)) { err = true; break; }
                            state = ST_CODES;
                        } else { err = true; break; }
                    } else if (state == ST_STORED) {
                        if (stored_left == 0) { state = last ? ST_DONE : ST_HEADER; continue; }
                        if (avail == 0) { if (all_in) err = true; break; }
                        if (outp >= out_limit) break;
                        s.win[outp & WIN_MASK] = (uint8_t)take(b, 8); ++outp; --stored_left;
                    } else if (state == ST_CODES) {
                        if ((!all_in && avail < 8) || outp >= out_limit) break;
                        int sym = decode_sym(b, s.lit_fast, LIT_BITS, s.lit_sym, s.lit_cnt);
                        if (sym < 0) { err = true; break; }
                        if (sym < 256) {
                            s.win[outp & WIN_MASK] = (uint8_t)sym; ++outp;
                            // runs of literals: stay in a loop of lookup + store (the direct table only; anything else goes round again)
                            for (;;) {
                                refill(b, s, staged0);
                                if (b.cnt < 24 || outp >= out_limit) break;
                                const uint16_t e = s.lit_fast[b.buf & ((1u << LIT_BITS) - 1)];
                                if (!e || (e & 0xFFF) >= 256) break;
                                const int l = e >> 12; b.buf >>= l; b.cnt -= l;
                                s.win[outp & WIN_MASK] = (uint8_t)(e & 0xFF); ++outp;
                            }
                        }
                        else if (sym == 256) state = last ? ST_DONE : ST_HEADER;
                        else {
                            sym -= 257;
                            if (sym >= 29) { err = true; break; }
                            const int len = s.lbase[sym] + (int)take(b, s.lext[sym]);
                            refill(b, s, staged0);
                            const int ds = decode_sym(b, s.dist_fast, DIST_BITS, s.dist_sym, s.dist_cnt);
                            if (ds < 0 || ds >= 30) { err = true; break; }
                            const uint32_t dist = s.dbase[ds] + take(b, s.dext[ds]);
                            if (dist > outp || outp + (uint32_t)len > 65536u) { err = true; break; }
                            int k = 0;
                            if (dist > (uint32_t)NEAR_MAX) {                      // beyond the LDS window: from the flushed output (see out_limit)
                                // Half of all matches come this way with a 2 KiB window, and a global load is a ~1 us round trip for the one
                                // decoding lane: 32 bytes are fetched per trip (four unaligned 64-bit loads in flight together), whatever the
                                // match needs of them.  (The first version read eight bytes per trip and the last 1..7 bytes one trip each.)
                                // Like the LDS copies below, a group may write past the match: up to 31 ring positions ahead of the output
                                // pointer, which is why matches this close to the window's far edge (NEAR_MAX) are served from here.
                                const uint8_t* gsrc = dst + outp - dist;
                                for (; k < len; k += 32) {
                                    uint64_t t[4];
#pragma unroll
                                    for (int q = 0; q < 4; ++q) __builtin_memcpy(&t[q], gsrc + k + 8 * q, 8);
#pragma unroll
                                    for (int q = 0; q < 4; ++q) {
                                        if (k + 8 * q >= len) break;
                                        const uint32_t dp = (outp + (uint32_t)(k + 8 * q)) & WIN_MASK;
                                        if (dp <= (uint32_t)WIN - 8u) __builtin_memcpy(&s.win[dp], &t[q], 8);
                                        else for (int b8 = 0; b8 < 8; ++b8) s.win[(dp + (uint32_t)b8) & WIN_MASK] = (uint8_t)(t[q] >> (8 * b8));
                                    }
                                }
                            } else {
                                // Copies go eight bytes at a time through unaligned 64-bit LDS accesses (one ds_read_b64 + one ds_write_b64
                                // per group instead of sixteen byte accesses with their address arithmetic: the decoder lane is bound by the
                                // instructions it issues).  A group may run up to seven bytes past the match: those ring positions are
                                // ahead of the output pointer, rewritten by whatever comes next, and never flushed (out_limit keeps 300
                                // bytes of slack).  A distance below 8 is a period: the first m * dist >= 8 bytes go byte by byte, the rest is
                                // the same copy at distance m * dist.
                                uint32_t d8 = dist;
                                if (dist < 8u) {
                                    const uint32_t m = (8u + dist - 1u) / dist;
                                    d8 = m * dist;
                                    const int head = len < (int)d8 ? len : (int)d8;
                                    for (; k < head; ++k) s.win[(outp + (uint32_t)k) & WIN_MASK] = s.win[(outp + (uint32_t)k - dist) & WIN_MASK];
                                }
                                for (; k < len; k += 8) {
                                    const uint32_t sp = (outp + (uint32_t)k - d8) & WIN_MASK, dp = (outp + (uint32_t)k) & WIN_MASK;
                                    if (sp <= (uint32_t)WIN - 8u && dp <= (uint32_t)WIN - 8u) {
                                        uint64_t t;
                                        __builtin_memcpy(&t, &s.win[sp], 8);
                                        __builtin_memcpy(&s.win[dp], &t, 8);
                                    } else {                                       // the group wraps round the ring
                                        uint8_t t[8];
#pragma unroll
                                        for (int q = 0; q < 8; ++q) t[q] = s.win[(sp + (uint32_t)q) & WIN_MASK];
#pragma unroll
                                        for (int q = 0; q < 8; ++q) s.win[(dp + (uint32_t)q) & WIN_MASK] = t[q];
                                    }
                                }
                            }
                            outp += (uint32_t)len;
                        }
                    } else { fin = true; break; }
                    if (outp > 65536u) { err = true; break; }
                }
                s.out_total = outp;
                s.in_used = (uint32_t)((int)b.pos - (b.cnt >> 3) > 0 ? (int)b.pos - (b.cnt >> 3) : 0) & ~3u;
                if (fin) s.done = 1;
                if (err) s.error = 1;
                // no progress possible and not finished: input exhausted in mid-stream
                if (!fin && !err && all_in && (int)staged0 - (int)b.pos + (b.cnt >> 3) <= 0 && state != ST_DONE) s.error = 1;
                if (state == ST_DONE) s.done = 1;
            }
            __syncthreads();
        }
        if (lane == 0) out_len[blk] = s.error ? 0xFFFFFFFFu : s.out_total;
        __syncthreads();
    }
}

// thj_k_inflate_lanes: one LANE per BGZF block -- 64 independent inflaters per wave.  A block decodes at a few MB/s whoever does it
// (every symbol is a chain of dependent table lookups), so throughput is the number of blocks in flight: 64 per wave instead of
// one.  Per lane: a 256-entry literal / 32-entry distance direct-lookup table plus the canonical symbol lists in LDS (1220 bytes,
// an odd word stride so that the lanes' tables start in different banks), the limits of the longer codes in registers (a
// compare chain, no memory), the compressed stream read in aligned words, and the lane's own 64 KiB output slot as the LZ77
// window -- a lane always sees its own stores.  Output bytes gather in a register and leave four at a time.
namespace ing {
static constexpr int LF = 8, DF = 5;                                   // direct-lookup bits
static constexpr int LANE_WORDS = 305;                                 // 128 + 16 + 144 + 16 words of tables, +1: odd (78 KB per wave: two per CU)
struct LaneTab {
    uint16_t* lit_fast; uint16_t* dist_fast; uint16_t* lit_sym; uint16_t* dist_sym;
};
struct Canon { uint32_t limit[16]; int32_t base[16]; };                // per code length: left-justified exclusive code limit, symbol index base

// the per-length counters / scatter offsets are a dynamically indexed private array (scratch): touched once per block header
__device__ __noinline__ bool lane_build(const uint8_t* lens, int n, uint16_t* fast, int fast_bits, uint16_t* sym, Canon& cn) {
    uint32_t tmp[16];
    for (int i = 0; i < 16; ++i) tmp[i] = 0;
    for (int i = 0; i < n; ++i) tmp[lens[i]]++;
    for (int i = 0; i < (1 << fast_bits); ++i) fast[i] = 0;
    uint32_t cnt[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) cnt[l] = tmp[l];
#pragma unroll
    for (int l = 0; l < 16; ++l) { cn.limit[l] = 0; cn.base[l] = 0; }
    if ((int)cnt[0] == n) return true;
    int left = 1;
#pragma unroll
    for (int l = 1; l < 16; ++l) { left <<= 1; left -= (int)cnt[l]; }
    {   // over-subscribed at some length?
        int lf = 1; bool bad = false;
#pragma unroll
        for (int l = 1; l < 16; ++l) { lf <<= 1; lf -= (int)cnt[l]; bad = bad || lf < 0; }
        if (bad) return false;
    }
    // offsets of each length's symbols in sym[] (kept in tmp for the scatter), first canonical code per length
    uint32_t off = 0, code = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        tmp[l] = off;
        cn.base[l] = (int32_t)off - (int32_t)code;
        cn.limit[l] = (code + cnt[l]) << (15 - l);
        off += cnt[l];
        code = (code + cnt[l]) << 1;
    }
    for (int i = 0; i < n; ++i) { const int l = lens[i]; if (l) sym[tmp[l]++] = (uint16_t)i; }
    // direct table for the codes of up to fast_bits bits
    uint32_t c2 = 0, idx = 0;
#pragma unroll
    for (int l = 1; l < 16; ++l) {
        if (l <= fast_bits) {
            for (uint32_t k = 0; k < cnt[l]; ++k, ++idx, ++c2) {
                const uint32_t r = rev_bits(c2, l);
                const uint16_t e = (uint16_t)(sym[idx] | (l << 12));
                for (uint32_t f = r; f < (1u << fast_bits); f += (1u << l)) fast[f] = e;
            }
            c2 <<= 1;
        }
    }
    return true;
}
__device__ __forceinline__ uint32_t lane_take(uint64_t& buf, int& cnt, int n) {
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n; cnt -= n;
    return v;
}
template <int FB>
__device__ __forceinline__ int lane_decode(uint64_t& buf, int& cnt, const uint16_t* fast, const uint16_t* sym, const Canon& cn) {
    const uint16_t e = fast[buf & ((1u << FB) - 1)];
    if (e) { const int l = e >> 12; buf >>= l; cnt -= l; return e & 0xFFF; }
    const uint32_t c15 = __brev((uint32_t)buf) >> 17;                   // the next 15 bits, first bit most significant
    int res = -1;
#pragma unroll
    for (int l = 15; l > FB; --l)                                       // the shortest length whose limit the code is under
        if (c15 < cn.limit[l]) res = l;
    if (res < 0) return -1;
    int l = res; int32_t b = 0;
#pragma unroll
    for (int q = FB + 1; q < 16; ++q) b = (q == l) ? cn.base[q] : b;
    const int s_ = sym[b + (int32_t)(c15 >> (15 - l))];
    buf >>= l; cnt -= l;
    return s_;
}
}  // namespace ing

__global__ __launch_bounds__(64) void thj_k_inflate_lanes(const uint8_t* __restrict__ comp, const thj_bgzf_block* __restrict__ blocks, int n_blocks,
                                                           uint8_t* __restrict__ out, uint32_t* __restrict__ out_len) {
    using namespace ing;
    __shared__ uint32_t lds[64 * LANE_WORDS];
    const int lane = threadIdx.x;
    const int blk = blockIdx.x * 64 + lane;
    if (blk >= n_blocks) return;
    uint32_t* my = lds + lane * LANE_WORDS;
    uint16_t* lit_fast = (uint16_t*)my;                 // 256 entries = 128 words
    uint16_t* dist_fast = (uint16_t*)(my + 128);        // 32 entries = 16 words
    uint16_t* lit_sym = (uint16_t*)(my + 144);          // 288 entries = 144 words
    uint16_t* dist_sym = (uint16_t*)(my + 288);         // 32 entries = 16 words  (+1 pad word)
    const uint8_t* in = comp + blocks[blk].in_off;
    const uint32_t in_len = blocks[blk].in_len;
    uint8_t* dst = out + ((size_t)blk << 16);
    // compressed stream in aligned words
    const uint32_t mis = (uint32_t)((uintptr_t)in & 3u);
    const uint32_t* wp = (const uint32_t*)(in - mis);
    const uint32_t n_words = (in_len + mis + 3) >> 2;
    uint32_t wi = 0;
    uint64_t buf = 0; int cnt = 0;
    if (n_words) { buf = (uint64_t)(wp[0] >> (8 * mis)); cnt = 32 - 8 * (int)mis; wi = 1; }
#define LREFILL() do { if (cnt <= 32) { const uint32_t w_ = wi < n_words ? wp[wi] : 0u; buf |= (uint64_t)w_ << cnt; cnt += 32; ++wi; } } while (0)
#define LTAKE(n_) lane_take(buf, cnt, (n_))
    uint32_t outp = 0, acc = 0;
#define LPUT(byte_) do { acc |= (uint32_t)(uint8_t)(byte_) << (8 * (outp & 3u)); ++outp; if ((outp & 3u) == 0) { *(uint32_t*)(dst + outp - 4) = acc; acc = 0; } } while (0)
    // the long-code limits stay in registers: the build fills a scratch copy (its address goes to a call), this copies it over
#define LCOPY(dst_, src_) do { _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) { dst_.limit[q_] = src_.limit[q_]; dst_.base[q_] = src_.base[q_]; } } while (0)
    Canon lc, dc;
#pragma unroll
    for (int q = 0; q < 16; ++q) { lc.limit[q] = dc.limit[q] = 0; lc.base[q] = dc.base[q] = 0; }
    uint8_t lens[320];
    bool err = false;
    int last = 0;
    while (!last && !err) {
        LREFILL();
        last = (int)LTAKE(1);
        const int type = (int)LTAKE(2);
        if (type == 0) {
            LTAKE(cnt & 7);
            LREFILL();
            const uint32_t len = LTAKE(16);
            LREFILL();
            const uint32_t nlen = LTAKE(16);
            if ((len ^ 0xFFFFu) != nlen || outp + len > 65536u) { err = true; break; }
            for (uint32_t k = 0; k < len; ++k) { LREFILL(); const uint32_t v = LTAKE(8); LPUT(v); }
            continue;
        }
        if (type == 3) { err = true; break; }
        if (type == 1) {
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            Canon t;
            lane_build(lens, 288, lit_fast, LF, lit_sym, t); LCOPY(lc, t);
            for (int i = 0; i < 30; ++i) lens[i] = 5;
            lane_build(lens, 30, dist_fast, DF, dist_sym, t); LCOPY(dc, t);
        } else {
            const int hlit = (int)LTAKE(5) + 257, hdist = (int)LTAKE(5) + 1, hclen = (int)LTAKE(4) + 4;
            if (hlit > 286 || hdist > 30) { err = true; break; }
            uint8_t cl[19];
            for (int i = 0; i < 19; ++i) cl[i] = 0;
            for (int i = 0; i < hclen; ++i) { LREFILL(); cl[CLORD[i]] = (uint8_t)LTAKE(3); }
            // the code-length code borrows the literal table's storage (19 symbols, codes of at most 7 bits: all direct)
            Canon cc;
            if (!lane_build(cl, 19, lit_fast, 7, dist_sym, cc)) { err = true; break; }
            int i = 0;
            while (i < hlit + hdist) {
                LREFILL();
                const uint16_t e = lit_fast[buf & 127u];
                if (!e) { err = true; break; }
                { const int l = e >> 12; buf >>= l; cnt -= l; }
                const int sym = e & 0xFFF;
                if (sym < 16) lens[i++] = (uint8_t)sym;
                else {
                    int rep, val = 0;
                    if (sym == 16) { if (i == 0) { err = true; break; } val = lens[i - 1]; rep = 3 + (int)LTAKE(2); }
                    else if (sym == 17) rep = 3 + (int)LTAKE(3);
                    else rep = 11 + (int)LTAKE(7);
                    if (i + rep > hlit + hdist) { err = true; break; }
                    while (rep--) lens[i++] = (uint8_t)val;
                }
            }
            if (err) break;
            if (lens[256] == 0) { err = true; break; }
            Canon t;
            if (!lane_build(lens + hlit, hdist, dist_fast, DF, dist_sym, t)) { err = true; break; }
            LCOPY(dc, t);
            if (!lane_build(lens, hlit, lit_fast, LF, lit_sym, t)) { err = true; break; }
            LCOPY(lc, t);
        }
        // ---- the block's symbols
        for (;;) {
            LREFILL();
            int sym = lane_decode<LF>(buf, cnt, lit_fast, lit_sym, lc);
            if (sym < 0) { err = true; break; }
            if (sym < 256) { if (outp >= 65536u) { err = true; break; } LPUT(sym); continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) { err = true; break; }
            const int len = (int)LBASE[sym] + (int)LTAKE(LEXT[sym]);
            LREFILL();
            const int ds = lane_decode<DF>(buf, cnt, dist_fast, dist_sym, dc);
            if (ds < 0 || ds >= 30) { err = true; break; }
            LREFILL();
            const uint32_t dist = (uint32_t)DBASE[ds] + LTAKE(DEXT[ds]);
            if (dist > outp || outp + (uint32_t)len > 65536u) { err = true; break; }
            int k = 0;
            if (dist >= 12) {
                // four bytes at a time: the source words were stored at least two words ago (the lane reads its own stores back)
                for (; k + 4 <= len; k += 4) {
                    const uint32_t sp = outp - dist, sh = 8 * (sp & 3u);
                    const uint32_t* swp = (const uint32_t*)(dst + (sp & ~3u));
                    uint32_t v = swp[0];
                    if (sh) v = (v >> sh) | (swp[1] << (32 - sh));
                    const uint32_t os = 8 * (outp & 3u);
                    if (os == 0) *(uint32_t*)(dst + outp) = v;
                    else { acc |= v << os; *(uint32_t*)(dst + (outp & ~3u)) = acc; acc = v >> (32 - os); }
                    outp += 4;
                }
            }
            for (; k < len; ++k) {                                    // bytes below outp & ~3 are in memory, the rest in acc
                const uint32_t sp = outp - dist;
                const uint32_t v = sp >= (outp & ~3u) ? (acc >> (8 * (sp & 3u))) & 0xFFu : (uint32_t)dst[sp];
                LPUT(v);
            }
        }
        if (cnt < 0) err = true;
    }
    if (!err && (outp & 3u)) { for (uint32_t k = 0; k < (outp & 3u); ++k) dst[(outp & ~3u) + k] = (uint8_t)(acc >> (8 * k)); }
    out_len[blk] = err ? 0xFFFFFFFFu : outp;
#undef LREFILL
#undef LTAKE
#undef LPUT
#undef LCOPY
}


// ================================================================================================ the two-kernel inflater (round 3)
// thj_inflate_core.h has the design: entropy decoding one member per lane (thj_k_huff), LZ77 resolution one wave per member (thj_k_lz).
#include "thj_inflate_core.h"

namespace inf2 {
struct WaveGpu { __device__ __forceinline__ bool any(bool p) const { return __any((int)p) != 0; } };
static constexpr int LZ_HIST = 32768, LZ_SLACK = 8192 - 64, LZ_CAP = LZ_HIST + LZ_SLACK;     // + 64 bytes of read slack = 40 KiB: four waves per CU

// wave-wide inclusive prefix sum in DPP steps (row_shr 1, 2, 4, 8 inside the rows of 16, then row_bcast 15 and 31 across them)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
    return v;
}
}  // namespace inf2

// LPW = lanes used per wave: a workgroup always holds 64 members (the LDS of one CU), as 64 / LPW waves
template <int LPW>
__global__ __launch_bounds__(64 * (64 / LPW)) void thj_k_huff(const uint8_t* __restrict__ comp, const thj_bgzf_block* __restrict__ blocks, int n_blocks,
                                                              uint32_t* __restrict__ tokens, uint32_t* __restrict__ ntok, uint32_t* __restrict__ out_len) {
    using namespace inf2;
    __shared__ uint32_t lds[64 * STRIDE_WORDS];
    const int lane = threadIdx.x & 63, slot = (int)(threadIdx.x >> 6) * LPW + lane;
    const int m = (int)blockIdx.x * 64 + slot;
    if (lane >= LPW || m >= n_blocks) return;                              // the wave-wide votes below only count the lanes that stay
    uint8_t* base = (uint8_t*)(lds + slot * STRIDE_WORDS);
    Lane L;
    L.lit = (uint16_t*)base; L.A = base + OFF_A; L.B = base + OFF_B; L.C = (uint16_t*)(base + OFF_C); L.ring = (uint32_t*)(base + OFF_RING); L.stage = (uint32_t*)(base + OFF_STAGE);
    const uint8_t* in = comp + blocks[m].in_off;
    const uint32_t skew = (uint32_t)((uintptr_t)in & 15u);
    L.src = in - skew; L.total = skew + blocks[m].in_len;
    L.buf = 0; L.cnt = 0; L.nextw = 0; L.rd = 4; L.ld = 0; L.outp = 0; L.ntok = 0; L.nflushed = 0; L.state = ST_HEADER; L.last = 0; L.inflight = false;
    L.pend[0] = L.pend[1] = L.pend[2] = L.pend[3] = 0;
    L.tok = tokens + (size_t)m * TOKCAP;
    run_member(L, true, skew, WaveGpu{});
    const bool good = L.state == ST_DONE && !overrun(L);
    ntok[m] = good ? L.ntok : NTOK_FALLBACK;
    out_len[m] = good ? L.outp : 0xFFFFFFFFu;
}

// One wave per member (the design note is in thj_inflate_core.h): lane 0 parses the block header and builds the tables, then the 64
// lanes decode 64 segments of the block's bits -- a warm-up pass from one segment before each border, passes until the lanes agree on
// where each segment's first symbol starts, and a last pass that stores the tokens.
namespace inf2 {
struct WaveOne { __device__ __forceinline__ bool any(bool p) const { return p; } };
// the wave the table builders are written against (build_lit_wave / build_dist_wave)
struct WaveTab {
    int lane;
    __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
    __device__ __forceinline__ void sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
};
}
#ifdef THJ_EXP
// developer build: where a wave of thj_k_huffp spends its clocks (lane 0's clock64 deltas, summed over all members)
__device__ unsigned long long thj_huffp_dbg[16];
extern "C" int thj_huffp_dbg_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(thj_huffp_dbg), sizeof thj_huffp_dbg) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(thj_huffp_dbg), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#define HP_T(i) do { const unsigned long long t__ = clock64(); hdbg[i] += t__ - htl; htl = t__; } while (0)
#define HP_N(i, v) do { hdbg[i] += (v); } while (0)
#else
#define HP_T(i) do { } while (0)
#define HP_N(i, v) do { } while (0)
#endif
__global__ __launch_bounds__(64) void thj_k_huffp(const uint8_t* __restrict__ comp, const thj_bgzf_block* __restrict__ blocks, int n_blocks,
                                                  uint32_t* __restrict__ tokens, uint32_t* __restrict__ slots, uint32_t* __restrict__ ntok, uint32_t* __restrict__ out_len, uint32_t comp_cap) {
    using namespace inf2;
    // [tables and lane 0's input ring: STRIDE_WORDS words][the member's compressed bytes, from the 16-byte granule its first byte is in: comp_cap bytes]
    // The 64 lanes read 64 places of the stream at once: from HBM that is 64 cache lines per load and, with a few waves on a CU, an L1
    // that holds none of them by the time they are wanted again (measured: 10 L2 requests per load, the kernel 4 x slower at two waves
    // per SIMD than at one).  In LDS every look at the stream is a ds_read.
    extern __shared__ uint32_t lds[];
    const int m = (int)blockIdx.x, lane = (int)threadIdx.x;
    uint8_t* base = (uint8_t*)lds;
    uint32_t* cw = lds + STRIDE_WORDS + 1;                                 // + 1: 16-byte aligned (STRIDE_WORDS is odd... 612 words = 2448 bytes)
    Lane H;
    H.lit = (uint16_t*)base; H.A = base + OFF_A; H.B = base + OFF_B; H.C = (uint16_t*)(base + OFF_C); H.ring = (uint32_t*)(base + OFF_RING); H.stage = (uint32_t*)(base + OFF_STAGE);
    const uint8_t* in = comp + blocks[m].in_off;
    const uint32_t skew = (uint32_t)((uintptr_t)in & 15u);
    H.src = in - skew; H.total = skew + blocks[m].in_len;
    H.buf = 0; H.cnt = 0; H.nextw = 0; H.rd = 4; H.ld = 0; H.outp = 0; H.ntok = 0; H.nflushed = 0; H.state = ST_HEADER; H.last = 0; H.inflight = false;
    H.pend[0] = H.pend[1] = H.pend[2] = H.pend[3] = 0;
    uint32_t* tk = tokens + (size_t)m * TOKCAP;
    uint32_t* slot = slots + ((size_t)m * 64 + (size_t)lane) * SLOT_TOKENS;    // this lane's tokens until the lanes before it have counted theirs
    H.tok = tk;
    const uint32_t limit = H.total * 8u;
    const uint32_t staged = (H.total + 15u + 16u) & ~15u;                   // the lanes read up to two words past the last byte
    if (staged > comp_cap) { if (lane == 0) { ntok[m] = NTOK_FALLBACK; out_len[m] = 0xFFFFFFFFu; } return; }      // an incompressible member: the one-lane kernel
#ifdef THJ_EXP
    unsigned long long hdbg[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, htl = clock64();
#endif
    for (uint32_t o = (uint32_t)lane * 16u; o < staged; o += 1024u) *(uint4*)((uint8_t*)cw + o) = *(const uint4*)(H.src + o);
    const uint32_t* w = cw;
    uint32_t tok_base = 0, out_base = 0, hpos = skew * 8u;
    bool fail = false;
    __syncthreads();
    HP_T(0);
    for (;;) {
        // ---- the block's header: lane 0
        int st = ST_FALLBACK, last = 0; uint32_t dstart = 0;
        // lane 0 reads the code lengths (a serial Huffman stream of its own), the wave builds the tables from them
        HeaderInfo hi{288, 32, false, false};
        bool hfall = false;
        if (lane == 0) { const HeaderW hw = parse_header_lengths_w(w, hpos, limit, H.lit, H.A, H.B, H.C); hi = hw.hi; last = hw.last; dstart = hw.end_bit; hfall = hw.fallback; }
        __syncthreads();
        HP_T(7);
        hi.hlit = __builtin_amdgcn_readfirstlane(hi.hlit); hi.hdist = __builtin_amdgcn_readfirstlane(hi.hdist);
        hi.build = __builtin_amdgcn_readfirstlane((int)hi.build) != 0; hi.ok = __builtin_amdgcn_readfirstlane((int)hi.ok) != 0;
        {
            WaveTab xt{lane};
            bool ok = hi.ok;
            ok = build_lit_wave(H.lit, H.C, H.A, hi.hlit, hi.build, xt) && ok;
            ok = build_dist_wave(H.A, H.B, H.A + hi.hlit, hi.hdist, hi.build, xt) && ok;
            if (lane == 0) st = (!hfall && hi.build && ok) ? ST_DECODE : ST_FALLBACK;
        }
        __syncthreads();
        st = __builtin_amdgcn_readfirstlane(st); last = __builtin_amdgcn_readfirstlane(last); dstart = (uint32_t)__builtin_amdgcn_readfirstlane((int)dstart);
        HP_T(1); HP_N(5, 1);
        if (st != ST_DECODE) { fail = true; break; }
        // ---- segments
        const uint32_t rem = limit > dstart ? limit - dstart : 0u;
        uint32_t seg = (rem + 63u) / 64u; if (seg < 64u) seg = 64u;
        const uint32_t border = dstart + (uint32_t)lane * seg, bnext = lane == 63 ? MARK : border + seg;
        uint32_t s = border;
        bool ch = lane > 0; int phase = 0;
        Seg r{border, 0, 0, 0};
        for (;;) {
            if (phase == 0) { if (ch) r = decode_segment<SEG_COUNT>(H.lit, H.A, H.B, w, limit, border - seg, border, (uint32_t*)nullptr, 0u, WaveGpu{}); }
            else if (ch) r = decode_segment<SEG_SLOT>(H.lit, H.A, H.B, w, limit, s, bnext, slot, SLOT_TOKENS, WaveGpu{});
            HP_N(6, 1);
            if (phase == 0) { if (ch && r.e < MARK) s = r.e; ch = true; phase = 1; HP_T(2); continue; }
            uint32_t ns = (uint32_t)__shfl_up((int)r.e, 1); if (lane == 0) ns = s;
            ch = ns < MARK && ns != s;                                        // a lane that failed says nothing about the next one's start
            if (!__any((int)ch)) break;
            s = ch ? ns : s;
        }
        HP_T(3);
        // the first lane that did not reach its border ended the block (or the stream is bad); the lanes behind it decoded nothing real
        const uint64_t markm = __ballot(r.e >= MARK);
        const int el = markm ? __builtin_ctzll(markm) : 64;
        if (el == 64 || (uint32_t)__builtin_amdgcn_readlane((int)r.e, el & 63) != MARK_EOB) { fail = true; break; }
        if (lane > el) { s = MARK_NONE; r.nt = 0; r.ob = 0; }
        const uint32_t inc_nt = wave_incl_scan(r.nt), inc_ob = wave_incl_scan(r.ob);
        const uint32_t tot_nt = (uint32_t)__builtin_amdgcn_readlane((int)inc_nt, 63), tot_ob = (uint32_t)__builtin_amdgcn_readlane((int)inc_ob, 63);
        if (tok_base + tot_nt > TOKCAP || out_base + tot_ob > 65536u) { fail = true; break; }
        // the tokens to their places: from the slots -- or, when a lane's did not fit its slot, by a pass that decodes them again
        if (!__any((int)(r.nt > SLOT_TOKENS))) {
            if (__any((int)!compact_segment(slot, r.nt, tk + tok_base + inc_nt - r.nt, out_base + inc_ob - r.ob))) { fail = true; break; }
        } else {
            const Seg f = decode_segment<SEG_FINAL>(H.lit, H.A, H.B, w, limit, s, bnext, tk + tok_base + inc_nt - r.nt, out_base + inc_ob - r.ob, WaveGpu{});
            if (__any((int)((lane <= el && f.e == MARK_ERR) || f.nt != r.nt || f.ob != r.ob))) { fail = true; break; }
        }
        HP_T(4);
        tok_base += tot_nt; out_base += tot_ob;
        hpos = (uint32_t)__builtin_amdgcn_readlane((int)r.eob_pos, el);
        if (last) break;
        __syncthreads();                                                  // the tables are about to be rebuilt
    }
    if (lane == 0) { ntok[m] = fail ? NTOK_FALLBACK : tok_base; out_len[m] = fail ? 0xFFFFFFFFu : out_base; }
#ifdef THJ_EXP
    if (lane == 0) { for (int k = 0; k < 10; ++k) if (hdbg[k]) atomicAdd(&thj_huffp_dbg[k], hdbg[k]); atomicAdd(&thj_huffp_dbg[12], 1ull); atomicAdd(&thj_huffp_dbg[13], (unsigned long long)tok_base); }
#endif
}

// one wave per member.  buf = the member's output from `origin` on (at least the last 32 KiB: DEFLATE's reach); when a batch does not
// fit, what is complete goes to HBM in 16-byte pieces and the buffer slides down.  Members kernel 1 refused are listed for the one-lane kernel.
#ifdef THJ_EXP
// developer build: where a wave of thj_k_lz spends its clocks (lane 0's s_memtime deltas, summed over all waves)
__device__ unsigned long long thj_lz_dbg[16];
extern "C" int thj_lz_dbg_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(thj_lz_dbg), sizeof thj_lz_dbg) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(thj_lz_dbg), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#define LZ_T(i) do { const unsigned long long t__ = clock64(); dbg[i] += t__ - tl; tl = t__; } while (0)
#define LZ_N(i, v) do { dbg[i] += (v); } while (0)
#else
#define LZ_T(i) do { } while (0)
#define LZ_N(i, v) do { } while (0)
#endif
__global__ __launch_bounds__(64) void thj_k_lz(const uint32_t* __restrict__ tokens, const uint32_t* __restrict__ ntok, int n_blocks, uint8_t* __restrict__ out,
                                               uint32_t* __restrict__ out_len, uint32_t* __restrict__ fb_list, uint32_t* __restrict__ fb_count) {
    using namespace inf2;
    __shared__ __attribute__((aligned(16))) uint8_t buf[LZ_CAP + 64];
    const int m = (int)blockIdx.x, lane = (int)threadIdx.x;
    const uint32_t n = ntok[m];
    if (n == NTOK_FALLBACK) { if (lane == 0) fb_list[atomicAdd(fb_count, 1u)] = (uint32_t)m; return; }
    const uint32_t* tk = tokens + (size_t)m * TOKCAP;
    uint8_t* dst = out + ((size_t)m << 16);
    uint32_t origin = 0, flushed = 0, pos = 0, i0 = 0;
    bool slid = false;
    uint32_t tok = (uint32_t)lane < n ? tk[lane] : 0u;
#ifdef THJ_EXP
    unsigned long long dbg[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tl = clock64();
#endif
    while (i0 < n) {
        // the next batch's tokens, on their way while this batch is done.  The load is unconditional (the index clamped, a lane past the end
        // reads the last token and never looks at it): around a load under a branch the compiler put `s_waitcnt vmcnt(0)` right behind it
        // -- the wave then sat out the whole round trip at the top of every batch, 1 500 of a batch's 4 900 clocks (THJ_EXP build, lz_timing)
        const uint32_t nx = i0 + 64u + (uint32_t)lane;
        const uint32_t tok_next = tk[nx < n ? nx : n - 1u];
        const bool valid = i0 + (uint32_t)lane < n;
        const bool is_m = valid && (tok >> 31);
        const uint32_t len = !valid ? 0u : is_m ? ((tok >> 15) & 255u) + 3u : 1u;
        const uint32_t incl = wave_incl_scan(len);
        const uint32_t room = origin + (uint32_t)LZ_CAP - pos;
        const uint64_t takes = __ballot(valid && incl <= room);                  // a prefix of the lanes
        const int ntake = __popcll(takes);
        const int nvalid = (int)(n - i0 < 64u ? n - i0 : 64u);
        if (ntake < nvalid && !slid) {
            // ---- no room for the whole batch: what is complete goes out, the buffer slides down, and the batch is looked at again
            // (room after a slide: at least 8 KiB less 15 bytes; a batch that is longer still is done in pieces)
            for (uint32_t o = flushed + (uint32_t)lane * 16u; o + 16u <= pos; o += 1024u) *(uint4*)(dst + o) = *(const uint4*)&buf[o - origin];
            if ((pos & ~15u) > flushed) flushed = pos & ~15u;
            const uint32_t no = pos > (uint32_t)LZ_HIST ? (pos - (uint32_t)LZ_HIST) & ~15u : 0u;
            if (no > origin) {
                const uint32_t shift = no - origin, keep = pos - no;
                for (uint32_t o = (uint32_t)lane * 16u; o < keep; o += 1024u) { const uint4 v = *(const uint4*)&buf[shift + o]; *(uint4*)&buf[o] = v; }
                origin = no;
            }
            slid = true;
            LZ_T(5);
            continue;
        }
        slid = false;
        if (ntake == 0) break;                                                   // cannot happen (a token is at most 258 bytes); reported below as a failure
        const bool take = ((takes >> lane) & 1ull) != 0;
        const uint32_t a = pos + incl - len - origin;                            // where this lane's bytes go in buf
        if (take && !is_m) buf[a] = (uint8_t)tok;
        const uint32_t dist = (tok & 0x7FFFu) + 1u;
        const uint32_t s = a - dist;                                             // >= 0: kernel 1 checked dist <= position, and origin keeps 32 KiB
        const uint32_t send = s + (len < dist ? len : dist);
        uint64_t pend = __ballot(take && is_m);
        LZ_T(0); LZ_N(6, 1);
        while (pend) {
            LZ_N(7, 1);
            const int f = __builtin_ctzll(pend);
            const uint32_t hwm = (uint32_t)__builtin_amdgcn_readlane((int)a, f);        // everything below the first open match is final
            const bool mine = ((pend >> lane) & 1ull) != 0;
            const bool rdy = mine && send <= hwm;
            const bool small = rdy && len <= 32u && dist >= len;
            if (small) {
                // up to 32 bytes, source and destination apart: (unaligned) 8-byte reads, then exactly len bytes written
                // (only the words the match needs: the resolver is bound by LDS conflicts of these scattered unaligned accesses -- four waves on
                // a member's window ran exactly as fast as one, 1.58 against 1.62 ms per 5.9 k members --, and a lane that does not read does not collide)
                uint64_t v0, v1 = 0, v2 = 0, v3 = 0;
                __builtin_memcpy(&v0, &buf[s], 8);
                if (len > 8u) __builtin_memcpy(&v1, &buf[s + 8], 8);
                if (len > 16u) __builtin_memcpy(&v2, &buf[s + 16], 8);
                if (len > 24u) __builtin_memcpy(&v3, &buf[s + 24], 8);
                // every value is in a register of its own before the first write: with the words shifted down through one register pair
                // (v0 = v1 after v0's write, ...) each write waited for the one before it -- its data register was about to be overwritten --
                // and a match of 20 bytes was five LDS round trips instead of two (THJ_EXP build, lz_timing: 870 clocks a round)
                const uint32_t q = len >> 3;                                             // whole 8-byte words
                const uint64_t vt = q == 0u ? v0 : q == 1u ? v1 : q == 2u ? v2 : v3;     // the word the last len & 7 bytes come from
                const uint32_t sh2 = (len & 4u) ? 32u : 0u;
                const uint32_t t4 = (uint32_t)vt;
                const uint16_t t2 = (uint16_t)(vt >> sh2);
                const uint8_t t1 = (uint8_t)(vt >> (sh2 + ((len & 2u) ? 16u : 0u)));
                const uint32_t wt = a + (len & ~7u);
                if (q >= 1u) __builtin_memcpy(&buf[a], &v0, 8);
                if (q >= 2u) __builtin_memcpy(&buf[a + 8u], &v1, 8);
                if (q >= 3u) __builtin_memcpy(&buf[a + 16u], &v2, 8);
                if (q >= 4u) __builtin_memcpy(&buf[a + 24u], &v3, 8);
                if (len & 4u) __builtin_memcpy(&buf[wt], &t4, 4);
                if (len & 2u) __builtin_memcpy(&buf[wt + ((len & 4u) ? 4u : 0u)], &t2, 2);
                if (len & 1u) buf[wt + (len & 6u)] = t1;
            }
            // the long and the self-overlapping ones: the whole wave on each, 64 bytes a step; a match that overlaps itself repeats its
            // first dist bytes, so every byte is read from those (all lanes read before any writes)
            uint64_t big = __ballot(rdy && !small);
            LZ_T(1); LZ_N(8, __popcll(big));
            while (big) {
                const int g = __builtin_ctzll(big); big &= big - 1;
                const uint32_t ga = (uint32_t)__builtin_amdgcn_readlane((int)a, g), gs = (uint32_t)__builtin_amdgcn_readlane((int)s, g);
                const uint32_t gl = (uint32_t)__builtin_amdgcn_readlane((int)len, g), gd = (uint32_t)__builtin_amdgcn_readlane((int)dist, g);
                if (gd >= gl) { for (uint32_t k = (uint32_t)lane; k < gl; k += 64u) { const uint8_t b = buf[gs + k]; buf[ga + k] = b; } }
                else { for (uint32_t k = (uint32_t)lane; k < gl; k += 64u) { const uint8_t b = buf[gs + k % gd]; buf[ga + k] = b; } }
            }
            pend &= ~__ballot(rdy);
            LZ_T(2);
        }
        pos += (uint32_t)__builtin_amdgcn_readlane((int)incl, ntake - 1);
        i0 += (uint32_t)ntake;
        tok = ntake == 64 ? tok_next : (i0 + (uint32_t)lane < n ? tk[i0 + (uint32_t)lane] : 0u);
        { const uint32_t probe = tok; LZ_T(3); if (probe == 0xFFFFFFFFu) LZ_N(9, 1); }        // (the wait for the next batch's tokens lands here)
    }
    // ---- what is left: whole 16-byte pieces, then the tail
    for (uint32_t o = flushed + (uint32_t)lane * 16u; o + 16u <= pos; o += 1024u) *(uint4*)(dst + o) = *(const uint4*)&buf[o - origin];
    if ((pos & ~15u) > flushed) flushed = pos & ~15u;
    if (flushed + (uint32_t)lane < pos) dst[flushed + (uint32_t)lane] = buf[flushed + (uint32_t)lane - origin];
    if (lane == 0) out_len[m] = i0 < n ? 0xFFFFFFFFu : pos;
#ifdef THJ_EXP
    LZ_T(4);
    if (lane == 0) { for (int k = 0; k < 12; ++k) if (dbg[k]) atomicAdd(&thj_lz_dbg[k], dbg[k]); atomicAdd(&thj_lz_dbg[12], 1ull); }
#endif
}

// Which inflater.  Default: the two kernels above, then the one-lane kernel over whatever members they handed back (stored blocks,
// members of more than TOKCAP symbols, corrupt streams -- normally none: its workgroups read a zero count and leave).
// THJ_INFLATE=one: the round-2 kernel alone; =lanes: the round-2 lane-per-member experiment; =member: entropy decoding one member per LANE
// (thj_k_huff, THJ_HUFF_LPW=16|32|64 lanes per wave) instead of one per wave (thj_k_huffp).
static int launch_inflate(thj_ctx* c, const uint8_t* d_comp, const thj_bgzf_block* d_blocks, int64_t nb, uint8_t* d_out, uint32_t* d_len, uint32_t max_in_len = 0) {
    static const char* force = getenv("THJ_INFLATE");
    static const int lpw = getenv("THJ_HUFF_LPW") ? atoi(getenv("THJ_HUFF_LPW")) : 64;
    if (force && force[0] == 'l') { hipLaunchKernelGGL(thj_k_inflate_lanes, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, c->stream, d_comp, d_blocks, (int)nb, d_out, d_len); return THJ_OK; }
    const int64_t grid1 = nb < (1 << 20) ? nb : (1 << 20);                    // a workgroup per member: the dispatcher balances the tail
    if (force && force[0] == 'o') {
        hipLaunchKernelGGL(thj_k_inflate, dim3((unsigned)grid1), dim3(64), 0, c->stream, d_comp, d_blocks, (int)nb, d_out, d_len, (const uint32_t*)nullptr, (const uint32_t*)nullptr);
        return THJ_OK;
    }
    // the token streams between the two kernels take TOKCAP words per member (more than the member's 64 KiB of output): a launch of
    // many members goes through in pieces of INFL_CHUNK of them over one scratch buffer, so that the scratch stays bounded (2 GB: 80 KiB of tokens and 160 KiB of slots a member)
    // whatever the caller hands over
    const int64_t INFL_CHUNK = getenv("THJ_INFLATE_CHUNK") && atoll(getenv("THJ_INFLATE_CHUNK")) > 0 ? atoll(getenv("THJ_INFLATE_CHUNK")) : 8192;      // (the variable: tests)
    const int64_t per = nb < INFL_CHUNK ? nb : INFL_CHUNK;
    const size_t need = (size_t)per * ((size_t)inf2::TOKCAP * 4 + (size_t)64 * inf2::SLOT_TOKENS * 4 + 8) + 256;      // token lists, the lanes' slots, counts
    if (c->infl_tmp_cap < need) {
        HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_infl_tmp); c->d_infl_tmp = nullptr; c->infl_tmp_cap = 0;
        HIPCHK(hipMalloc(&c->d_infl_tmp, need + need / 4)); c->infl_tmp_cap = need + need / 4;
    }
    uint32_t* d_tok = (uint32_t*)c->d_infl_tmp;
    uint32_t* d_slots = d_tok + (size_t)per * inf2::TOKCAP;
    uint32_t* d_ntok = d_slots + (size_t)per * 64 * inf2::SLOT_TOKENS;
    uint32_t* d_fb = d_ntok + per;
    uint32_t* d_fbn = d_fb + per;
    for (int64_t at = 0; at < nb; at += per) {
        const int64_t n = nb - at < per ? nb - at : per;
        const thj_bgzf_block* blk = d_blocks + at;
        uint8_t* out = d_out + ((size_t)at << 16);
        uint32_t* len = d_len + at;
        HIPCHK(hipMemsetAsync(d_fbn, 0, 4, c->stream));
        const dim3 g((unsigned)((n + 63) / 64));
        if (!(force && force[0] == 'm')) {
            // LDS per wave: the tables + the largest member's compressed bytes (max_in_len = 0: the caller does not know -- 24 KiB, what a
            // 64 KiB BAM member comes to at worst in practice); members beyond 60 KiB of LDS go to the one-lane kernel
            uint32_t cap = max_in_len ? max_in_len + 64u : 24576u;
            cap = (cap + 255u) & ~255u;
            if (cap > 61440u - 2560u) cap = 61440u - 2560u;
            hipLaunchKernelGGL(thj_k_huffp, dim3((unsigned)n), dim3(64), (size_t)(inf2::STRIDE_WORDS + 1) * 4 + cap, c->stream, d_comp, blk, (int)n, d_tok, d_slots, d_ntok, len, cap);
        }
        else if (lpw == 16) hipLaunchKernelGGL(thj_k_huff<16>, g, dim3(256), 0, c->stream, d_comp, blk, (int)n, d_tok, d_ntok, len);
        else if (lpw == 32) hipLaunchKernelGGL(thj_k_huff<32>, g, dim3(128), 0, c->stream, d_comp, blk, (int)n, d_tok, d_ntok, len);
        else hipLaunchKernelGGL(thj_k_huff<64>, g, dim3(64), 0, c->stream, d_comp, blk, (int)n, d_tok, d_ntok, len);
        hipLaunchKernelGGL(thj_k_lz, dim3((unsigned)n), dim3(64), 0, c->stream, d_tok, d_ntok, (int)n, out, len, d_fb, d_fbn);
        hipLaunchKernelGGL(thj_k_inflate, dim3((unsigned)(n < 256 ? n : 256)), dim3(64), 0, c->stream, d_comp, blk, (int)n, out, len, (const uint32_t*)d_fb, (const uint32_t*)d_fbn);
    }
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

// ------------------------------------------------------------------------------------------------ C ABI

extern "C" int thj_bgzf_inflate(thj_ctx* c, const uint8_t* comp, int64_t comp_bytes, const thj_bgzf_block* blocks, int64_t n_blocks,
                                uint8_t* out, uint32_t* out_len, int32_t on_device) {
    // Test / tool entry point: inflates n_blocks members.  on_device == 0: comp / blocks are host arrays, out (65536 bytes per
    // block) and out_len host arrays; != 0: all four are device arrays and the call only enqueues.
    if (!c || n_blocks < 0 || (n_blocks > 0 && (!comp || !blocks || !out || !out_len))) { thj_set_error("thj_bgzf_inflate: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (n_blocks == 0) return THJ_OK;
    const uint8_t* d_comp = comp; const thj_bgzf_block* d_blocks = blocks; uint8_t* d_out = out; uint32_t* d_len = out_len;
    void *t0 = nullptr, *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;
    if (!on_device) {
        HIPCHK(hipMalloc(&t0, (size_t)comp_bytes + 64)); HIPCHK(hipMalloc(&t1, (size_t)n_blocks * sizeof(thj_bgzf_block)));
        HIPCHK(hipMalloc(&t2, (size_t)n_blocks << 16)); HIPCHK(hipMalloc(&t3, (size_t)n_blocks * 4));
        HIPCHK(hipMemcpyAsync(t0, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(t1, blocks, (size_t)n_blocks * sizeof(thj_bgzf_block), hipMemcpyHostToDevice, c->stream));
        d_comp = (const uint8_t*)t0; d_blocks = (const thj_bgzf_block*)t1; d_out = (uint8_t*)t2; d_len = (uint32_t*)t3;
    }
    { uint32_t mx = 0; if (!on_device) for (int64_t k = 0; k < n_blocks; ++k) mx = blocks[k].in_len > mx ? blocks[k].in_len : mx;
      const int rc_ = launch_inflate(c, d_comp, d_blocks, n_blocks, d_out, d_len, mx); if (rc_) return rc_; }
    HIPCHK(hipGetLastError());
    if (!on_device) {
        HIPCHK(hipMemcpyAsync(out_len, t3, (size_t)n_blocks * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(out, t2, (size_t)n_blocks << 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(t0); hipFree(t1); hipFree(t2); hipFree(t3);
    }
    return THJ_OK;
}

// ================================================================================================ BAM records on the device
//
// After the inflate every member sits in its own 64 KiB slot.  bam_write1 starts a new member rather than let a record
// straddle two (bgzf_flush_try), so the members of a file can be walked independently:
//   thj_k_walk        one wave per member follows the block_size chain (through a 4 KiB LDS window) and notes where records start
//   thj_k_parse_hits  one workgroup per member, one thread per record: BAMHitFactory::get_hit_from_buf (bwt_map.cpp:1101-1452)
//                     straight from the record bytes -> (insert_id, thj_hit, thj_span_hit), plus the shard's id-range filter
//   (scan + scatter)  the records the factory keeps, densely, in file order
// and then the k-way merge by read id that look_for_hit_group does with stream look-aheads becomes array work over the
// (dense) id range of the shard: per file "first record / record count of id", a visited flag per id, prefix sums for the
// rows and the CSR offsets, one scatter per file.

#include <hipcub/hipcub.hpp>
#include <atomic>
#include <chrono>

namespace ing {

static constexpr int MAXREC = 1824;             // a BAM record is at least 36 bytes: 65536 / 36
struct Hit16 { uint32_t ref_id; int32_t left, right; uint32_t meta; };                 // == thj_hit:  flags | edit_dist << 8 | mismatches << 16 | read_len << 24
struct Hit32 { uint32_t ref_id; int32_t left; uint32_t meta; uint32_t cigar[5]; };     // == thj_span_hit: flags | mismatches << 8 | edit_dist << 16 | n_cigar << 24
static_assert(sizeof(Hit16) == sizeof(thj_hit) && sizeof(Hit32) == sizeof(thj_span_hit), "hit layouts");

struct FileInfo { uint32_t first_block, n_blocks, first_skip, kind, tid_base, n_tid, rec_base, n_rec; };
enum { KIND_HITS = 0, KIND_READS = 1 };
enum { ST_CORRUPT = 0, ST_STRADDLE = 1, ST_XF = 2, ST_CIGAR = 3, ST_MISSING_READ = 4, ST_N };

__device__ __forceinline__ uint32_t ld32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }

// One WAVE per member: the block_size chain is a chain of dependent loads (~280 records a member), from HBM 0.44 ms per launch with a
// thread per member.  The wave copies the member through a 4 KiB LDS window (one coalesced sweep per window), lane 0 follows the chain
// in LDS.
__global__ __launch_bounds__(64) void thj_k_walk(const uint8_t* __restrict__ infl, const uint32_t* __restrict__ len, const uint8_t* __restrict__ blk_file,
                                                 const FileInfo* __restrict__ files, int n_blocks, uint16_t* __restrict__ rec_off, uint32_t* __restrict__ cnt,
                                                 unsigned int* __restrict__ status) {
    constexpr uint32_t WIN = 4096, WINX = WIN + 16;               // a block_size field may begin in the window's last bytes
    __shared__ __attribute__((aligned(16))) uint8_t win[WINX];
    const int b = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (b >= n_blocks) return;
    const uint32_t L = len[b];
    if (L == 0xFFFFFFFFu || L > 65536u) { if (lane == 0) { atomicExch(&status[ST_CORRUPT], 1u); cnt[b] = 0; } return; }
    const FileInfo f = files[blk_file[b]];
    uint32_t p = (uint32_t)b == f.first_block ? f.first_skip : 0u, k = 0;
    const uint8_t* base = infl + ((size_t)b << 16);
    while (p + 4 <= L && k < (uint32_t)MAXREC) {
        const uint32_t c0 = p & ~(WIN - 1);
        for (uint32_t i = (uint32_t)lane * 16u; i < WINX && c0 + i + 16u <= 65536u; i += 1024u) *(uint4*)&win[i] = *(const uint4*)(base + c0 + i);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t stop = 0;
        if (lane == 0) {
            while (p + 4 <= L && k < (uint32_t)MAXREC && p + 4 <= c0 + WINX) {
                const uint32_t bs = ld32(&win[p - c0]);
                if (bs < 32u || p + 4 + bs > L) { stop = 1; break; }
                rec_off[(size_t)b * MAXREC + k++] = (uint16_t)p;
                p += 4 + bs;
            }
        }
        p = (uint32_t)__builtin_amdgcn_readfirstlane((int)p); k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
        stop = (uint32_t)__builtin_amdgcn_readfirstlane((int)stop);
        __builtin_amdgcn_wave_barrier();                          // the window is free again
        if (stop) break;
    }
    if (lane == 0) {
        if (p != L) atomicExch(&status[ST_STRADDLE], 1u);       // a record runs past the member (or garbage): not samtools' layout
        cnt[b] = k;
    }
}

struct ParseOut { uint32_t* id; uint32_t* valid; Hit16* h16; Hit32* h32; uint32_t* loc; };

// get_hit_from_buf for record `d` (after its block_size field, `bs` bytes); false = the factory drops the record
__device__ bool parse_hit(const uint8_t* d, uint32_t bs, const uint32_t* tid2ref, uint32_t n_tid, int max_report_intron, uint32_t& id, Hit16& h16, Hit32& h32,
                          unsigned int* status) {
    const int32_t tid = (int32_t)ld32(d), pos = (int32_t)ld32(d + 4), mtid = (int32_t)ld32(d + 20);
    const uint32_t bin_mq_nl = ld32(d + 8), flag_nc = ld32(d + 12), l_seq = ld32(d + 16);
    const uint32_t l_rn = bin_mq_nl & 0xFF, n_cig = flag_nc & 0xFFFF, flag = flag_nc >> 16;
    // the record's own header must fit its block_size before anything walks it (a damaged file is an error, not a wild read)
    if (bs < 32u || l_rn == 0u || l_seq > 0x7FFFFFFFu || 32ull + l_rn + 4ull * n_cig + ((unsigned long long)l_seq + 1ull) / 2ull + l_seq > (unsigned long long)bs) { atomicAdd(&status[ST_CORRUPT], 1u); id = 0; return false; }
    // qname "<id>|<offset>:<segment>:<segments>" (tophat.py:2948): insert_id = atoi, end = (segment + 1 == segments)
    const uint8_t* q = d + 32;
    uint32_t v = 0, i = 0;
    while (i + 1 < l_rn && q[i] >= '0' && q[i] <= '9') { v = v * 10u + (uint32_t)(q[i] - '0'); ++i; }
    id = v;
    bool end = true;
    {
        int pipe = -1;
        for (uint32_t k = 0; k + 1 < l_rn; ++k) if (q[k] == '|') pipe = (int)k;
        if (pipe >= 0) {
            bool colon = false;
            for (uint32_t k = (uint32_t)pipe + 1; k + 1 < l_rn; ++k) if (q[k] == ':') colon = true;
            if (colon) {                                   // strtoul(a) ':' strtoul(b) ':' strtoul(c), missing fields stay 0 (bwt_map.cpp:1125-1143)
                uint32_t k = (uint32_t)pipe + 1, bb = 0, cc = 0;
                while (k + 1 < l_rn && q[k] >= '0' && q[k] <= '9') ++k;
                if (k + 1 < l_rn && q[k] == ':') {
                    ++k;
                    while (k + 1 < l_rn && q[k] >= '0' && q[k] <= '9') { bb = bb * 10u + (uint32_t)(q[k] - '0'); ++k; }
                    if (k + 1 < l_rn && q[k] == ':') { ++k; while (k + 1 < l_rn && q[k] >= '0' && q[k] <= '9') { cc = cc * 10u + (uint32_t)(q[k] - '0'); ++k; } }
                }
                end = (bb + 1 == cc);
            }
        }
    }
    if (tid < 0 || (flag & 4u)) return false;
    uint32_t pp = 32 + l_rn;
    int right = pos, read_len = 0, gap = 0, ind = 0, n32 = 0;
    bool spliced = false;
    h32.cigar[0] = h32.cigar[1] = h32.cigar[2] = h32.cigar[3] = h32.cigar[4] = 0;
    for (uint32_t c = 0; c < n_cig; ++c) {
        const uint32_t w = ld32(d + pp); pp += 4;
        const uint32_t len = w >> 4, bop = w & 0xF;
        if (len == 0) return false;
        uint32_t op;
        switch (bop) {
        case 0: case 7: case 8: op = 1; right += (int)len; read_len += (int)len; break;
        case 1: op = 3; read_len += (int)len; gap += (int)len; ind += (int)len; break;
        case 2: op = 5; right += (int)len; gap += (int)len; ind += (int)len; break;
        case 4: op = 13; read_len += (int)len; break;
        case 5: continue;
        case 6: op = 15; break;
        case 3: op = 11; spliced = true; if ((int)len > max_report_intron) return false; right += (int)len; break;
        default: return false;
        }
        if (n32 < 5) h32.cigar[n32] = (op << 28) | (len & 0x0FFFFFFFu);
        ++n32;
    }
    if (mtid >= 0 && mtid != tid) return false;
    pp += (l_seq + 1) / 2 + l_seq;
    int nm = 0; char xs = 0;
    while (pp + 3 <= bs) {
        const char t0 = (char)d[pp], t1 = (char)d[pp + 1], ty = (char)d[pp + 2];
        pp += 3;
        {   // a tag's value must lie inside the record
            const uint32_t fixed = (ty == 'A' || ty == 'c' || ty == 'C') ? 1u : (ty == 's' || ty == 'S') ? 2u : (ty == 'i' || ty == 'I' || ty == 'f') ? 4u : ty == 'd' ? 8u : ty == 'B' ? 5u : 0u;
            if (fixed > bs - pp) { atomicAdd(&status[ST_CORRUPT], 1u); return false; }
        }
        int iv = 0; bool isint = false;
        switch (ty) {
        case 'A': if (t0 == 'X' && t1 == 'S') xs = (char)d[pp]; pp += 1; break;
        case 'c': iv = (int8_t)d[pp]; isint = true; pp += 1; break;
        case 'C': iv = d[pp]; isint = true; pp += 1; break;
        case 's': iv = (int16_t)(d[pp] | (d[pp + 1] << 8)); isint = true; pp += 2; break;
        case 'S': iv = (int)(d[pp] | (d[pp + 1] << 8)); isint = true; pp += 2; break;
        case 'i': case 'I': iv = (int)ld32(d + pp); isint = true; pp += 4; break;
        case 'f': pp += 4; break;
        case 'd': pp += 8; break;
        case 'Z': case 'H':
            if (t0 == 'X' && t1 == 'F') atomicExch(&status[ST_XF], 1u);
            while (pp < bs && d[pp]) ++pp;
            ++pp;
            break;
        case 'B': { const char st = (char)d[pp]; const uint32_t cnt = ld32(d + pp + 1); pp += 5 + cnt * ((st == 'c' || st == 'C') ? 1u : (st == 's' || st == 'S') ? 2u : 4u); break; }
        default: pp = bs; break;
        }
        if (isint && t0 == 'N' && t1 == 'M') nm = iv;
    }
    const uint32_t ref_id = (uint32_t)tid < n_tid ? tid2ref[tid] : 0u;
    if (ref_id == 0) return false;
    if (n32 > 5) { atomicExch(&status[ST_CIGAR], 1u); return false; }
    const uint32_t mism = (uint32_t)(uint8_t)((uint8_t)nm - (uint8_t)ind), ed = (uint32_t)(uint8_t)(mism + (uint32_t)gap);
    const bool anti = (flag & 0x10u) != 0;
    const uint32_t fl = (anti ? 1u : 0u) | (end ? 2u : 0u);
    h16.ref_id = ref_id; h16.left = pos; h16.right = right;
    h16.meta = fl | (ed << 8) | (mism << 16) | ((uint32_t)(read_len > 255 ? 255 : read_len) << 24);
    h32.ref_id = ref_id; h32.left = pos;
    h32.meta = (fl | ((spliced && xs == '-') ? 4u : 0u)) | (mism << 8) | (ed << 16) | ((uint32_t)n32 << 24);
    return true;
}

__global__ __launch_bounds__(256) void thj_k_parse(const uint8_t* __restrict__ infl, const uint8_t* __restrict__ blk_file, const FileInfo* __restrict__ files,
                                                   const uint16_t* __restrict__ rec_off, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ base,
                                                   const uint32_t* __restrict__ tid2ref, uint32_t begin_id, uint32_t end_id, int max_report_intron, int want32,
                                                   ParseOut o, unsigned int* status) {
    const int b = blockIdx.x;
    const FileInfo f = files[blk_file[b]];
    const uint32_t n = cnt[b];
    const uint8_t* slot = infl + ((size_t)b << 16);
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) {
        const uint32_t p = rec_off[(size_t)b * MAXREC + k];
        const uint32_t bs = ld32(slot + p);
        const uint8_t* d = slot + p + 4;
        const uint32_t i = base[b] + k;
        uint32_t id = 0; bool ok;
        if (f.kind == KIND_READS) {
            const uint32_t l_rn = ld32(d + 8) & 0xFF;
            uint32_t v = 0;
            for (uint32_t c = 0; c + 1 < l_rn && d[32 + c] >= '0' && d[32 + c] <= '9'; ++c) v = v * 10u + (uint32_t)(d[32 + c] - '0');
            id = v; ok = true;
            o.loc[i] = ((uint32_t)b << 16) | p;
        } else {
            Hit16 h16; Hit32 h32;
            ok = parse_hit(d, bs, tid2ref + f.tid_base, f.n_tid, max_report_intron, id, h16, h32, status);
            if (ok) { if (want32) o.h32[i] = h32; else o.h16[i] = h16; }
        }
        if (id < begin_id || id >= end_id) ok = false;
        o.id[i] = id;
        o.valid[i] = ok ? 1u : 0u;
    }
}

// records the factory keeps, densely, file after file (dst = exclusive prefix sum of valid)
__global__ __launch_bounds__(256) void thj_k_compact(int64_t n, const uint32_t* __restrict__ valid, const uint32_t* __restrict__ dst, ParseOut in, ParseOut out, int want32,
                                                     const uint32_t* __restrict__ is_reads) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (!valid[i]) continue;
        const uint32_t j = dst[i];
        out.id[j] = in.id[i];
        if (is_reads[i]) out.loc[j] = in.loc[i];
        else if (want32) out.h32[j] = in.h32[i];
        else out.h16[j] = in.h16[i];
    }
}

__global__ __launch_bounds__(256) void thj_k_mark_reads(const FileInfo* files, int n_files, const uint32_t* base, uint32_t* is_reads, int64_t n) {
    // record -> "belongs to a reads file" (files are few: linear scan)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t r = 0;
        for (int f = 0; f < n_files; ++f) if (files[f].kind == KIND_READS && i >= files[f].rec_base && i < (int64_t)files[f].rec_base + files[f].n_rec) r = 1;
        is_reads[i] = r;
    }
    (void)base;
}

// ---- merge by read id over the dense id range [id0, id0 + span)
// first[f][idl] = compact index of the first record of that id in file f, cntf[f][idl] = how many
__global__ __launch_bounds__(256) void thj_k_runs_clipped(const uint32_t* __restrict__ id, uint32_t a, uint32_t b, uint32_t id0, uint32_t span, uint32_t* __restrict__ first,
                                                            uint32_t* __restrict__ cntf) {
    for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x) {
        const uint32_t idl = id[i] - id0;
        if (idl >= span) continue;                               // a mate outside the id range of the shard's segment hits
        if (i == a || id[i - 1] != id[i]) first[idl] = i;
        atomicAdd(&cntf[idl], 1u);
    }
}

// visited (look_for_hit_group's visiting set, tophat_amd/batch.py): some segment above the first has a hit -- or any segment
// when first-segment-only reads ride along (fusion / coverage search)
__global__ __launch_bounds__(256) void thj_k_visited(const uint32_t* __restrict__ cntf, int nseg, uint32_t span, int include_top0, uint32_t* __restrict__ vis) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < span; i += gridDim.x * blockDim.x) {
        uint32_t any = 0;
        for (int s = include_top0 ? 0 : 1; s < nseg; ++s) any |= cntf[(size_t)s * span + i];
        vis[i] = any ? 1u : 0u;
    }
}

__global__ __launch_bounds__(256) void thj_k_row_counts(const uint32_t* __restrict__ vis, const uint32_t* __restrict__ row, const uint32_t* __restrict__ cntf, int nseg, uint32_t span,
                                                        uint32_t* __restrict__ cell, const uint32_t* __restrict__ cnt_full, const uint32_t* __restrict__ cnt_last,
                                                        uint32_t* __restrict__ mcell, uint32_t* __restrict__ row_id, uint32_t id0) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < span; i += gridDim.x * blockDim.x) {
        if (!vis[i]) continue;
        const uint32_t r = row[i];
        for (int s = 0; s < nseg; ++s) cell[(size_t)r * nseg + s] = cntf[(size_t)s * span + i];
        if (mcell) { const uint32_t cf = cnt_full ? cnt_full[i] : 0u; mcell[r] = cf ? cf : (cnt_last ? cnt_last[i] : 0u); }
        row_id[r] = id0 + i;
    }
}

template <class T>
__global__ __launch_bounds__(256) void thj_k_scatter_hits(const uint32_t* __restrict__ id, const T* __restrict__ src, uint32_t a, uint32_t b, uint32_t id0, uint32_t span,
                                                          const uint32_t* __restrict__ vis, const uint32_t* __restrict__ row, const uint32_t* __restrict__ first,
                                                          const uint32_t* __restrict__ off, int nseg, int s, T* __restrict__ dst) {
    for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x) {
        const uint32_t idl = id[i] - id0;
        if (idl >= span || !vis[idl]) continue;                  // (long_spanning_reads: the id range is the first segment map's)
        dst[off[(size_t)row[idl] * nseg + s] + (i - first[idl])] = src[i];
    }
}

// long_spanning_reads' records and, beside them, the dense array of their first 16 bytes (thj_span_batch.hit_heads)
__global__ __launch_bounds__(256) void thj_k_scatter_span(const uint32_t* __restrict__ id, const Hit32* __restrict__ src, uint32_t a, uint32_t b, uint32_t id0, uint32_t span,
                                                          const uint32_t* __restrict__ vis, const uint32_t* __restrict__ row, const uint32_t* __restrict__ first,
                                                          const uint32_t* __restrict__ off, int nseg, int s, Hit32* __restrict__ dst, uint4* __restrict__ heads) {
    for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x) {
        const uint32_t idl = id[i] - id0;
        if (idl >= span || !vis[idl]) continue;
        const uint32_t at = off[(size_t)row[idl] * nseg + s] + (i - first[idl]);
        const Hit32 h = src[i];
        dst[at] = h;
        heads[at] = make_uint4(h.ref_id, (uint32_t)h.left, h.meta, h.cigar[0]);
    }
}

// mate hits: the mate's whole-read map when it has the id, else the mate's last segment map (find_gaps :3321-3348)
__global__ __launch_bounds__(256) void thj_k_scatter_mates(const uint32_t* __restrict__ id, const Hit16* __restrict__ src, uint32_t a, uint32_t b, uint32_t id0, uint32_t span,
                                                           const uint32_t* __restrict__ vis, const uint32_t* __restrict__ row, const uint32_t* __restrict__ first,
                                                           const uint32_t* __restrict__ moff, const uint32_t* __restrict__ cnt_full, int is_last, Hit16* __restrict__ dst) {
    for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x) {
        const uint32_t idl = id[i] - id0;
        if (idl >= span || !vis[idl]) continue;
        if (is_last && cnt_full && cnt_full[idl]) continue;
        dst[moff[row[idl]] + (i - first[idl])] = src[i];
    }
}

// reads: SEQ nibbles ("=ACMGRSVTWYHKDBN") -> {lo, hi, N} bit planes of thj_reads_pack; row = the read's row in the batch
__global__ __launch_bounds__(256) void thj_k_read_planes(const uint8_t* __restrict__ infl, const uint32_t* __restrict__ id, const uint32_t* __restrict__ loc, uint32_t a, uint32_t b,
                                                         uint32_t id0, uint32_t span, const uint32_t* __restrict__ vis, const uint32_t* __restrict__ row, int W, u64* __restrict__ planes,
                                                         uint16_t* __restrict__ rlen, uint32_t* __restrict__ seen, unsigned int* status,
                                                         uint8_t* __restrict__ quals = nullptr, int qstride = 0, uint32_t* __restrict__ row_loc = nullptr, int wide = 0) {
    for (uint32_t i = a + blockIdx.x * blockDim.x + threadIdx.x; i < b; i += gridDim.x * blockDim.x) {
        const uint32_t idl = id[i] - id0;
        if (idl >= span || !vis[idl]) continue;
        const uint32_t r = row[idl];
        const uint8_t* d = infl + ((size_t)(loc[i] >> 16) << 16) + (loc[i] & 0xFFFFu) + 4;
        if ((ld32(d + 12) >> 16) & 0x200u) continue;               // BAM_FQCFAIL: ReadStream::get_direct reads on past such a record (reads.cpp:556)
        if (atomicExch(&seen[r], 1u)) continue;                    // the first record of an id is the read (ReadStream::getRead)
        const uint32_t l_rn = ld32(d + 8) & 0xFF, n_cig = ld32(d + 12) & 0xFFFF, l_seq = ld32(d + 16);
        const uint8_t* sq = d + 32 + l_rn + 4 * n_cig;
        u64* pl = planes + (size_t)r * 3 * W;
        const uint32_t L = l_seq > (uint32_t)W * 64u ? (uint32_t)W * 64u : l_seq;
        for (int w = 0; w < W; ++w) {
            u64 lo = 0, hi = 0, nn = 0;
            // sixteen bases a step while whole groups are left, then base by base (a dependent byte load per two bases was 1.6 ms per launch)
            for (uint32_t k = 0; k < 64 && (uint32_t)w * 64 + k < L; ) {
                const uint32_t bi = (uint32_t)w * 64 + k;
                if (bi + 16 <= L) {
                    // eight byte loads in flight, not one 8-byte load: as `global_load_dwordx2` from these byte-aligned addresses the group came back
                    // wrong now and then (a run of 80 000 pairs lost ~200 alignments, different ones each time; byte loads: none in ten runs)
                    u64 v = 0;
                    if (wide) __builtin_memcpy(&v, sq + (bi >> 1), 8);          // THJ_PLANES_LOAD64 (developer switch): the one 8-byte load of round 3, for the reproducer
                    else {
#pragma unroll
                        for (int bb = 0; bb < 8; ++bb) v |= (u64)sq[(bi >> 1) + bb] << (8 * bb);
                    }
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint32_t nib = (uint32_t)(v >> (8 * (j >> 1) + ((j & 1) ? 0 : 4))) & 0xFu;
                        const u64 b0 = (nib == 2u) | (nib == 8u), b1 = (nib == 4u) | (nib == 8u), isn = !((nib == 1u) | (nib == 2u) | (nib == 4u) | (nib == 8u));
                        lo |= b0 << (k + j); hi |= b1 << (k + j); nn |= isn << (k + j);
                    }
                    k += 16;
                } else {
                    const uint32_t nib = (sq[bi >> 1] >> ((bi & 1) ? 0 : 4)) & 0xF;
                    const u64 b0 = (nib == 2u) | (nib == 8u), b1 = (nib == 4u) | (nib == 8u), isn = !((nib == 1u) | (nib == 2u) | (nib == 4u) | (nib == 8u));
                    lo |= b0 << k; hi |= b1 << k; nn |= isn << k;
                    ++k;
                }
            }
            pl[w] = lo; pl[W + w] = hi; pl[2 * W + w] = nn;
        }
        rlen[r] = (uint16_t)l_seq;
        if (quals) {                                               // phred+33 text, as thj_span_batch.quals wants it
            const uint8_t* q = sq + ((l_seq + 1) >> 1);
            uint8_t* dq = quals + (size_t)r * qstride;
            const uint32_t nq = l_seq < (uint32_t)qstride ? l_seq : (uint32_t)qstride;
            uint32_t k = 0;
            for (; k + 8 <= nq; k += 8) {                          // eight bytes in flight (byte accesses: see the note on the bases above)
                uint8_t t[8];
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) t[bb] = q[k + bb];
#pragma unroll
                for (int bb = 0; bb < 8; ++bb) dq[k + bb] = (uint8_t)(t[bb] + 33);
            }
            for (; k < nq; ++k) dq[k] = (uint8_t)(q[k] + 33);
        }
        if (row_loc) row_loc[r] = loc[i];
        (void)status;
    }
}

__global__ void thj_k_check_seen(const uint32_t* seen, uint32_t n, unsigned int* status) {
    // (one atomic per wave that has something to report, and none once it is reported: thousands of lanes on one address took 0.4 ms)
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        const bool miss = i < n && !seen[i];
        if (__ballot(miss) && (threadIdx.x & 63) == 0 && !__atomic_load_n(&status[ST_MISSING_READ], __ATOMIC_RELAXED)) atomicExch(&status[ST_MISSING_READ], 1u);
    }
}

}  // namespace ing

// ------------------------------------------------------------------------------------------------ host side of the ingest

namespace ing {

// a bump allocator over one device allocation per call site: the ingest of a shard needs two dozen arrays whose sizes are known
// up front or after one look at a counter; hipMalloc / hipFree per array would cost more than the kernels
struct Arena {
    char* base = nullptr; size_t cap = 0, used = 0;
    template <class T> T* take(size_t n) {
        const size_t bytes = (n * sizeof(T) + 255) & ~(size_t)255;
        if (used + bytes > cap) return nullptr;
        T* p = (T*)(base + used); used += bytes; return p;
    }
};

// BGZF member table of a piece of a BAM file: (payload offset, payload length) per member; stops at the EOF member / end of data
static bool member_table(const uint8_t* d, int64_t n, std::vector<thj_bgzf_block>& out, int64_t base_off) {
    int64_t off = 0;
    while (off + 28 <= n) {
        if (d[off] != 31 || d[off + 1] != 139 || d[off + 2] != 8 || !(d[off + 3] & 4)) return false;
        const uint32_t xlen = d[off + 10] | (d[off + 11] << 8);
        // the BC subfield holds BSIZE; samtools writes it first (XLEN 6), others may not
        int64_t x = off + 12; const int64_t xe = x + xlen; uint32_t bsize = 0;
        while (x + 4 <= xe) {
            const uint32_t slen = d[x + 2] | (d[x + 3] << 8);
            if (d[x] == 'B' && d[x + 1] == 'C' && slen == 2) bsize = (uint32_t)(d[x + 4] | (d[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        if (!bsize || off + bsize > n) return false;
        const uint32_t payload = bsize - (12 + xlen) - 8;
        const uint32_t isize = d[off + bsize - 4] | (d[off + bsize - 3] << 8) | (d[off + bsize - 2] << 16) | ((uint32_t)d[off + bsize - 1] << 24);
        if (isize) out.push_back({(uint64_t)(base_off + off + 12 + xlen), payload, isize});
        off += bsize;
    }
    return off == n;
}

}  // namespace ing

struct IngestOwned { thj_seg_batch desc; void* ptrs[6]; };          // same layout as the uploaded batches: thj_batch_free releases it
using IngestOwnedSpan = OwnedSpanBatch;     // ... thj_span_batch_free
__global__ void thj_k_rebase_u32(const uint32_t* __restrict__ in, uint32_t n, uint32_t base, uint32_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = in[i] - base;
}

namespace ing {

// What the front half of an ingest leaves on the device: the records the hit factory keeps (and the reads' locations), densely,
// file after file; fb[f] .. fb[f + 1] = file f's range.
struct Parsed {
    uint32_t* id = nullptr; Hit16* h16 = nullptr; Hit32* h32 = nullptr; uint32_t* loc = nullptr;
    std::vector<uint32_t> fb;
    std::vector<uint32_t> file_first_block, file_blocks;           // where each file's members sit in `infl` (64 KiB slots)
    uint8_t* infl = nullptr; unsigned int* status = nullptr;
    Arena a1;
    int64_t n = 0;
};

static int grow_scan_tmp(thj_ctx* c, size_t need) {
    if (need < ((size_t)1 << 20)) need = (size_t)1 << 20;        // (once: hipFree waits for the whole device)
    if (need > c->sort_tmp_bytes) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; HIPCHK(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
    return THJ_OK;
}
// exclusive prefix sum of n 32-bit counts (thj_scan.h: three small kernels, not hipcub::DeviceScan)
static int exclusive_sum(thj_ctx* c, const uint32_t* in, uint32_t* out, int64_t n) {
    if (n <= 0) return THJ_OK;
    int rc = grow_scan_tmp(c, thj_scan::scratch_bytes(n, 4));
    if (rc) return rc;
    thj_scan::exclusive_sum<uint32_t, uint32_t>(c->stream, in, out, n, c->d_sort_tmp);
    return THJ_OK;
}
static unsigned grid_for(int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

#define ING_TAKE(ar, var, T, n)                                                                                 \
    T* var = (ar).take<T>((size_t)(n));                                                                         \
    if (!var) { thj_set_error("thj_ingest: scratch arena too small (%zu of %zu bytes used)", (ar).used, (ar).cap); return THJ_ENOMEM; }

// inflate + walk + parse + compact for a list of pieces (kinds[f]: KIND_HITS / KIND_READS).  extra1 = bytes the caller will still
// take from the second arena.  Two synchronisations (record total, compact ranges).
// THJ_INGEST_TIMING=1: wall clock per phase of the ingest (a stream synchronisation at every mark, so only for diagnosis);
// thj_ingest_timing_report() prints the sums
struct PhaseClock {
    static std::atomic<long long>& slot(int k) { static std::atomic<long long> ns[12]; return ns[k]; }
    static bool on() { static const bool v = getenv("THJ_INGEST_TIMING") != nullptr; return v; }
    thj_ctx* c; long long t;
    explicit PhaseClock(thj_ctx* c_) : c(c_), t(0) { if (on()) { hipStreamSynchronize(c->stream); t = now(); } }
    static long long now() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void mark(int k) { if (!on()) return; hipStreamSynchronize(c->stream); const long long n = now(); slot(k) += n - t; t = n; }
};

static int ingest_front(thj_ctx* c, const thj_params* tp, const std::vector<const thj_bam_piece*>& pieces, const std::vector<uint32_t>& kinds, uint32_t begin_id,
                        uint32_t end_id, int want32, size_t extra1_per_rec, size_t extra1_fixed, Parsed& P) {
    const int nf = (int)pieces.size();
    std::vector<thj_bgzf_block> blocks;
    std::vector<uint8_t> blk_file;
    std::vector<FileInfo> files((size_t)nf);
    std::vector<uint32_t> tid2ref;
    int64_t comp_total = 0;
    for (int f = 0; f < nf; ++f) {
        const thj_bam_piece& p = *pieces[(size_t)f];
        FileInfo& fi = files[(size_t)f];
        fi.first_block = (uint32_t)blocks.size();
        if (p.comp_bytes > 0 && !member_table(p.comp, p.comp_bytes, blocks, comp_total)) { thj_set_error("thj_ingest: input %d is not a run of whole BGZF members", f); return THJ_EFALLBACK; }
        fi.n_blocks = (uint32_t)blocks.size() - fi.first_block;
        fi.first_skip = p.first_skip; fi.kind = kinds[(size_t)f];
        fi.tid_base = (uint32_t)tid2ref.size(); fi.n_tid = (uint32_t)p.n_tid; fi.rec_base = 0; fi.n_rec = 0;
        tid2ref.insert(tid2ref.end(), p.tid2ref, p.tid2ref + p.n_tid);
        blk_file.insert(blk_file.end(), fi.n_blocks, (uint8_t)f);
        comp_total += p.comp_bytes;
    }
    const int64_t nb = (int64_t)blocks.size();
    // a record's place is kept as member << 16 | offset in 32 bits (ParseOut::loc, thj_k_read_planes, the host's rinfl + loc): a shard of
    // more members than that takes the host readers -- the executables size their shards well below it (bytes of input per shard)
    if (nb > 65535) { thj_set_error("thj_ingest: %lld BGZF members in one shard (at most 65535)", (long long)nb); return THJ_EFALLBACK; }
    P.file_first_block.clear(); P.file_blocks.clear();
    for (int f = 0; f < nf; ++f) { P.file_first_block.push_back(files[(size_t)f].first_block); P.file_blocks.push_back(files[(size_t)f].n_blocks); }
    P.fb.assign((size_t)nf + 1, 0);
    P.n = 0;
    if (nb == 0) return THJ_OK;
    const size_t need0 = (size_t)comp_total + 64 + (size_t)nb * (65536 + sizeof(thj_bgzf_block) + 1 + 4 + 4 + 4 + (size_t)MAXREC * 2) + (size_t)nf * sizeof(FileInfo) +
                         tid2ref.size() * 4 + (1 << 20);
    // THJ_TRACE: where a context's first ingest spends its time (seconds since here), on stderr
    static const bool trace_env = getenv("THJ_TRACE") != nullptr;
    const bool trace_first = trace_env && c->ing_cap0 == 0;
    const long long tr0 = PhaseClock::now();
    auto lapse = [&](const char* what) { if (trace_first) { hipStreamSynchronize(c->stream); fprintf(stderr, "[trace] first ingest of a context: %-30s %.4f\n", what, (double)(PhaseClock::now() - tr0) * 1e-9); } };
    if (c->ing_cap0 < need0) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_ing0); c->d_ing0 = nullptr; c->ing_cap0 = 0; HIPCHK(hipMalloc(&c->d_ing0, need0 + need0 / 4)); c->ing_cap0 = need0 + need0 / 4; }
    Arena ar{(char*)c->d_ing0, c->ing_cap0, 0};
    ING_TAKE(ar, d_comp, uint8_t, comp_total + 64);
    ING_TAKE(ar, d_blocks, thj_bgzf_block, nb);
    ING_TAKE(ar, d_blk_file, uint8_t, nb);
    ING_TAKE(ar, d_files, FileInfo, nf);
    ING_TAKE(ar, d_tid, uint32_t, tid2ref.size() + 1);
    ING_TAKE(ar, d_infl, uint8_t, (size_t)nb << 16);
    ING_TAKE(ar, d_len, uint32_t, nb);
    ING_TAKE(ar, d_cnt, uint32_t, nb + 1);
    ING_TAKE(ar, d_base, uint32_t, nb + 1);
    ING_TAKE(ar, d_recoff, uint16_t, (size_t)nb * MAXREC);
    ING_TAKE(ar, d_status, unsigned int, 16);
    P.infl = d_infl; P.status = d_status;
    lapse("first arena");
    PhaseClock pc(c);
    {
        int64_t at = 0;
        for (int f = 0; f < nf; ++f) { const thj_bam_piece& p = *pieces[(size_t)f]; if (p.comp_bytes) HIPCHK(hipMemcpyAsync(d_comp + at, p.comp, (size_t)p.comp_bytes, hipMemcpyHostToDevice, c->stream)); at += p.comp_bytes; }
    }
    HIPCHK(hipMemcpyAsync(d_blocks, blocks.data(), (size_t)nb * sizeof(thj_bgzf_block), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_blk_file, blk_file.data(), (size_t)nb, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_files, files.data(), (size_t)nf * sizeof(FileInfo), hipMemcpyHostToDevice, c->stream));
    if (!tid2ref.empty()) HIPCHK(hipMemcpyAsync(d_tid, tid2ref.data(), tid2ref.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(d_status, 0, 64, c->stream));
    HIPCHK(hipMemsetAsync(d_cnt + nb, 0, 4, c->stream));
    pc.mark(0);
    lapse("compressed pieces on the device");
    { uint32_t mx = 0; for (const auto& bk : blocks) mx = bk.in_len > mx ? bk.in_len : mx;
      const int rc_ = launch_inflate(c, d_comp, d_blocks, nb, d_infl, d_len, mx); if (rc_) return rc_; }
    pc.mark(1);
    lapse("inflated (scratch, kernels)");
    hipLaunchKernelGGL(thj_k_walk, dim3((unsigned)nb), dim3(64), 0, c->stream, d_infl, d_len, d_blk_file, d_files, (int)nb, d_recoff, d_cnt, d_status);
    int rc = exclusive_sum(c, d_cnt, d_base, nb + 1);
    if (rc) return rc;
    std::vector<uint32_t> h_base((size_t)nb + 1);
    unsigned int h_status[16];
    HIPCHK(hipMemcpyAsync(h_base.data(), d_base, (size_t)(nb + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(h_status, d_status, 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    pc.mark(2);
    lapse("records walked");
    if (h_status[ST_CORRUPT]) { thj_set_error("thj_ingest: a BGZF member does not inflate (corrupt input)"); return THJ_EINVAL; }
    if (h_status[ST_STRADDLE]) { thj_set_error("BAM records straddle BGZF members (not written by samtools' bam_write1)"); return THJ_EFALLBACK; }
    const int64_t T = h_base[(size_t)nb];
    for (int f = 0; f < nf; ++f) { files[(size_t)f].rec_base = h_base[files[(size_t)f].first_block]; files[(size_t)f].n_rec = h_base[files[(size_t)f].first_block + files[(size_t)f].n_blocks] - files[(size_t)f].rec_base; }
    HIPCHK(hipMemcpyAsync(d_files, files.data(), (size_t)nf * sizeof(FileInfo), hipMemcpyHostToDevice, c->stream));
    if (T == 0) return THJ_OK;
    const size_t hit_b = want32 ? sizeof(Hit32) : sizeof(Hit16);
    const size_t need1 = (size_t)(T + 64) * (4 + 4 + 4 + 4 + hit_b + 4 + 4 + hit_b + 4 + extra1_per_rec) + extra1_fixed + (size_t)(nf + 64) * 1024 + (1 << 20);
    if (c->ing_cap1 < need1) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_ing1); c->d_ing1 = nullptr; c->ing_cap1 = 0; HIPCHK(hipMalloc(&c->d_ing1, need1 + need1 / 4)); c->ing_cap1 = need1 + need1 / 4; }
    lapse("second arena");
    P.a1 = Arena{(char*)c->d_ing1, c->ing_cap1, 0};
    ING_TAKE(P.a1, p_id, uint32_t, T); ING_TAKE(P.a1, p_valid, uint32_t, T + 1); ING_TAKE(P.a1, p_dst, uint32_t, T + 1); ING_TAKE(P.a1, p_isr, uint32_t, T);
    ING_TAKE(P.a1, p_hit, uint8_t, (size_t)T * hit_b); ING_TAKE(P.a1, p_loc, uint32_t, T);
    ING_TAKE(P.a1, q_id, uint32_t, T); ING_TAKE(P.a1, q_hit, uint8_t, (size_t)T * hit_b); ING_TAKE(P.a1, q_loc, uint32_t, T);
    ParseOut po{p_id, p_valid, want32 ? nullptr : (Hit16*)p_hit, want32 ? (Hit32*)p_hit : nullptr, p_loc};
    ParseOut qo{q_id, nullptr, want32 ? nullptr : (Hit16*)q_hit, want32 ? (Hit32*)q_hit : nullptr, q_loc};
    HIPCHK(hipMemsetAsync(p_valid + T, 0, 4, c->stream));
    hipLaunchKernelGGL(thj_k_parse, dim3((unsigned)nb), dim3(256), 0, c->stream, d_infl, d_blk_file, d_files, d_recoff, d_cnt, d_base, d_tid, begin_id, end_id,
                       (int)tp->max_report_intron, want32, po, d_status);
    pc.mark(3);
    lapse("parse kernel");
    hipLaunchKernelGGL(thj_k_mark_reads, dim3(grid_for(T)), dim3(256), 0, c->stream, d_files, nf, d_base, p_isr, T);
    lapse("mark");
    if ((rc = exclusive_sum(c, p_valid, p_dst, T + 1))) return rc;
    lapse("mark + scan");
    hipLaunchKernelGGL(thj_k_compact, dim3(grid_for(T)), dim3(256), 0, c->stream, T, p_valid, p_dst, po, qo, want32, p_isr);
    for (int f = 0; f < nf; ++f) HIPCHK(hipMemcpyAsync(&P.fb[(size_t)f], p_dst + files[(size_t)f].rec_base, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&P.fb[(size_t)nf], p_dst + T, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(h_status, d_status, 64, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    pc.mark(4);
    lapse("parsed and compacted");
    if (h_status[ST_CORRUPT]) { thj_set_error("thj_ingest: malformed BAM record (its header or a tag does not fit its block_size)"); return THJ_EINVAL; }
    if (h_status[ST_XF]) { thj_set_error("fusion (XF) alignments are not supported by this build"); return THJ_EINVAL; }
    if (h_status[ST_CIGAR]) { thj_set_error("a segment alignment has more than 5 CIGAR operations (this build supports 5)"); return THJ_EINVAL; }
    P.id = q_id; P.h16 = qo.h16; P.h32 = qo.h32; P.loc = q_loc; P.n = P.fb[(size_t)nf];
    return THJ_OK;
}

}  // namespace ing

extern "C" void thj_ingest_timing_report(void) {
    if (!ing::PhaseClock::on()) return;
    static const char* const nm[8] = {"host-to-device copies of the compressed pieces", "inflate kernel", "record walk + scan + round trip", "parse kernel",
                                      "compact + round trip", "merge by read id + scatter (seg batch)", "merge by read id + scatter (span batch)", "reads: planes + device-to-host"};
    for (int k = 0; k < 8; ++k) fprintf(stderr, "[ingest-seconds] %-52s %8.3f\n", nm[k], (double)ing::PhaseClock::slot(k).load() * 1e-9);
}

extern "C" int thj_ingest_seg_batch(thj_ctx* c, const thj_params* tp, int32_t nseg, const thj_bam_piece* segs, const thj_bam_piece* mate_full,
                                    const thj_bam_piece* mate_last, const thj_bam_piece* reads, uint32_t begin_id, uint32_t end_id, int32_t include_top0,
                                    uint32_t ordinal_base, thj_seg_batch** out, int64_t* n_reads_out) {
    using namespace ing;
    if (nseg > 8 && nseg <= 16) { thj_set_error("reads of more than eight segments"); return THJ_EFALLBACK; }       // the host readers take them (the kernels do up to 16)
    if (!c || !tp || nseg < 1 || nseg > 8 || !segs || !reads || !out) { thj_set_error("thj_ingest_seg_batch: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    *out = nullptr;
    if (n_reads_out) *n_reads_out = 0;
    // words per plane: a read of nseg segments is shorter than (nseg + 1) segment lengths (the last segment takes the remainder)
    int W = (tp->segment_length * (nseg + 1) - 1 + 63) / 64;
    if (W < 1) W = 1;
    if (W > 4) { thj_set_error("reads longer than 256 bases"); return THJ_EFALLBACK; }
    std::vector<const thj_bam_piece*> pieces;
    std::vector<uint32_t> kinds;
    for (int s = 0; s < nseg; ++s) { pieces.push_back(&segs[s]); kinds.push_back(KIND_HITS); }
    const int f_full = mate_full ? (int)pieces.size() : -1; if (mate_full) { pieces.push_back(mate_full); kinds.push_back(KIND_HITS); }
    const int f_last = mate_last ? (int)pieces.size() : -1; if (mate_last) { pieces.push_back(mate_last); kinds.push_back(KIND_HITS); }
    const int f_reads = (int)pieces.size(); pieces.push_back(reads); kinds.push_back(KIND_READS);
    Parsed P;
    // the merge takes, per id of the shard's id range, two words per map + two; per row a handful more
    int rc = ingest_front(c, tp, pieces, kinds, begin_id, end_id, 0, 0, 0, P);
    if (rc) return rc;
    PhaseClock pc(c);
    struct PcEnd { PhaseClock& p; ~PcEnd() { p.mark(5); } } pc_end{pc};
    const std::vector<uint32_t>& fb = P.fb;
    if (P.n == 0) return THJ_OK;
    uint32_t id_lo = 0xFFFFFFFFu, id_hi = 0;
    {
        std::vector<uint32_t> ends((size_t)nseg * 2, 0);
        for (int s = 0; s < nseg; ++s) if (fb[(size_t)s + 1] > fb[(size_t)s]) {
            HIPCHK(hipMemcpyAsync(&ends[(size_t)s * 2], P.id + fb[(size_t)s], 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipMemcpyAsync(&ends[(size_t)s * 2 + 1], P.id + fb[(size_t)s + 1] - 1, 4, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int s = 0; s < nseg; ++s) if (fb[(size_t)s + 1] > fb[(size_t)s]) { id_lo = std::min(id_lo, ends[(size_t)s * 2]); id_hi = std::max(id_hi, ends[(size_t)s * 2 + 1]); }
    }
    if (id_lo > id_hi) return THJ_OK;                            // no segment hit in range
    const uint32_t span = id_hi - id_lo + 1;
    // ---- merge by id (scratch of its own: sized by the id range)
    const int nmaps = nseg + (f_full >= 0 ? 1 : 0) + (f_last >= 0 ? 1 : 0);
    const size_t need2 = (size_t)span * 4 * (2 * (size_t)nmaps + 2 + (size_t)nseg + 4) + (1 << 20);
    void* d_merge = nullptr;
    { int rc_ = thj_dev_alloc(c, &d_merge, need2); if (rc_) return rc_; }
    struct Guard { thj_ctx* c; void* p; ~Guard() { hipStreamSynchronize(c->stream); thj_dev_release(c, p); } } guard{c, d_merge};
    Arena am{(char*)d_merge, need2, 0};
    ING_TAKE(am, m_first, uint32_t, (size_t)nmaps * span); ING_TAKE(am, m_cnt, uint32_t, (size_t)nmaps * span);
    ING_TAKE(am, m_vis, uint32_t, span + 1); ING_TAKE(am, m_row, uint32_t, span + 1);
    HIPCHK(hipMemsetAsync(m_cnt, 0, (size_t)nmaps * span * 4, c->stream));
    HIPCHK(hipMemsetAsync(m_vis + span, 0, 4, c->stream));
    for (int m = 0; m < nmaps; ++m) {
        const int f = m < nseg ? m : (m == nseg && f_full >= 0 ? f_full : f_last);
        const uint32_t a = fb[(size_t)f], b = fb[(size_t)f + 1];
        if (b > a) hipLaunchKernelGGL(thj_k_runs_clipped, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.id, a, b, id_lo, span, m_first + (size_t)m * span, m_cnt + (size_t)m * span);
    }
    hipLaunchKernelGGL(thj_k_visited, dim3(grid_for(span)), dim3(256), 0, c->stream, m_cnt, nseg, span, (int)include_top0, m_vis);
    if ((rc = exclusive_sum(c, m_vis, m_row, span + 1))) return rc;
    uint32_t n_rows = 0;
    HIPCHK(hipMemcpyAsync(&n_rows, m_row + span, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_rows == 0) return THJ_OK;
    // ---- the batch (device arrays of its own: it outlives the scratch)
    const bool have_mate = f_full >= 0 || f_last >= 0;
    IngestOwned* ob = new IngestOwned();
    memset(ob, 0, sizeof *ob);
    auto fail = [&](int code) { hipStreamSynchronize(c->stream); for (size_t i = 0; i < sizeof ob->ptrs / sizeof *ob->ptrs; ++i) thj_dev_release(c, ob->ptrs[i]); delete ob; return code; };
    uint32_t* b_off = nullptr; Hit16* b_hits = nullptr; u64* b_planes = nullptr; uint16_t* b_len = nullptr; uint32_t* b_moff = nullptr; Hit16* b_mh = nullptr;
    uint32_t* cell = am.take<uint32_t>((size_t)n_rows * nseg + 1); uint32_t* mcell = am.take<uint32_t>((size_t)n_rows + 1);
    uint32_t* row_id = am.take<uint32_t>(n_rows); uint32_t* seen = am.take<uint32_t>(n_rows);
    if (!cell || !mcell || !row_id || !seen) { thj_set_error("thj_ingest: merge scratch too small"); return fail(THJ_ENOMEM); }
#define ING_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { thj_set_error("%s: %s", #expr, hipGetErrorString(e__)); return fail(THJ_EHIP); } } while (0)
    { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, ((size_t)n_rows * nseg + 1) * 4)) return fail(THJ_EHIP); b_off = (decltype(b_off))v_; ob->ptrs[0] = b_off; }
    { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, (size_t)n_rows * 3 * W * 8)) return fail(THJ_EHIP); b_planes = (decltype(b_planes))v_; ob->ptrs[2] = b_planes; }
    { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, (size_t)n_rows * 2)) return fail(THJ_EHIP); b_len = (decltype(b_len))v_; ob->ptrs[3] = b_len; }
    ING_HIP(hipMemsetAsync(cell + (size_t)n_rows * nseg, 0, 4, c->stream));
    ING_HIP(hipMemsetAsync(mcell + n_rows, 0, 4, c->stream));
    ING_HIP(hipMemsetAsync(seen, 0, (size_t)n_rows * 4, c->stream));
    ING_HIP(hipMemsetAsync(b_len, 0, (size_t)n_rows * 2, c->stream));
    hipLaunchKernelGGL(thj_k_row_counts, dim3(grid_for(span)), dim3(256), 0, c->stream, m_vis, m_row, m_cnt, nseg, span, cell,
                       f_full >= 0 ? m_cnt + (size_t)nseg * span : (const uint32_t*)nullptr,
                       f_last >= 0 ? m_cnt + (size_t)(nseg + (f_full >= 0 ? 1 : 0)) * span : (const uint32_t*)nullptr, have_mate ? mcell : (uint32_t*)nullptr, row_id, id_lo);
    if ((rc = exclusive_sum(c, cell, b_off, (int64_t)n_rows * nseg + 1))) return fail(rc);
    uint32_t n_hits = 0, n_mh = 0;
    ING_HIP(hipMemcpyAsync(&n_hits, b_off + (size_t)n_rows * nseg, 4, hipMemcpyDeviceToHost, c->stream));
    if (have_mate) {
        { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, ((size_t)n_rows + 1) * 4)) return fail(THJ_EHIP); b_moff = (decltype(b_moff))v_; ob->ptrs[4] = b_moff; }
        if ((rc = exclusive_sum(c, mcell, b_moff, (int64_t)n_rows + 1))) return fail(rc);
        ING_HIP(hipMemcpyAsync(&n_mh, b_moff + n_rows, 4, hipMemcpyDeviceToHost, c->stream));
    }
    ING_HIP(hipStreamSynchronize(c->stream));
    { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, (size_t)(n_hits ? n_hits : 1) * 16)) return fail(THJ_EHIP); b_hits = (decltype(b_hits))v_; ob->ptrs[1] = b_hits; }
    for (int s = 0; s < nseg; ++s) {
        const uint32_t a = fb[(size_t)s], b = fb[(size_t)s + 1];
        if (b > a) hipLaunchKernelGGL(thj_k_scatter_hits<Hit16>, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.id, P.h16, a, b, id_lo, span, m_vis, m_row, m_first + (size_t)s * span, b_off, nseg, s, b_hits);
    }
    if (have_mate) {
        { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, (size_t)(n_mh ? n_mh : 1) * 16)) return fail(THJ_EHIP); b_mh = (decltype(b_mh))v_; ob->ptrs[5] = b_mh; }
        int m = nseg;
        const uint32_t* cf = f_full >= 0 ? m_cnt + (size_t)nseg * span : nullptr;
        if (f_full >= 0) {
            const uint32_t a = fb[(size_t)f_full], b = fb[(size_t)f_full + 1];
            if (b > a) hipLaunchKernelGGL(thj_k_scatter_mates, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.id, P.h16, a, b, id_lo, span, m_vis, m_row, m_first + (size_t)m * span, b_moff, cf, 0, b_mh);
            ++m;
        }
        if (f_last >= 0) {
            const uint32_t a = fb[(size_t)f_last], b = fb[(size_t)f_last + 1];
            if (b > a) hipLaunchKernelGGL(thj_k_scatter_mates, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.id, P.h16, a, b, id_lo, span, m_vis, m_row, m_first + (size_t)m * span, b_moff, cf, 1, b_mh);
        }
    }
    {
        const uint32_t a = fb[(size_t)f_reads], b = fb[(size_t)f_reads + 1];
        if (b > a) hipLaunchKernelGGL(thj_k_read_planes, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.infl, P.id, P.loc, a, b, id_lo, span, m_vis, m_row, W, b_planes, b_len, seen, P.status);
        hipLaunchKernelGGL(thj_k_check_seen, dim3(grid_for(n_rows)), dim3(256), 0, c->stream, seen, n_rows, P.status);
    }
    unsigned int h_status[16];
    ING_HIP(hipMemcpyAsync(h_status, P.status, 64, hipMemcpyDeviceToHost, c->stream));
    ING_HIP(hipStreamSynchronize(c->stream));
    ING_HIP(hipGetLastError());
    if (h_status[ST_MISSING_READ]) { thj_set_error("Error: could not get a read of the shard from the reads file"); return fail(THJ_EINVAL); }
    ob->desc.n_reads = (int32_t)n_rows; ob->desc.nseg = nseg; ob->desc.words_per_plane = W;
    ob->desc.seg_off = b_off; ob->desc.hits = (const thj_hit*)b_hits; ob->desc.read_planes = (const uint64_t*)b_planes; ob->desc.read_len = b_len;
    ob->desc.mate_off = b_moff; ob->desc.mate_hits = (const thj_hit*)b_mh; ob->desc.ordinal_base = ordinal_base;
    *out = &ob->desc;
    if (n_reads_out) *n_reads_out = n_rows;
    return THJ_OK;
}

// long_spanning_reads: the contig segment maps of one shard -> per (read, segment) CSR of thj_span_hit for the reads that have
// a hit in the first segment map (the groups JoinSegmentsWorker iterates over, long_spanning_reads.cpp:2706-2765); hits of a
// later segment whose read has none in the first are dropped, as look_right_for_hit_group never asks for them.  row_ids (host,
// n_rows entries, caller frees with free()) = the reads' ids in row order: the caller fetches these reads (it needs their
// names, bases and qualities for the BAM records anyway) and completes the batch with thj_span_batch_attach_reads.
// reads != nullptr: the shard's piece of the reads file rides along -- its records are inflated and located with the maps', the
// batch gets its read planes / lengths / quality strings on the device, and the inflated read records come back to the host
// (*reads_infl, malloc'd, member m at m << 16; row_loc[r] = location of row r's record in it) for the BAM output
static int span_ingest_impl(thj_ctx* c, const thj_params* tp, int32_t nseg, const thj_bam_piece* segs, const thj_bam_piece* reads, uint32_t begin_id,
                            uint32_t end_id, thj_span_batch** out, uint32_t** row_ids, int64_t* n_rows_out, uint8_t** reads_infl, int64_t* reads_infl_bytes,
                            uint32_t** row_loc_out) {
    using namespace ing;
    HIPCHK(hipSetDevice(c->device));
    *out = nullptr; *row_ids = nullptr; *n_rows_out = 0;
    if (reads_infl) { *reads_infl = nullptr; *reads_infl_bytes = 0; *row_loc_out = nullptr; }
    std::vector<const thj_bam_piece*> pieces;
    std::vector<uint32_t> kinds;
    for (int s = 0; s < nseg; ++s) { pieces.push_back(&segs[s]); kinds.push_back(KIND_HITS); }
    const int f_reads = reads ? nseg : -1;
    if (reads) { pieces.push_back(reads); kinds.push_back(KIND_READS); }
    Parsed P;
    int rc = ingest_front(c, tp, pieces, kinds, begin_id, end_id, 1, 0, 0, P);
    if (rc) return rc;
    PhaseClock pc(c);
    const std::vector<uint32_t>& fb = P.fb;
    if (P.n == 0 || fb[1] == fb[0]) return THJ_OK;               // no first-segment hit in range
    uint32_t ends[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(&ends[0], P.id + fb[0], 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&ends[1], P.id + fb[1] - 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t id_lo = ends[0], span = ends[1] - ends[0] + 1;
    const size_t need2 = (size_t)span * 4 * (2 * (size_t)nseg + 2 + (size_t)nseg + 8) + (1 << 20);
    void* d_merge = nullptr;
    { int rc_ = thj_dev_alloc(c, &d_merge, need2); if (rc_) return rc_; }
    struct Guard { thj_ctx* c; void* p; ~Guard() { hipStreamSynchronize(c->stream); thj_dev_release(c, p); } } guard{c, d_merge};
    Arena am{(char*)d_merge, need2, 0};
    ING_TAKE(am, m_first, uint32_t, (size_t)nseg * span); ING_TAKE(am, m_cnt, uint32_t, (size_t)nseg * span);
    ING_TAKE(am, m_vis, uint32_t, span + 1); ING_TAKE(am, m_row, uint32_t, span + 1);
    HIPCHK(hipMemsetAsync(m_cnt, 0, (size_t)nseg * span * 4, c->stream));
    HIPCHK(hipMemsetAsync(m_vis + span, 0, 4, c->stream));
    for (int s = 0; s < nseg; ++s) {
        const uint32_t a = fb[(size_t)s], b = fb[(size_t)s + 1];
        if (b > a) hipLaunchKernelGGL(thj_k_runs_clipped, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.id, a, b, id_lo, span, m_first + (size_t)s * span, m_cnt + (size_t)s * span);
    }
    // visited = has a hit in segment 0: thj_k_visited over that one map
    hipLaunchKernelGGL(thj_k_visited, dim3(grid_for(span)), dim3(256), 0, c->stream, m_cnt, 1, span, 1, m_vis);
    if ((rc = exclusive_sum(c, m_vis, m_row, span + 1))) return rc;
    uint32_t n_rows = 0;
    HIPCHK(hipMemcpyAsync(&n_rows, m_row + span, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_rows == 0) return THJ_OK;
    IngestOwnedSpan* ob = new IngestOwnedSpan();
    memset(ob, 0, sizeof *ob);
    auto fail = [&](int code) { hipStreamSynchronize(c->stream); for (size_t i = 0; i < sizeof ob->ptrs / sizeof *ob->ptrs; ++i) thj_dev_release(c, ob->ptrs[i]); delete ob; return code; };
    uint32_t* cell = am.take<uint32_t>((size_t)n_rows * nseg + 1); uint32_t* row_id = am.take<uint32_t>(n_rows);
    if (!cell || !row_id) { thj_set_error("thj_ingest: merge scratch too small"); return fail(THJ_ENOMEM); }
    uint32_t* b_off = nullptr; Hit32* b_hits = nullptr;
    { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, ((size_t)n_rows * nseg + 1) * 4)) return fail(THJ_EHIP); b_off = (decltype(b_off))v_; ob->ptrs[0] = b_off; }
    ING_HIP(hipMemsetAsync(cell + (size_t)n_rows * nseg, 0, 4, c->stream));
    hipLaunchKernelGGL(thj_k_row_counts, dim3(grid_for(span)), dim3(256), 0, c->stream, m_vis, m_row, m_cnt, nseg, span, cell, (const uint32_t*)nullptr, (const uint32_t*)nullptr,
                       (uint32_t*)nullptr, row_id, id_lo);
    if ((rc = exclusive_sum(c, cell, b_off, (int64_t)n_rows * nseg + 1))) return fail(rc);
    uint32_t n_hits = 0;
    ING_HIP(hipMemcpyAsync(&n_hits, b_off + (size_t)n_rows * nseg, 4, hipMemcpyDeviceToHost, c->stream));
    uint32_t* h_ids = (uint32_t*)malloc((size_t)n_rows * 4);
    if (!h_ids) return fail(THJ_ENOMEM);
    ING_HIP(hipMemcpyAsync(h_ids, row_id, (size_t)n_rows * 4, hipMemcpyDeviceToHost, c->stream));
    ING_HIP(hipStreamSynchronize(c->stream));
    { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, (size_t)(n_hits ? n_hits : 1) * 32)) return fail(THJ_EHIP); b_hits = (decltype(b_hits))v_; ob->ptrs[1] = b_hits; }
    uint4* b_heads = nullptr;
    { void* v_ = nullptr; if (thj_dev_alloc(c, &v_, (size_t)(n_hits ? n_hits : 1) * 16)) return fail(THJ_EHIP); b_heads = (uint4*)v_; ob->ptrs[5] = b_heads; }
    for (int s = 0; s < nseg; ++s) {
        const uint32_t a = fb[(size_t)s], b = fb[(size_t)s + 1];
        if (b > a) hipLaunchKernelGGL(thj_k_scatter_span, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.id, P.h32, a, b, id_lo, span, m_vis, m_row, m_first + (size_t)s * span, b_off, nseg, s, b_hits, b_heads);
    }
    pc.mark(6);
    uint32_t* h_loc = nullptr; uint8_t* h_infl = nullptr;
    auto fail2 = [&](int code) { free(h_ids); free(h_loc); thj_pinned_free(h_infl); return fail(code); };
    if (reads) {
        // the reads of the rows: planes / lengths / quality strings straight from the BAM records, where the kernels will read them
        int W = (tp->segment_length * (nseg + 1) - 1 + 63) / 64;
        if (W < 1) W = 1;
        if (W > 4) { thj_set_error("reads longer than 256 bases"); return fail2(THJ_EFALLBACK); }
        const int qstride = (tp->segment_length * (nseg + 1) + 3) / 4 * 4;
        uint32_t* seen = am.take<uint32_t>(n_rows); uint32_t* d_loc = am.take<uint32_t>(n_rows);
        if (!seen || !d_loc) { thj_set_error("thj_ingest: merge scratch too small"); return fail2(THJ_ENOMEM); }
        if (thj_dev_alloc(c, &ob->ptrs[2], (size_t)n_rows * 3 * W * 8) || thj_dev_alloc(c, &ob->ptrs[3], (size_t)n_rows * 2) ||
            thj_dev_alloc(c, &ob->ptrs[4], (size_t)n_rows * qstride)) return fail2(THJ_EHIP);
        ING_HIP(hipMemsetAsync(seen, 0, (size_t)n_rows * 4, c->stream));
        ING_HIP(hipMemsetAsync(ob->ptrs[3], 0, (size_t)n_rows * 2, c->stream));
        ING_HIP(hipMemsetAsync(ob->ptrs[4], 0, (size_t)n_rows * qstride, c->stream));
        const uint32_t a = fb[(size_t)f_reads], b = fb[(size_t)f_reads + 1];
        static const int planes_wide = getenv("THJ_PLANES_LOAD64") ? atoi(getenv("THJ_PLANES_LOAD64")) : 0;
        static const bool ingest_sync = getenv("THJ_INGEST_SYNC") != nullptr;          // developer switch: the device idle before the reads' planes are made
        if (ingest_sync) ING_HIP(hipDeviceSynchronize());
        if (b > a) hipLaunchKernelGGL(thj_k_read_planes, dim3(grid_for(b - a)), dim3(256), 0, c->stream, P.infl, P.id, P.loc, a, b, id_lo, span, m_vis, m_row, W,
                                      (u64*)ob->ptrs[2], (uint16_t*)ob->ptrs[3], seen, P.status, (uint8_t*)ob->ptrs[4], qstride, d_loc, planes_wide);
        hipLaunchKernelGGL(thj_k_check_seen, dim3(grid_for(n_rows)), dim3(256), 0, c->stream, seen, n_rows, P.status);
        // the rows' own BAM records stay on the device with the batch (thj_span_bam_encode copies names, bases and qualities from
        // them); the host copy is made only for a caller that asks for it
        const size_t ib = (size_t)P.file_blocks[(size_t)f_reads] << 16;
        const uint32_t base = P.file_first_block[(size_t)f_reads] << 16;
        if (thj_dev_alloc(c, &ob->ptrs[6], ib ? ib : 16) || thj_dev_alloc(c, &ob->ptrs[7], (size_t)n_rows * 4)) return fail2(THJ_EHIP);
        ob->reads_infl_bytes = ib;
        if (ib) ING_HIP(hipMemcpyAsync(ob->ptrs[6], P.infl + (size_t)base, ib, hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(thj_k_rebase_u32, dim3(grid_for(n_rows)), dim3(256), 0, c->stream, (const uint32_t*)d_loc, n_rows, base, (uint32_t*)ob->ptrs[7]);
        unsigned int h_status[16];
        ING_HIP(hipMemcpyAsync(h_status, P.status, 64, hipMemcpyDeviceToHost, c->stream));
        if (reads_infl) {
            h_loc = (uint32_t*)malloc((size_t)n_rows * 4);
            h_infl = (uint8_t*)thj_pinned_alloc(ib ? ib : 16);
            if (!h_loc || !h_infl) return fail2(THJ_ENOMEM);
            ING_HIP(hipMemcpyAsync(h_loc, ob->ptrs[7], (size_t)n_rows * 4, hipMemcpyDeviceToHost, c->stream));
            if (ib) ING_HIP(hipMemcpyAsync(h_infl, ob->ptrs[6], ib, hipMemcpyDeviceToHost, c->stream));
        }
        ING_HIP(hipStreamSynchronize(c->stream));
        if (h_status[ST_MISSING_READ]) { thj_set_error("Error: could not get a read of the shard from the reads file"); return fail2(THJ_EINVAL); }
        ob->desc.words_per_plane = W; ob->desc.qual_stride = qstride;
        ob->desc.read_planes = (const uint64_t*)ob->ptrs[2]; ob->desc.read_len = (const uint16_t*)ob->ptrs[3]; ob->desc.quals = (const uint8_t*)ob->ptrs[4];
        if (reads_infl) { *reads_infl = h_infl; *reads_infl_bytes = (int64_t)ib; *row_loc_out = h_loc; }
        pc.mark(7);
    }
    ING_HIP(hipStreamSynchronize(c->stream));
    ING_HIP(hipGetLastError());
    ob->desc.n_reads = (int32_t)n_rows; ob->desc.nseg = nseg;
    ob->desc.seg_off = b_off; ob->desc.hits = (const thj_span_hit*)b_hits; ob->desc.hit_heads = b_heads;
    *out = &ob->desc; *row_ids = h_ids; *n_rows_out = n_rows;
    return THJ_OK;
}

// ---- the initially unmapped reads (--ium-reads) of the coverage / butterfly search from an unaligned BAM: every record of the piece is a
// read (ReadStream::get_direct, reads.cpp:600-630; index_read_mers segment_juncs.cpp:548-571 looks at a read's first 32 bases), its SEQ
// nibbles become the bit planes thj_covsearch_add_reads takes -- one word per plane: the first 64 bases -- without leaving the device
namespace ing {
__global__ __launch_bounds__(256) void thj_k_ium_planes(const uint8_t* __restrict__ infl, const uint32_t* __restrict__ loc, uint32_t n, u64* __restrict__ planes, uint16_t* __restrict__ rlen) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint8_t* d = infl + ((size_t)(loc[i] >> 16) << 16) + (loc[i] & 0xFFFFu) + 4;
        const uint32_t l_rn = ld32(d + 8) & 0xFF, flag_nc = ld32(d + 12), n_cig = flag_nc & 0xFFFF, l_seq = ld32(d + 16);
        const uint8_t* sq = d + 32 + l_rn + 4 * n_cig;
        // a QC-failed record (BAM_FQCFAIL) is not a read (ReadStream::get_direct, reads.cpp:556): length 0, which the extension table ignores
        const bool qc_fail = ((flag_nc >> 16) & 0x200u) != 0;
        const uint32_t L = qc_fail ? 0u : (l_seq > 64u ? 64u : l_seq);
        u64 lo = 0, hi = 0, nn = 0;
        for (uint32_t k = 0; k < L; ++k) {
            const uint32_t nib = (sq[k >> 1] >> ((k & 1) ? 0 : 4)) & 0xF;
            const u64 b0 = (nib == 2u) | (nib == 8u), b1 = (nib == 4u) | (nib == 8u), isn = !((nib == 1u) | (nib == 2u) | (nib == 4u) | (nib == 8u));
            lo |= b0 << k; hi |= b1 << k; nn |= isn << k;
        }
        planes[(size_t)i * 3] = lo; planes[(size_t)i * 3 + 1] = hi; planes[(size_t)i * 3 + 2] = nn;
        rlen[i] = qc_fail ? (uint16_t)0 : (uint16_t)(l_seq > 0xFFFFu ? 0xFFFFu : l_seq);
    }
}
}  // namespace ing

extern "C" int thj_covsearch_add_reads(thj_ctx* c, int64_t n_reads, int32_t words_per_plane, const uint64_t* planes, const uint16_t* lens, int32_t on_device);
extern "C" int thj_covsearch_add_reads_bam(thj_ctx* c, const thj_bam_piece* reads, int64_t* n_reads_out) {
    using namespace ing;
    if (!c || !reads) { thj_set_error("thj_covsearch_add_reads_bam: null argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (n_reads_out) *n_reads_out = 0;
    thj_params p; thj_params_default(&p);
    std::vector<const thj_bam_piece*> pieces{reads};
    std::vector<uint32_t> kinds{KIND_READS};
    Parsed P;
    int rc = ingest_front(c, &p, pieces, kinds, 0u, 0xFFFFFFFFu, 0, 3 * 8 + 2, 1024, P);
    if (rc) return rc;
    const int64_t n = P.n;
    if (n == 0) return THJ_OK;
    u64* d_planes = P.a1.take<u64>((size_t)n * 3);
    uint16_t* d_len = P.a1.take<uint16_t>((size_t)n);
    if (!d_planes || !d_len) { thj_set_error("thj_covsearch_add_reads_bam: scratch too small"); return THJ_ENOMEM; }
    hipLaunchKernelGGL(thj_k_ium_planes, dim3(grid_for(n)), dim3(256), 0, c->stream, P.infl, P.loc, (uint32_t)n, d_planes, d_len);
    HIPCHK(hipGetLastError());
    if ((rc = thj_covsearch_add_reads(c, n, 1, (const uint64_t*)d_planes, d_len, 1))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));          // the scratch is the context's: the next ingest call writes over it
    if (n_reads_out) *n_reads_out = n;
    return THJ_OK;
}

// ---- page-locked host buffers, pooled for the process.  Copies to and from pageable memory go through the runtime's staging
// buffers on one thread (measured in long_spanning_reads with its CPUs busy encoding: 2.9 GB/s up, 6 GB/s down, half of a
// shard's time under the GPU's lock); from page-locked memory they are plain DMA.  Locking pages costs ~0.2 ms per MB, so
// buffers are kept and handed out again.
namespace {
struct PinnedPool {
    struct Blk { void* p; size_t cap; bool used; bool pinned; uint64_t stamp; };
    std::mutex mu; std::vector<Blk> blks; size_t bytes = 0; uint64_t clock = 0;
    static constexpr size_t PIN_LIMIT = (size_t)24 << 30, KEEP_LIMIT = (size_t)32 << 30;
    static void release(const Blk& b) { if (b.pinned) (void)hipHostFree(b.p); else free(b.p); }
    void* get(size_t n) {
        if (n < 4096) n = 4096;
        const size_t cap = n + n / 8;
        std::vector<Blk> drop;
        bool pin;
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (size_t i = 0; i < blks.size(); ++i)
                if (!blks[i].used && blks[i].cap >= n && blks[i].cap <= 2 * n + (1 << 20) && (best < 0 || blks[i].cap < blks[(size_t)best].cap)) best = (int)i;
            if (best >= 0) { blks[(size_t)best].used = true; blks[(size_t)best].stamp = ++clock; return blks[(size_t)best].p; }
            // nothing fits: before growing past the limits, let go of the idle blocks that were used longest ago (shards of varying
            // size would otherwise leave a trail of blocks nobody asks for again)
            while (bytes + cap > KEEP_LIMIT) {
                int old = -1;
                for (size_t i = 0; i < blks.size(); ++i) if (!blks[i].used && (old < 0 || blks[i].stamp < blks[(size_t)old].stamp)) old = (int)i;
                if (old < 0) break;
                drop.push_back(blks[(size_t)old]); bytes -= blks[(size_t)old].cap;
                blks.erase(blks.begin() + old);
            }
            size_t pinned_bytes = 0; for (auto& b : blks) if (b.pinned) pinned_bytes += b.cap;
            pin = pinned_bytes + cap <= PIN_LIMIT;
            bytes += cap;                                     // reserved under the lock; the allocation itself runs outside it
        }
        for (auto& b : drop) release(b);
        void* p = nullptr; bool pinned = pin;
        if (!pin || hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess || !p) { (void)hipGetLastError(); p = malloc(cap); pinned = false; }
        std::lock_guard<std::mutex> lk(mu);
        if (!p) { bytes -= cap; return nullptr; }
        blks.push_back({p, cap, true, pinned, ++clock});
        return p;
    }
    bool draining = false;
    void put(void* p) {
        if (!p) return;
        Blk gone{nullptr, 0, false, false, 0};
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < blks.size(); ++i) if (blks[i].p == p) {
                if (!draining) { blks[i].used = false; return; }
                gone = blks[i]; bytes -= gone.cap; blks.erase(blks.begin() + (ptrdiff_t)i);
                break;
            }
        }
        if (gone.p) release(gone);                         // outside the lock: unlocking pages takes its time
    }
    void drain() {
        std::vector<Blk> idle;
        {
            std::lock_guard<std::mutex> lk(mu);
            draining = true;
            for (size_t i = 0; i < blks.size();) {
                if (!blks[i].used) { idle.push_back(blks[i]); bytes -= blks[i].cap; blks.erase(blks.begin() + (ptrdiff_t)i); } else ++i;
            }
        }
        for (auto& b : idle) release(b);
    }
};
PinnedPool& pinned_pool() { static PinnedPool* pp = new PinnedPool(); return *pp; }      // never destroyed: the process leaves with _exit
}  // namespace
extern "C" void* thj_pinned_alloc(size_t bytes) { return pinned_pool().get(bytes); }
extern "C" void thj_pinned_free(void* p) { pinned_pool().put(p); }
extern "C" void thj_pinned_drain(void) { pinned_pool().drain(); }

extern "C" int thj_ingest_span_hits(thj_ctx* c, const thj_params* tp, int32_t nseg, const thj_bam_piece* segs, uint32_t begin_id, uint32_t end_id,
                                    thj_span_batch** out, uint32_t** row_ids, int64_t* n_rows_out) {
    if (nseg > 8 && nseg <= 16) { thj_set_error("reads of more than eight segments"); return THJ_EFALLBACK; }
    if (!c || !tp || nseg < 1 || nseg > 8 || !segs || !out || !row_ids || !n_rows_out) { thj_set_error("thj_ingest_span_hits: bad argument"); return THJ_EINVAL; }
    return span_ingest_impl(c, tp, nseg, segs, nullptr, begin_id, end_id, out, row_ids, n_rows_out, nullptr, nullptr, nullptr);
}
extern "C" int thj_ingest_span_batch(thj_ctx* c, const thj_params* tp, int32_t nseg, const thj_bam_piece* segs, const thj_bam_piece* reads, uint32_t begin_id,
                                     uint32_t end_id, thj_span_batch** out, uint32_t** row_ids, int64_t* n_rows_out, uint8_t** reads_infl,
                                     int64_t* reads_infl_bytes, uint32_t** row_loc) {
    const int host_copy = (reads_infl != nullptr) + (reads_infl_bytes != nullptr) + (row_loc != nullptr);       // all three or none
    if (nseg > 8 && nseg <= 16) { thj_set_error("reads of more than eight segments"); return THJ_EFALLBACK; }
    if (!c || !tp || nseg < 1 || nseg > 8 || !segs || !reads || !out || !row_ids || !n_rows_out || (host_copy != 0 && host_copy != 3)) {
        thj_set_error("thj_ingest_span_batch: bad argument"); return THJ_EINVAL;
    }
    return span_ingest_impl(c, tp, nseg, segs, reads, begin_id, end_id, out, row_ids, n_rows_out, reads_infl, reads_infl_bytes, row_loc);
}

// the host copy of a batch's read records, for a caller that let thj_ingest_span_batch keep them on the device and needs them after all
extern "C" int thj_span_batch_reads_host(thj_ctx* c, const thj_span_batch* batch, uint8_t** reads_infl, int64_t* reads_infl_bytes, uint32_t** row_loc) {
    if (!c || !batch || !reads_infl || !reads_infl_bytes || !row_loc) { thj_set_error("thj_span_batch_reads_host: bad argument"); return THJ_EINVAL; }
    const OwnedSpanBatch* ob = (const OwnedSpanBatch*)batch;
    if (!ob->ptrs[6] || !ob->ptrs[7]) { thj_set_error("thj_span_batch_reads_host: the batch holds no read records"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    const size_t ib = ob->reads_infl_bytes, n = (size_t)batch->n_reads;
    uint32_t* h_loc = (uint32_t*)malloc((n ? n : 1) * 4);
    uint8_t* h_infl = (uint8_t*)thj_pinned_alloc(ib ? ib : 16);
    if (!h_loc || !h_infl) { free(h_loc); thj_pinned_free(h_infl); thj_set_error("out of memory"); return THJ_ENOMEM; }
    if (n) HIPCHK(hipMemcpyAsync(h_loc, ob->ptrs[7], n * 4, hipMemcpyDeviceToHost, c->stream));
    if (ib) HIPCHK(hipMemcpyAsync(h_infl, ob->ptrs[6], ib, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *reads_infl = h_infl; *reads_infl_bytes = (int64_t)ib; *row_loc = h_loc;
    return THJ_OK;
}

// the reads of a batch made by thj_ingest_span_hits, in row order (HOST arrays as thj_reads_pack / thj_span_batch describe them)
extern "C" int thj_span_batch_attach_reads(thj_ctx* c, thj_span_batch* batch, int32_t words_per_plane, int32_t qual_stride, const uint64_t* planes,
                                           const uint16_t* lens, const uint8_t* quals) {
    if (!c || !batch || !planes || !lens || !quals || words_per_plane < 1 || words_per_plane > 8 || qual_stride < 0) { thj_set_error("thj_span_batch_attach_reads: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    IngestOwnedSpan* ob = (IngestOwnedSpan*)batch;
    const size_t n = (size_t)batch->n_reads;
    const size_t sizes[3] = {n * 3 * (size_t)words_per_plane * 8, n * 2, n * (size_t)qual_stride};
    const void* src[3] = {planes, lens, quals};
    for (int i = 0; i < 3; ++i) {
        thj_dev_release(c, ob->ptrs[2 + i]); ob->ptrs[2 + i] = nullptr;
        { int rc_ = thj_dev_alloc(c, &ob->ptrs[2 + i], sizes[i] ? sizes[i] : 16); if (rc_) return rc_; }
        if (sizes[i]) HIPCHK(hipMemcpyAsync(ob->ptrs[2 + i], src[i], sizes[i], hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    batch->words_per_plane = words_per_plane; batch->qual_stride = qual_stride;
    batch->read_planes = (const uint64_t*)ob->ptrs[2]; batch->read_len = (const uint16_t*)ob->ptrs[3]; batch->quals = (const uint8_t*)ob->ptrs[4];
    return THJ_OK;
}

// the runtime loads a translation unit's code object at its first launch (tens of milliseconds): thj_ctx_warm makes that happen early
__global__ void thj_k_warm_ingest(int* p) { if (p) *p = 0; }
void thj_warm_ingest(hipStream_t s) { hipLaunchKernelGGL(thj_k_warm_ingest, dim3(1), dim3(64), 0, s, (int*)nullptr); }
