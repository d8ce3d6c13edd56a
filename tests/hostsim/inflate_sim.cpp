// inflate_sim.cpp -- TEST-ONLY: compiles tophat_amd/csrc/thj_inflate_core.h (the lane logic of thj_k_huff) for the CPU, one lane at a
// time, and restates thj_k_lz's batch algorithm (64 tokens, prefix sum, rounds behind a high-water mark, sliding 40 KiB buffer)
// with explicit lane loops, so that both halves of the device inflater can be checked against zlib without a GPU.
// Never linked into libthj_hip.so.
#include "../../tophat_amd/csrc/thj_inflate_core.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "simt.h"

using namespace inf2;

struct WaveCpu { bool any(bool p) const { return p; } };
// the wave the table builders of thj_k_huffp are written against, over simt.h's fibers
struct WaveSimT {
    simt::Block* b; int lane;
    uint64_t ballot(bool p) { const uint32_t* a = b->exchange(lane, p ? 1u : 0u); uint64_t m = 0; for (int i = 0; i < 64; ++i) m |= (uint64_t)(a[i] & 1u) << i; return m; }
    void sync() { b->exchange(lane, 0); }
};
// the block header as thj_k_huffp takes it: lane 0 reads the code lengths, the wave builds the tables -- which must come out byte for
// byte as the one-lane builders make them.  Returns false on a table mismatch (a test failure, not a fallback).
static uint32_t g_slot_cap = SLOT_TOKENS;          // tokens of a lane that fit its slot (tests make it small: the pass of the lanes whose tokens do not fit)
extern "C" void inflate_sim_set_slot(uint32_t n) { g_slot_cap = n && n < SLOT_TOKENS ? n : SLOT_TOKENS; }
static uint32_t g_hdr_end = 0;          // where the header read from the staged words ended
static bool header_by_wave(Lane& H, std::vector<uint8_t>& lds, const uint32_t* w, uint32_t hpos, uint32_t limit) {
    // the code lengths: read from the staged words as thj_k_huffp's lane 0 does, and through the Lane's ring as the lane-per-member
    // kernel does -- same lengths, same verdict, same place in the stream
    HeaderInfo hi;
    {
        std::vector<uint8_t> l2 = lds;
        Lane R = H;
        R.lit = (uint16_t*)l2.data(); R.A = l2.data() + OFF_A; R.B = l2.data() + OFF_B; R.C = (uint16_t*)(l2.data() + OFF_C); R.ring = (uint32_t*)(l2.data() + OFF_RING); R.stage = (uint32_t*)(l2.data() + OFF_STAGE);
        lane_seek(R, hpos); R.state = ST_HEADER;
        const HeaderInfo hr = parse_header_lengths(R, WaveCpu{});
        const bool fb_r = R.state == ST_FALLBACK || overrun(R);
        const HeaderW hw = parse_header_lengths_w(w, hpos, limit, H.lit, H.A, H.B, H.C);
        hi = hw.hi;
        const bool trace = getenv("THJ_SIM_TRACE") != nullptr;
        if (hw.fallback != fb_r && !(hw.fallback && !hi.ok)) { if (trace) fprintf(stderr, "header: fallback %d, through the ring %d\n", (int)hw.fallback, (int)fb_r); return false; }
        if (!hw.fallback) {
            if (hi.hlit != hr.hlit || hi.hdist != hr.hdist || hi.build != hr.build || hi.ok != hr.ok) { if (trace) fprintf(stderr, "header: info differs\n"); return false; }
            if (hi.build && (memcmp(H.A, R.A, (size_t)(hi.hlit + hi.hdist)) || hw.last != R.last || hw.end_bit != lane_bitpos(R))) { if (trace) fprintf(stderr, "header: lengths / end differ (%u vs %u)\n", hw.end_bit, lane_bitpos(R)); return false; }
        }
        H.last = hw.last;
        H.state = hw.fallback ? ST_FALLBACK : ST_HEADER;
        g_hdr_end = hw.end_bit;
    }
    // the one-lane tables from the same lengths, on a copy of the LDS slice
    std::vector<uint8_t> ref = lds;
    Lane R = H;
    R.lit = (uint16_t*)ref.data(); R.A = ref.data() + OFF_A; R.B = ref.data() + OFF_B; R.C = (uint16_t*)(ref.data() + OFF_C);
    bool ok_ref = hi.ok;
    ok_ref = build_lit(R, R.A, hi.hlit, hi.build, WaveCpu{}) && ok_ref;
    ok_ref = build_dist(R, R.A + hi.hlit, hi.hdist, hi.build, WaveCpu{}) && ok_ref;
    bool ok_lane[64];
    simt::run_block(64, [&](simt::Block& blk, int tid) {
        WaveSimT x{&blk, tid};
        bool ok = hi.ok;
        ok = build_lit_wave(H.lit, H.C, H.A, hi.hlit, hi.build, x) && ok;
        ok = build_dist_wave(H.A, H.B, H.A + hi.hlit, hi.hdist, hi.build, x) && ok;
        ok_lane[tid] = ok;
    });
    for (int l = 1; l < 64; ++l) if (ok_lane[l] != ok_lane[0]) { if (getenv("THJ_SIM_TRACE")) fprintf(stderr, "ok differs between lanes\n"); return false; }
    const bool ok = ok_lane[0];
    if (ok != ok_ref) { if (getenv("THJ_SIM_TRACE")) fprintf(stderr, "ok %d ref %d build %d hlit %d hdist %d\n", (int)ok, (int)ok_ref, (int)hi.build, hi.hlit, hi.hdist); return false; }
    if (hi.build && ok) {
        if (memcmp(H.lit, R.lit, LIT_ENTRIES * 2) || memcmp(H.A, R.A, DROOT_SIZE)) {
            if (getenv("THJ_SIM_TRACE")) {
                for (int i = 0; i < LIT_ENTRIES; ++i) if (H.lit[i] != R.lit[i]) { fprintf(stderr, "lit[%d] = %04x, one lane: %04x\n", i, H.lit[i], R.lit[i]); break; }
                for (int i = 0; i < DROOT_SIZE; ++i) if (H.A[i] != R.A[i]) { fprintf(stderr, "A[%d] = %02x, one lane: %02x\n", i, H.A[i], R.A[i]); break; }
            }
            return false;
        }
        // B: the per-length words and the symbol list of the long distance codes (only as many symbols as there are long codes)
        const uint32_t *bw = (const uint32_t*)H.B, *br = (const uint32_t*)R.B;
        uint32_t nlong = 0;
        for (int q = 0; q < 7; ++q) { if (bw[q] != br[q]) { if (getenv("THJ_SIM_TRACE")) fprintf(stderr, "B word %d = %08x, one lane: %08x\n", q, bw[q], br[q]); return false; } nlong += (br[q] >> 15) & 63u; }
        if (memcmp(H.B + 28, R.B + 28, nlong)) return false;
    }
    if (H.state == ST_HEADER) H.state = (hi.build && ok) ? ST_DECODE : ST_FALLBACK;
    return true;
}

// one member through the lane logic.  comp / in_len: the raw DEFLATE stream; skew (0..15): how far into a 16-byte granule it starts.
// Returns 0 and fills tokens / ntok / outp, or 1 when the lane hands the member to the fallback.
extern "C" int inflate_sim_huff(const uint8_t* comp, uint32_t in_len, uint32_t skew, uint32_t* tokens, uint32_t* ntok, uint32_t* outp) {
    std::vector<uint8_t> lds((size_t)STRIDE_WORDS * 4, 0xA5);                    // stale LDS
    std::vector<uint8_t> stream((size_t)skew + in_len + 64, 0x5A);                // bytes around the member are someone else's
    memcpy(stream.data() + skew, comp, in_len);
    Lane L;
    memset(&L, 0, sizeof L);
    L.lit = (uint16_t*)lds.data(); L.A = lds.data() + OFF_A; L.B = lds.data() + OFF_B; L.C = (uint16_t*)(lds.data() + OFF_C); L.ring = (uint32_t*)(lds.data() + OFF_RING); L.stage = (uint32_t*)(lds.data() + OFF_STAGE);
    L.src = stream.data(); L.total = skew + in_len; L.rd = 4; L.tok = tokens;
    run_member(L, true, skew, WaveCpu{});
    const bool good = L.state == ST_DONE && !overrun(L);
    *ntok = good ? L.ntok : NTOK_FALLBACK; *outp = good ? L.outp : 0;
    return good ? 0 : 1;
}

// thj_k_huffp restated around the shared lane code: lane 0's header parse (Lane / parse_header), then the 64 lanes' segment passes
// (decode_segment) with the wave's shuffles and votes as loops over a lane array.  passes_out: segment passes run (statistics).
extern "C" int inflate_sim_huffp(const uint8_t* comp, uint32_t in_len, uint32_t skew, uint32_t* tokens, uint32_t* ntok, uint32_t* outp, int64_t* passes_out) {
    std::vector<uint8_t> lds((size_t)STRIDE_WORDS * 4, 0xA5);
    std::vector<uint8_t> stream((size_t)skew + in_len + 64 + 16, 0x5A);
    uint8_t* st0 = stream.data() + ((16 - ((uintptr_t)stream.data() & 15)) & 15);      // 16-byte aligned like Lane::src on the device
    memcpy(st0 + skew, comp, in_len);
    Lane H;
    memset(&H, 0, sizeof H);
    H.lit = (uint16_t*)lds.data(); H.A = lds.data() + OFF_A; H.B = lds.data() + OFF_B; H.C = (uint16_t*)(lds.data() + OFF_C); H.ring = (uint32_t*)(lds.data() + OFF_RING); H.stage = (uint32_t*)(lds.data() + OFF_STAGE);
    H.src = st0; H.total = skew + in_len; H.tok = tokens;
    const uint32_t* w = (const uint32_t*)st0;
    const uint32_t limit = H.total * 8;
    uint32_t tok_base = 0, out_base = 0, hpos = skew * 8;
    int64_t passes = 0;
    *ntok = NTOK_FALLBACK; *outp = 0;
    for (;;) {
        H.state = ST_HEADER;
        if (!header_by_wave(H, lds, w, hpos, limit)) return -9;            // the wave's tables differ from the one-lane builders'
        if (H.state != ST_DECODE) return 1;
        const uint32_t dstart = g_hdr_end;
        const uint32_t rem = limit > dstart ? limit - dstart : 0;
        uint32_t seg = (rem + 63) / 64; if (seg < 64) seg = 64;
        uint32_t s[64], bn[64]; Seg r[64]; bool ch[64];
        const uint32_t slot_cap = g_slot_cap;
        std::vector<uint32_t> slots_mem((size_t)64 * SLOT_TOKENS + 4, 0xDEADBEEFu);
        uint32_t* slots = slots_mem.data() + ((16 - ((uintptr_t)slots_mem.data() & 15)) & 15) / 4;
        for (int l = 0; l < 64; ++l) { s[l] = dstart + (uint32_t)l * seg; bn[l] = l == 63 ? MARK : s[l] + seg; ch[l] = true; }
        {   // warm-up: lane l > 0 enters the stream some segments early and takes the first symbol at or beyond its border as its start
            static const int wseg = getenv("THJ_SIM_WARMUP") ? atoi(getenv("THJ_SIM_WARMUP")) : 2;
            for (int l = 1; l < 64 && wseg > 0; ++l) {
                const uint32_t border = s[l];
                const uint32_t back = (uint32_t)std::min(l, wseg) * seg;
                const Seg wu = decode_segment<SEG_COUNT>(H.lit, H.A, H.B, w, limit, border - back, border, nullptr, 0, WaveCpu{});
                if (wu.e < MARK) s[l] = wu.e;
            }
            if (wseg > 0) ++passes;
        }
        for (;;) {
            for (int l = 0; l < 64; ++l) if (ch[l]) r[l] = decode_segment<SEG_SLOT>(H.lit, H.A, H.B, w, limit, s[l], bn[l], slots + (size_t)l * SLOT_TOKENS, slot_cap, WaveCpu{});
            ++passes;
            bool any = false;
            uint32_t ns[64];
            for (int l = 0; l < 64; ++l) { ns[l] = l ? r[l - 1].e : s[0]; ch[l] = ns[l] < MARK && ns[l] != s[l]; any = any || ch[l]; }       // a lane that failed says nothing about the next one's start
            if (!any) break;
            if (getenv("THJ_SIM_TRACE")) { fprintf(stderr, "round %lld changed:", (long long)passes); for (int l = 0; l < 64; ++l) if (ch[l]) fprintf(stderr, " %d(%+d)", l, (int)(ns[l] - s[l])); fprintf(stderr, "\n"); }
            for (int l = 0; l < 64; ++l) if (ch[l]) s[l] = ns[l];
        }
        if (getenv("THJ_SIM_SYNCSTAT")) {
            // how long a decoder entered at a segment border takes to fall into step with the true one (tokens of the true stream j, of the false one k)
            int mj = 0, mjk = 0, nosync = 0, early = 0; long sj = 0;
            for (int l = 1; l < 64; ++l) {
                if (s[l] >= MARK) continue;
                const uint32_t border = dstart + (uint32_t)l * seg;
                uint32_t px = s[l], py = border; int j = 0, k = 0; bool fail = false;
                while (px != py) {
                    const bool ax = px < py;
                    const uint32_t p0 = ax ? px : py;
                    if (p0 >= bn[l]) { fail = true; break; }
                    const Seg one = decode_segment<SEG_COUNT>(H.lit, H.A, H.B, w, limit, p0, p0 + 1, nullptr, 0, WaveCpu{});
                    if (one.e >= MARK) { if (!ax) ++early; fail = true; break; }
                    if (ax) { px = one.e; ++j; } else { py = one.e; ++k; }
                }
                if (fail) { ++nosync; continue; }
                mj = std::max(mj, j); mjk = std::max(mjk, j + k); sj += j;
            }
            fprintf(stderr, "SYNC seg_bits %u tokens %u maxj %d maxjk %d nosync %d early %d meanj %.1f\n", seg, 0u, mj, mjk, nosync, early, sj / 63.0);
        }
        // the first lane that did not reach its border ended the block (or the stream is bad); the lanes behind it decoded nothing real
        int el = -1; uint32_t tot_nt = 0, tot_ob = 0, off_nt[64], off_ob[64];
        for (int l = 0; l < 64 && el < 0; ++l) if (r[l].e >= MARK) el = l;
        if (el < 0 || r[el].e != MARK_EOB) return 1;
        for (int l = 0; l < 64; ++l) {
            if (l > el) { s[l] = MARK_NONE; r[l].nt = r[l].ob = 0; }
            off_nt[l] = tot_nt; off_ob[l] = tot_ob; tot_nt += r[l].nt; tot_ob += r[l].ob;
        }
        if (tok_base + tot_nt > TOKCAP || out_base + tot_ob > 65536u) return 1;
        bool fits = true;
        for (int l = 0; l < 64; ++l) fits = fits && r[l].nt <= slot_cap;
        if (fits) {
            bool ok = true;
            for (int l = 0; l < 64; ++l) ok = compact_segment(slots + (size_t)l * SLOT_TOKENS, r[l].nt, tokens + tok_base + off_nt[l], out_base + off_ob[l]) && ok;
            if (!ok) return 1;
        } else {
            for (int l = 0; l < 64; ++l) {
                const Seg f = decode_segment<SEG_FINAL>(H.lit, H.A, H.B, w, limit, s[l], bn[l], tokens + tok_base + off_nt[l], out_base + off_ob[l], WaveCpu{});
                if ((l <= el && f.e == MARK_ERR) || f.nt != r[l].nt || f.ob != r[l].ob) return 1;
            }
            ++passes;
        }
        tok_base += tot_nt; out_base += tot_ob;
        hpos = r[el].eob_pos;
        if (H.last) break;
    }
    *ntok = tok_base; *outp = out_base;
    if (passes_out) *passes_out = passes;
    return 0;
}

// thj_k_lz restated: returns the number of bytes written to out (65536 bytes), or -1 on an inconsistency
extern "C" int64_t inflate_sim_lz(const uint32_t* tokens, uint32_t n, uint8_t* out, int64_t* rounds_out) {
    constexpr uint32_t HIST = 32768, CAP = 32768 + 8192;
    std::vector<uint8_t> buf(CAP + 64, 0xEE);
    uint32_t origin = 0, flushed = 0, pos = 0, i0 = 0;
    int64_t rounds = 0;
    auto slide = [&]() {
        for (uint32_t o = flushed; o + 16 <= pos; o += 16) memcpy(out + o, &buf[o - origin], 16);
        flushed = std::max(flushed, pos & ~15u);
        const uint32_t no = pos > HIST ? (pos - HIST) & ~15u : 0u;
        if (no > origin) { memmove(buf.data(), buf.data() + (no - origin), pos - no); origin = no; }
    };
    while (i0 < n) {
        uint32_t len[64], incl[64], a[64], dist[64]; bool valid[64], ism[64], take[64];
        uint32_t run = 0;
        for (int l = 0; l < 64; ++l) {
            valid[l] = i0 + l < n;
            const uint32_t t = valid[l] ? tokens[i0 + l] : 0;
            ism[l] = valid[l] && (t >> 31);
            len[l] = !valid[l] ? 0 : ism[l] ? ((t >> 15) & 255u) + 3u : 1u;
            dist[l] = (t & 0x7FFFu) + 1u;
            run += len[l]; incl[l] = run;
        }
        const uint32_t room = origin + CAP - pos;
        int ntake = 0;
        for (int l = 0; l < 64; ++l) { take[l] = valid[l] && incl[l] <= room; if (take[l]) ++ntake; }
        if (ntake == 0) { slide(); if (origin + CAP - pos < 258) return -1; continue; }
        for (int l = 0; l < 64; ++l) a[l] = pos + incl[l] - len[l] - origin;
        for (int l = 0; l < ntake; ++l) if (!ism[l]) buf[a[l]] = (uint8_t)tokens[i0 + l];
        uint64_t pend = 0;
        for (int l = 0; l < ntake; ++l) if (ism[l]) { if (dist[l] > a[l] + origin) return -1; pend |= 1ull << l; }
        while (pend) {
            ++rounds;
            const int f = __builtin_ctzll(pend);
            const uint32_t hwm = a[f];
            uint64_t rdy = 0;
            for (int l = 0; l < ntake; ++l) if ((pend >> l) & 1) { const uint32_t s = a[l] - dist[l]; if (s + std::min(len[l], dist[l]) <= hwm) rdy |= 1ull << l; }
            // every ready lane reads before any writes (the lanes' sources are final bytes, their destinations disjoint)
            std::vector<std::vector<uint8_t>> tmp(64);
            for (int l = 0; l < ntake; ++l) if ((rdy >> l) & 1) { const uint32_t s = a[l] - dist[l]; tmp[l].resize(len[l]); for (uint32_t k = 0; k < len[l]; ++k) tmp[l][k] = buf[s + (dist[l] >= len[l] ? k : k % dist[l])]; }
            for (int l = 0; l < ntake; ++l) if ((rdy >> l) & 1) memcpy(&buf[a[l]], tmp[l].data(), len[l]);
            pend &= ~rdy;
        }
        pos += incl[ntake - 1]; i0 += (uint32_t)ntake;
    }
    for (uint32_t o = flushed; o < pos; ++o) out[o] = buf[o - origin];
    if (rounds_out) *rounds_out = rounds;
    return pos;
}
