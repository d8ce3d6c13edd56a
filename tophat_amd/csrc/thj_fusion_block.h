// thj_fusion_block.h -- find_fusions (segment_juncs.cpp:2976-3291) as ONE WORKGROUP of 256 threads runs it over its tiles of 256 reads: the
// body of thj_k_fusion (thj_segjuncs.hip), written against an execution context X so that tests/hostsim runs the same code over the fibers of
// simt.h (tests/test_hostsim_fusions.py) -- what a thread does for one read is fusion_read_pairs / fusion_read_mates / fusion_eval of thj_core.h.
//
// Three phases per tile: (a) every thread looks at its read's pairs of real hits (a few comparisons for almost every read) and queues the
// candidate pairs; a read that also wants the mate-anchored search -- two flank scans per mate hit -- is only listed; (b) once ~200 such reads
// are listed, they are searched with every lane busy (their candidate pairs join the queue); (c) once ~200 pairs are queued, detect_fusion
// (~20 000 instructions: it walks both flanks base by base) runs over them, a pair a lane.  With detect_fusion called where the pair is found,
// 2 % chimeric reads meant three waves in four ran it with one or two lanes active (4.6 ms per launch of 6.25 M reads); with the flank scans
// where the read is met, configs[3]'s shape ran them at one lane in ten (1.9-2.5 ms per launch).
//
// Round 6, reads of a repeat family.  (a') A read whose first and last segment have 64 or more (hit, hit) pairs -- k x k -- is not enumerated
// by its own thread: the workgroup takes such reads one after the other, a pair a thread.  (b') The mate-anchored part pairs every pseudo-hit
// (two a mate hit) with every first-segment hit, for every first-segment hit: k x k x 2 x k for a chimeric read whose first segment lies in
// the family -- 137 000 pairs at k = 41, and one thread took a second over them, the launch waiting (all of thj_k_fusion's 1.1 s a launch on
// bench.py's mix).  Such reads are listed too: the flank scan of every mate hit once (a thread each: it does not depend on the left hit), then
// per left hit the (mate hit, pseudo-hit, first-segment hit) triples a thread each.  The events are counted per key afterwards
// (FusionSimpleSet), so their order in the buffer means nothing.
//
// X: tid (0..255), lane, sync() (workgroup barrier), ballot(), shfl(v, src), shfl_up(v, d), atomic_add(uint32_t*, v), atomic_or(uint32_t*, v),
//    detect_now(...) (a pair evaluated where it is found, when the queue is full).
// B: the batch -- view(r) -> ReadView, read_len(r), planes(r), W.
// Out: ignored(ref), fusion(...) (one event), reserve(n) -> first position of n events, put(pos, ...).
#pragma once
#include "thj_core.h"

namespace thj {

struct FusTask { uint32_t read; uint32_t flags; Hit lh; Hit rh; };        // flags: rc | dir << 1
static constexpr int FUS_QCAP = 1024;
static constexpr int FUS_RCAP = 512;
static constexpr uint32_t FUS_HEAVY_READ_PAIRS = 64;
static constexpr uint64_t FUS_HEAVY_MATE_TRIPLES = 256;      // (first-segment hit, mate hit, first-segment hit) triples from which the workgroup takes a read's mate-anchored part
static constexpr int FUS_MSCAN = 64;                         // ... of reads with at most this many mate hits
struct FusBlockShared {
    FusTask q[FUS_QCAP];
    uint32_t rq[FUS_RCAP];
    uint32_t hq[256];
    int32_t ms_f[FUS_MSCAN], ms_r[FUS_MSCAN];
    uint8_t ms_ok[FUS_MSCAN];
    uint32_t q_n, rq_n, hq_n, h_partner;
};

THJ_HD bool fusion_pairs_heavy(const ReadView& v) {
    if (v.nseg == 0) return false;
    int last = v.nseg - 1;
    while (last > 0 && rv_count_raw(v, last) == 0) --last;
    if (last == 0) return false;
    const uint64_t pairs = (uint64_t)(v.so[1] - v.so[0]) * (v.so[last + 1] - v.so[last]);
    return pairs >= FUS_HEAVY_READ_PAIRS && pairs < (1ull << 31);            // (the workgroup's loop counts them in 32 bits)
}

template <class X, class Out>
struct FusDeferSink {
    X& x; Out& out; FusBlockShared& sh; uint32_t read;
    const Genome& g; const Params& p; const u64* rp; int W; int rl;
    THJ_HD bool ignored(uint32_t ref) const { return out.ignored(ref); }
    THJ_HD void defer(bool rc, const Hit& lh, const Hit& rh, int dir) {
        const uint32_t k = x.atomic_add(&sh.q_n, 1u);
        if (k < (uint32_t)FUS_QCAP) { FusTask t; t.read = read; t.flags = (rc ? 1u : 0u) | ((uint32_t)dir << 1); t.lh = lh; t.rh = rh; sh.q[k] = t; }
        else x.detect_now(g, p, rp, W, rl, rc, lh, rh, dir, out);
    }
};

// (c) detect_fusion over the queued pairs: every lane evaluates one; the wave then takes the room for all its events with one atomic (one per
// event was 7 x 10^5 returning atomics on one address per launch) and the lanes write theirs.  Called by all threads.
template <class X, class B, class Out>
THJ_HD void fusion_block_run_queue(X& x, const Genome& g, const Params& p, const B& b, Out& out, FusBlockShared& sh) {
    const uint32_t have = sh.q_n < (uint32_t)FUS_QCAP ? sh.q_n : (uint32_t)FUS_QCAP;
    for (uint32_t k0 = 0; k0 < have; k0 += 256u) {
        const uint32_t k = k0 + (uint32_t)x.tid;
        FusTask t; FusEval fe; int n = 0, rl = 0;
        t.read = 0; t.flags = 0;
        if (k < have && !THJ_EXPF(1 << 27)) {
            t = sh.q[k];
            rl = b.read_len(t.read);
            n = fusion_eval(g, p, b.planes(t.read), b.W, rl, (t.flags & 1u) != 0, t.lh, t.rh, (int)(t.flags >> 1), fe);
        }
        int incl = n;
        for (int d = 1; d < 64; d <<= 1) { const int y = x.shfl_up(incl, d); if (x.lane >= d) incl += y; }
        const int tot = x.shfl(incl, 63);
        unsigned long long base = 0;
        if (x.lane == 0 && tot) base = out.reserve((unsigned long long)tot);
        base = ((unsigned long long)(unsigned int)x.shfl((int)(base >> 32), 0) << 32) | (unsigned long long)(unsigned int)x.shfl((int)(base & 0xFFFFFFFFull), 0);
        if (n) {
            const unsigned long long first = base + (unsigned long long)(incl - n);
            fusion_emit(fe, rl, t.lh, t.rh, (int)(t.flags >> 1), [&](int kk, uint32_t r1, uint32_t r2, uint32_t l, uint32_t r, uint32_t dir, uint32_t ed) {
                out.put(first + (unsigned long long)kk, r1, r2, l, r, dir, ed);
            });
        }
    }
    x.sync();
    if (x.tid == 0) sh.q_n = 0;
    x.sync();
}

// the tiles first_tile, first_tile + tile_stride, ... of a batch of n_reads reads
template <class X, class B, class Out>
THJ_HD void fusion_block(X& x, const Genome& g, const Params& p, const B& b, int n_reads, int first_tile, int tile_stride, Out& out, FusBlockShared& sh) {
    const int tid = x.tid, lane = x.lane;
    if (tid == 0) { sh.q_n = 0; sh.rq_n = 0; sh.hq_n = 0; }
    x.sync();
    const int n_tiles = (n_reads + 255) / 256;
    for (int tile = first_tile; tile < n_tiles; tile += tile_stride) {
        const int r = tile * 256 + tid;
        const bool last = tile + tile_stride >= n_tiles;
        bool wants = false, heavy = false;
        if (r < n_reads) {
            ReadView v = b.view(r);
            heavy = fusion_pairs_heavy(v);
            FusDeferSink<X, Out> ds{x, out, sh, (uint32_t)r, g, p, v.rp, v.W, v.rl};
            if (!heavy && !THJ_EXPF(1 << 25)) wants = fusion_read_pairs<true>(g, p, v, ds);
        }
        {   // one LDS atomic per wave and list
            const unsigned long long mw = x.ballot(wants), mh = x.ballot(heavy);
            const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
            uint32_t base = 0, hbase = 0;
            if (lane == 0 && mw) base = x.atomic_add(&sh.rq_n, (uint32_t)popc((u64)mw));
            if (lane == 0 && mh) hbase = x.atomic_add(&sh.hq_n, (uint32_t)popc((u64)mh));
            base = (uint32_t)x.shfl((int)base, 0); hbase = (uint32_t)x.shfl((int)hbase, 0);
            if (wants) sh.rq[base + (uint32_t)popc((u64)(mw & below))] = (uint32_t)r;   // < 192 + 256 entries
            if (heavy) sh.hq[hbase + (uint32_t)popc((u64)(mh & below))] = (uint32_t)r;
        }
        x.sync();
        // (a') the tile's family reads, one after the other: fusion_read_pairs with a pair a thread
        const uint32_t n_heavy = sh.hq_n;
        for (uint32_t h = 0; h < n_heavy; ++h) {
            const int rr = (int)sh.hq[h];
            ReadView v = b.view(rr);
            int lastseg = v.nseg - 1;
            while (lastseg > 0 && rv_count_raw(v, lastseg) == 0) --lastseg;
            const uint32_t l0 = v.so[0], l1 = v.so[1], r0 = v.so[lastseg], r1 = v.so[lastseg + 1];
            const uint32_t nr = r1 - r0, total = (l1 - l0) * nr;
            if (tid == 0) sh.h_partner = 0u;
            x.sync();
            FusDeferSink<X, Out> ds{x, out, sh, (uint32_t)rr, g, p, v.rp, v.W, v.rl};
            for (uint32_t t0 = 0; t0 < total; t0 += 256u) {
                const uint32_t queued = sh.q_n;                                   // (nobody adds between the barrier before and the one below)
                x.sync();
                if (queued > (uint32_t)FUS_QCAP - 256u) fusion_block_run_queue(x, g, p, b, out, sh);
                const uint32_t t = t0 + (uint32_t)tid;
                if (t < total) {
                    const uint32_t i = t / nr, j = t - i * nr;
                    const Hit lh = v.hits[l0 + i], rh = v.hits[r0 + j];
                    if (lh.ref_id == rh.ref_id && hit_anti(lh) == hit_anti(rh)) {           // check_partner (:3082-3100)
                        const int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
                        if (dist > -p.max_insertion_length && dist <= p.fusion_min_dist) x.atomic_or(&sh.h_partner, 1u);
                    }
                    fusion_pair<true>(g, p, v.rp, v.W, v.rl, lh, rh, ds);
                }
                x.sync();
            }
            if (tid == 0 && !sh.h_partner && v.n_mate > 0 && l1 > l0) sh.rq[x.atomic_add(&sh.rq_n, 1u)] = (uint32_t)rr;      // (a tile's reads: still < 192 + 256)
            x.sync();
        }
        if (tid == 0) sh.hq_n = 0;
        x.sync();
        const uint32_t have_r = sh.rq_n;
        if (have_r >= 192u || (last && have_r > 0u)) {
            for (uint32_t k = (uint32_t)tid; k < have_r; k += 256u) {
                const int rr = (int)sh.rq[k];
                ReadView v = b.view(rr);
                const uint32_t nl = v.so[1] - v.so[0];
                if ((uint64_t)nl * nl * (uint32_t)v.n_mate >= FUS_HEAVY_MATE_TRIPLES && v.n_mate <= FUS_MSCAN && nl < (1u << 24)) {
                    const uint32_t at = x.atomic_add(&sh.hq_n, 1u);
                    if (at < 256u) { sh.hq[at] = (uint32_t)rr; continue; }
                }
                FusDeferSink<X, Out> ds{x, out, sh, (uint32_t)rr, g, p, v.rp, v.W, v.rl};
                if (!THJ_EXPF(1 << 26)) fusion_read_mates<true>(g, p, v, ds);
            }
            x.sync();
            // (b') fusion_read_mates by the workgroup, one listed read after the other
            const uint32_t n_hm = sh.hq_n < 256u ? sh.hq_n : 256u;
            for (uint32_t h = 0; h < n_hm && !THJ_EXPF(1 << 26); ++h) {
                const int rr = (int)sh.hq[h];
                ReadView v = b.view(rr);
                const uint32_t l0 = v.so[0], nl = v.so[1] - v.so[0];
                const int nm = v.n_mate;
                if (tid < nm) {
                    int32_t f = SLOT_NONE, rv = SLOT_NONE;
                    const bool ok = rescue_scan(g, p, v.rp, v.W, v.rl, v.mate[tid], f, rv);
                    sh.ms_f[tid] = f; sh.ms_r[tid] = rv; sh.ms_ok[tid] = ok ? 1u : 0u;
                }
                x.sync();
                FusDeferSink<X, Out> ds{x, out, sh, (uint32_t)rr, g, p, v.rp, v.W, v.rl};
                const int minus_dist = -p.max_insertion_length * 2;
                int cl = p.segment_length - p.segment_mismatches - 3; if (cl > 15) cl = 15;
                auto passed_over = [&](const Hit& lh, const Hit& rh) {                   // :3131-3137: a proper pair is no fusion candidate
                    if (lh.ref_id != rh.ref_id || hit_anti(lh) == hit_anti(rh)) return false;
                    const int dist = hit_anti(lh) ? lh.left - rh.right : rh.left - lh.right;
                    return dist > minus_dist && dist <= p.fusion_min_dist;
                };
                for (uint32_t l = 0; l < nl; ++l) {
                    const Hit lh = v.hits[l0 + l];
                    int m_stop = nm;                                                     // the mate loop ends at a flank that would start before the contig
                    for (int m = 0; m < nm; ++m) { if (passed_over(lh, v.mate[m])) continue; if (sh.ms_f[m] == SLOT_BREAK) { m_stop = m; break; } }
                    const uint32_t items = (uint32_t)m_stop * 2u * nl;
                    for (uint32_t t0 = 0; t0 < items; t0 += 256u) {
                        const uint32_t queued = sh.q_n;
                        x.sync();
                        if (queued > (uint32_t)FUS_QCAP - 256u) fusion_block_run_queue(x, g, p, b, out, sh);
                        const uint32_t t = t0 + (uint32_t)tid;
                        if (t < items) {
                            const uint32_t m = t / (2u * nl), rem = t - m * 2u * nl, k = rem / nl, i = rem - k * nl;
                            const Hit rh = v.mate[m];
                            const int32_t pos = k == 0 ? sh.ms_f[m] : sh.ms_r[m];
                            if (!passed_over(lh, rh) && sh.ms_ok[m] && pos != SLOT_NONE) {
                                Hit ph; ph.ref_id = rh.ref_id; ph.left = pos; ph.right = ph.left + cl;
                                ph.meta = (k == 0 ? 2u : 3u) | ((uint32_t)cl << 24);
                                fusion_pair<true>(g, p, v.rp, v.W, v.rl, v.hits[l0 + i], ph, ds);
                            }
                        }
                        x.sync();
                    }
                }
                x.sync();
            }
            if (tid == 0) { sh.rq_n = 0; sh.hq_n = 0; }
            x.sync();
        }
        if (sh.q_n >= 192u || (last && sh.q_n > 0u)) fusion_block_run_queue(x, g, p, b, out, sh);
        x.sync();
    }
}

}  // namespace thj
