// hostsim.cpp -- TEST-ONLY: compiles tophat_amd/csrc/thj_core.h for the CPU and
// runs the per-read logic serially, so the bit-parallel formulations can be
// checked against the oracle without a GPU.  Never linked into libthj_hip.so
// and never used by the product path, bench.py or smoke().
#include "../../include/thj.h"
#include "../../tophat_amd/csrc/thj_core.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace thj;

struct Collect {
    std::vector<thj_junction> juncs, dels;
    struct Ins { uint32_t ref, left; int len; uint32_t seq; u64 prio; };
    std::vector<Ins> ins;
    void junction(uint32_t ref, uint32_t l, uint32_t r, bool a) { juncs.push_back({ref, l, r, a ? 1u : 0u}); }
    std::vector<std::pair<uint32_t, thj_junction>> cov;      // (skip count, junction) of the coverage search
    void cov_junction(uint32_t ref, uint32_t l, uint32_t r, bool a, uint32_t skip) { cov.push_back({skip, {ref, l, r, a ? 1u : 0u}}); }
    void deletion(uint32_t ref, uint32_t l, uint32_t r) { dels.push_back({ref, l, r, 0u}); }
    void insertion(uint32_t ref, uint32_t l, int len, uint32_t seq, u64 prio) { ins.push_back({ref, l, len, seq, prio}); }
};

struct ExecSink {
    const Genome& g; const Params& p; const ReadView& v; Collect& c; uint32_t ordinal;
    int64_t n_windows = 0, n_indels = 0;
    void window(uint32_t ref, int32_t wl, int32_t wr, bool anti, int start, int slen) {
        ++n_windows;
        window_exec(g, p, v, ref, wl, wr, anti, start, slen, c);
    }
    void indel(int i, uint32_t lidx, uint32_t ridx, int li, int ri, bool anti, int plen, bool is_del) {
        ++n_indels;
        indel_exec(g, p, v, i, lidx, ridx, anti, plen, is_del, ins_prio(ordinal, i, li, ri), c);
    }
};

// the task words of the kernels' queues, executed where they are emitted (what thj_k_sj_tasks does with them)
struct DecodeEmit {
    const Genome& g; const Params& p; const ReadView& v; Collect& c; uint32_t ordinal;
    void task(bool valid, uint32_t a, uint32_t b, uint32_t cc, uint32_t d) {
        if (!valid) return;
        const bool anti = task_anti(a);
        if (task_is_indel(a)) {
            const int i = task_indel_i(a);
            indel_exec(g, p, v, i, b, cc, anti, task_indel_plen(a), task_is_del(a), ins_prio(ordinal, i, (int)(d & 0xFFFF), (int)(d >> 16)), c);
        } else
            window_exec(g, p, v, b, (int32_t)cc, (int32_t)d, anti, task_window_start(a), task_window_slen(a), c);
    }
};

// a read with at most one hit per segment the way thj_k_sj_flat / thj_k_sj_rescue_flat handle it; false: not such a read
template <int NS>
static bool flat_path(const Genome& g, const Params& p, const ReadView& v, Collect& c, uint32_t ordinal, int64_t& nw, int64_t& ni, int64_t& nr, int64_t& n_trivial) {
    if (v.nseg > NS) return false;
    uint32_t so[NS + 1];
    Hit h[NS];
    for (int s = 0; s <= NS; ++s) so[s] = v.so[s <= v.nseg ? s : v.nseg];
    for (int s = 0; s < NS; ++s) {
        if (so[s + 1] - so[s] > 1u) return false;
        h[s] = Hit{0, 0, 0, 0};
        if (so[s + 1] != so[s]) h[s] = v.hits[so[s]];
    }
    DecodeEmit em{g, p, v, c, ordinal};
    const FlatResult res = flat_read<NS>(p, v.nseg, so, h, v.rl, v.n_mate, em);
    nw += res.n_windows; ni += res.n_indels;
    if (!res.rescue && res.n_windows + res.n_indels == 0) ++n_trivial;
    if (res.rescue) {
        if (v.n_mate > 2) {                       // more mate hits than the flat rescue takes: the general rescue, the indel pairs being done
            ReadView w = v;
            bool wants;
            gaps_prepare(p, w, wants);
            std::vector<int32_t> slots((size_t)2 * rv_count_raw(w, 0) * w.n_mate, SLOT_NONE);
            const int n_left = rv_count_raw(w, 0);
            for (int l = 0; l < n_left; ++l)
                for (int m = 0; m < w.n_mate; ++m) {
                    if (rescue_pair(g, p, w.rp, w.W, w.rl, w.hits[w.so[0] + l], w.mate[m], slots[2 * (l * w.n_mate + m)], slots[2 * (l * w.n_mate + m) + 1])) ++nr;
                    if (slots[2 * (l * w.n_mate + m)] == SLOT_BREAK) break;
                }
            w.slots = slots.data(); w.rescue = true;
            ExecSink sink{g, p, w, c, ordinal};
            gaps_enumerate(p, w, sink);
            nw += sink.n_windows;
            return true;
        }
        Hit mh[2] = {Hit{0, 0, 0, 0}, Hit{0, 0, 0, 0}};
        int32_t sc[4] = {SLOT_NONE, SLOT_NONE, SLOT_NONE, SLOT_NONE};
        for (int m = 0; m < v.n_mate; ++m) {      // thj_k_sj_rescue_scan: every mate hit, whatever the left hit is
            mh[m] = v.mate[m];
            if (!rescue_scan(g, p, v.rp, v.W, v.rl, mh[m], sc[2 * m], sc[2 * m + 1]) && sc[2 * m] != SLOT_BREAK) sc[2 * m] = SLOT_UNSCANNED;
        }
        int w = 0;
        nr += flat_rescue<2>(p, so[1] != so[0], h[0], res.size, v.rl, mh, v.n_mate, sc, em, w);
        nw += w;
    }
    return true;
}

#include "simt.h"
// the wave operations wave_read_enumerate is written against, over simt.h's fibers
struct SjWaveSim {
    simt::Block* b; int tid, lane;
    unsigned long long ballot(bool p) { const uint32_t* a = b->exchange(tid, p ? 1u : 0u); unsigned long long m = 0; for (int i = 0; i < 64; ++i) m |= (unsigned long long)(a[i] & 1u) << i; return m; }
    uint32_t bcast(uint32_t v, int src) { return b->exchange(tid, v)[src & 63]; }
};
// tasks of one fiber, executed when the wave is done (the sinks of this file are not shared among fibers while they run)
struct WaveTaskLog {
    struct T { bool is_indel; int i; uint32_t lidx, ridx; int li, ri; bool anti; int plen; bool is_del; uint32_t ref; int32_t wl, wr; int start, slen; };
    std::vector<T> v;
    void window(uint32_t ref, int32_t wl, int32_t wr, bool anti, int start, int slen) { T t{}; t.is_indel = false; t.ref = ref; t.wl = wl; t.wr = wr; t.anti = anti; t.start = start; t.slen = slen; v.push_back(t); }
    void indel(int i, uint32_t lidx, uint32_t ridx, int li, int ri, bool anti, int plen, bool is_del) { T t{}; t.is_indel = true; t.i = i; t.lidx = lidx; t.ridx = ridx; t.li = li; t.ri = ri; t.anti = anti; t.plen = plen; t.is_del = is_del; v.push_back(t); }
};

extern "C" int hostsim_segjuncs(const thj_params* tp, const uint64_t* blocks, const uint32_t* contig_blk,
                                const int32_t* contig_len, int32_t n_contigs, const thj_seg_batch* b,
                                thj_junction** juncs, int64_t* n_juncs, thj_junction** dels, int64_t* n_dels,
                                uint32_t** ins /* 6 u32 per insertion: ref,left,len,seq,prio_lo,prio_hi */, int64_t* n_ins,
                                int64_t* stats /* windows, indel pairs, rescue pairs, reads skipped as trivial */) {
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    Params p;
    static_assert(sizeof(Params) == sizeof(thj_params), "params layout");
    memcpy(&p, tp, sizeof p);
    Collect c;
    int64_t nw = 0, ni = 0, nr = 0, n_trivial = 0;
    const bool lazy = getenv("THJ_HOSTSIM_LAZY") != nullptr;        // exercise the on-the-fly rescue of rv_foreach
    const bool no_skip = getenv("THJ_HOSTSIM_NO_SKIP") != nullptr;  // run the general enumeration on every read
    const bool no_flat = getenv("THJ_HOSTSIM_NO_FLAT") != nullptr;  // ... also on the reads with at most one hit per segment (the kernels give those to flat_read / flat_rescue)
    std::vector<int32_t> slots, mscan;
    std::vector<PHit> plist;
    const bool mscan_mode = getenv("THJ_HOSTSIM_MSCAN") != nullptr; // the rescue pairs from one scan per mate hit (ReadView::mscan)
    const bool wave_mode = getenv("THJ_HOSTSIM_WAVE") != nullptr;   // reads with several hits a segment by a wave of fibers with the hits in registers (wave_read_enumerate)
    const bool plist_mode = getenv("THJ_HOSTSIM_PLIST") != nullptr; // ... and the pseudo-hit list built once, a left hit at a time (rescue_pseudo_hits: thj_k_segjuncs_rescue_shared since round 6)
    for (int32_t r = 0; r < b->n_reads; ++r) {
        ReadView v;
        v.hits = (const Hit*)b->hits;
        v.so = b->seg_off + (int64_t)r * b->nseg;
        v.nseg = b->nseg;
        v.W = b->words_per_plane;
        v.rp = (const u64*)b->read_planes + (int64_t)r * 3 * v.W;
        v.rl = b->read_len[r];
        v.mate = nullptr; v.n_mate = 0; v.slots = nullptr; v.mscan = nullptr;
        if (b->mate_off) {
            v.mate = (const Hit*)b->mate_hits + b->mate_off[r];
            v.n_mate = (int)(b->mate_off[r + 1] - b->mate_off[r]);
        }
        ExecSink sink{g, p, v, c, b->ordinal_base + (uint32_t)r};
        if (!no_flat && (v.nseg <= 4 ? flat_path<4>(g, p, v, c, b->ordinal_base + (uint32_t)r, nw, ni, nr, n_trivial)
                         : v.nseg <= 8 ? flat_path<8>(g, p, v, c, b->ordinal_base + (uint32_t)r, nw, ni, nr, n_trivial)
                         : flat_path<16>(g, p, v, c, b->ordinal_base + (uint32_t)r, nw, ni, nr, n_trivial))) continue;
        if (!no_skip && read_is_trivial(p, v)) { ++n_trivial; continue; }      // nothing can come out of this read
        if (wave_mode && wave_read_fits(v)) {
            // thj_k_segjuncs_shared since round 6: the read by one wave, its hits in registers (wave_read_enumerate); a read that takes
            // the rescue is handed on (below, as the rescue kernels take it: indel pairs, rescue slots, the enumeration)
            WaveTaskLog logs[64];
            bool dg[64], wr[64];
            ReadView views[64];
            simt::run_block(64, [&](simt::Block& blk, int tid) {
                SjWaveSim x{&blk, tid, tid};
                views[tid] = v;
                dg[tid] = wave_read_enumerate(x, p, views[tid], logs[tid], wr[tid]);
            }, 256 * 1024);
            for (int l = 1; l < 64; ++l) if (dg[l] != dg[0] || wr[l] != wr[0]) return 8;       // the verdicts are the wave's
            if (!(dg[0] && wr[0])) {
                for (int l = 0; l < 64; ++l)
                    for (auto& t : logs[l].v) {
                        if (t.is_indel) sink.indel(t.i, t.lidx, t.ridx, t.li, t.ri, t.anti, t.plen, t.is_del);
                        else sink.window(t.ref, t.wl, t.wr, t.anti, t.start, t.slen);
                    }
                nw += sink.n_windows; ni += sink.n_indels;
                continue;
            }
            for (int l = 0; l < 64; ++l) if (!logs[l].v.empty()) return 9;                   // a rescue read: nothing enumerated here
        }
        indels_enumerate(p, v, sink);
        bool wants;
        if (gaps_prepare(p, v, wants)) {
            if (wants) {
                int n_left = rv_count_raw(v, 0);
                slots.assign((size_t)2 * n_left * v.n_mate, SLOT_NONE);
                for (int l = 0; l < n_left; ++l)
                    for (int m = 0; m < v.n_mate; ++m) {
                        if (rescue_pair(g, p, v.rp, v.W, v.rl, v.hits[v.so[0] + l], v.mate[m],
                                        slots[2 * (l * v.n_mate + m)], slots[2 * (l * v.n_mate + m) + 1])) ++nr;
                        if (slots[2 * (l * v.n_mate + m)] == SLOT_BREAK) break;       // the reference leaves the mate loop here
                    }
                v.slots = lazy ? nullptr : slots.data();     // lazy: the kernel's fallback when its LDS slot buffer is full
                if (mscan_mode) {                           // thj_k_segjuncs_rescue_shared: one scan per mate hit, the pairs looked up
                    mscan.assign((size_t)2 * v.n_mate, SLOT_NONE);
                    for (int m = 0; m < v.n_mate; ++m)
                        if (!rescue_scan(g, p, v.rp, v.W, v.rl, v.mate[m], mscan[2 * m], mscan[2 * m + 1]) && mscan[2 * m] != SLOT_BREAK) mscan[2 * m] = SLOT_UNSCANNED;
                    v.slots = nullptr; v.mscan = mscan.data();
                }
                if (plist_mode) {
                    mscan.assign((size_t)2 * v.n_mate, SLOT_NONE);
                    for (int m = 0; m < v.n_mate; ++m)
                        if (!rescue_scan(g, p, v.rp, v.W, v.rl, v.mate[m], mscan[2 * m], mscan[2 * m + 1]) && mscan[2 * m] != SLOT_BREAK) mscan[2 * m] = SLOT_UNSCANNED;
                    plist.clear();
                    int64_t scanned_pl = 0;
                    for (int l = 0; l < n_left; ++l) {           // the kernel: a lane per left hit, count, prefix sum, write
                        int sc = 0, dummy = 0;
                        const Hit lh = v.hits[v.so[0] + l];
                        const int mine = rescue_pseudo_hits(lh, v.mate, v.n_mate, mscan.data(), nullptr, sc);
                        const size_t at = plist.size();
                        plist.resize(at + (size_t)mine);
                        if (mine) rescue_pseudo_hits(lh, v.mate, v.n_mate, mscan.data(), plist.data() + at, dummy);
                        scanned_pl += sc;
                    }
                    int64_t nr_ref = 0;                          // the statistic must be the pair loop's
                    for (int l = 0; l < n_left; ++l)
                        for (int m = 0; m < v.n_mate; ++m) {
                            int32_t a, bb;
                            if (rescue_pair(g, p, v.rp, v.W, v.rl, v.hits[v.so[0] + l], v.mate[m], a, bb)) ++nr_ref;
                            if (a == SLOT_BREAK) break;
                        }
                    if (nr_ref != scanned_pl) { fprintf(stderr, "hostsim: rescue-pair statistic of the list differs (%lld vs %lld)\n", (long long)scanned_pl, (long long)nr_ref); return 7; }
                    v.slots = nullptr; v.mscan = nullptr; v.plist = plist.data(); v.n_plist = (int)plist.size();
                }
                v.lazy_g = &g; v.lazy_p = &p;
                v.rescue = true;
            }
            gaps_enumerate(p, v, sink);
        }
        nw += sink.n_windows; ni += sink.n_indels;
    }
    *n_juncs = (int64_t)c.juncs.size();
    *juncs = (thj_junction*)malloc(sizeof(thj_junction) * (c.juncs.size() + 1));
    memcpy(*juncs, c.juncs.data(), sizeof(thj_junction) * c.juncs.size());
    *n_dels = (int64_t)c.dels.size();
    *dels = (thj_junction*)malloc(sizeof(thj_junction) * (c.dels.size() + 1));
    memcpy(*dels, c.dels.data(), sizeof(thj_junction) * c.dels.size());
    *n_ins = (int64_t)c.ins.size();
    *ins = (uint32_t*)malloc(sizeof(uint32_t) * 6 * (c.ins.size() + 1));
    for (size_t k = 0; k < c.ins.size(); ++k) {
        uint32_t* o = *ins + 6 * k;
        o[0] = c.ins[k].ref; o[1] = c.ins[k].left; o[2] = (uint32_t)c.ins[k].len; o[3] = c.ins[k].seq;
        o[4] = (uint32_t)(c.ins[k].prio & 0xffffffffu); o[5] = (uint32_t)(c.ins[k].prio >> 32);
    }
    stats[0] = nw; stats[1] = ni; stats[2] = nr; stats[3] = n_trivial;
    return 0;
}

extern "C" void hostsim_free(void* p) { free(p); }

// ---------------------------------------------------------------- fusion search
struct FusCollect {
    std::vector<thj_fusion> v;
    std::vector<uint32_t> ign;
    bool ignored(uint32_t ref) const { for (uint32_t x : ign) if (x == ref) return true; return false; }
    void fusion(uint32_t r1, uint32_t r2, uint32_t l, uint32_t r, uint32_t dir, uint32_t ed) { v.push_back({r1, r2, l, r, dir, 1u, ed, 0u}); }
};

extern "C" int hostsim_fusions(const thj_params* tp, const uint64_t* blocks, const uint32_t* contig_blk, const int32_t* contig_len,
                               int32_t n_contigs, const thj_seg_batch* b, const uint32_t* ignore, int32_t n_ignore,
                               thj_fusion** out, int64_t* n_out) {
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    Params p;
    memcpy(&p, tp, sizeof p);
    FusCollect c;
    c.ign.assign(ignore, ignore + n_ignore);
    for (int32_t r = 0; r < b->n_reads; ++r) {
        ReadView v;
        v.hits = (const Hit*)b->hits;
        v.so = b->seg_off + (int64_t)r * b->nseg;
        v.nseg = b->nseg; v.W = b->words_per_plane;
        v.rp = (const u64*)b->read_planes + (int64_t)r * 3 * v.W;
        v.rl = b->read_len[r];
        v.mate = nullptr; v.n_mate = 0; v.slots = nullptr; v.mscan = nullptr;
        if (b->mate_off) { v.mate = (const Hit*)b->mate_hits + b->mate_off[r]; v.n_mate = (int)(b->mate_off[r + 1] - b->mate_off[r]); }
        fusion_read(g, p, v, c);
    }
    *n_out = (int64_t)c.v.size();
    *out = (thj_fusion*)malloc(sizeof(thj_fusion) * (c.v.size() + 1));
    memcpy(*out, c.v.data(), sizeof(thj_fusion) * c.v.size());
    return 0;
}

// The same through thj_k_fusion's workgroup algorithm (thj_fusion_block.h) over simt.h's fibers: one workgroup of 256 threads takes all the
// tiles -- or `blocks` of them take every blocks-th tile, one after the other (their queues and lists carry over from tile to tile as on the
// device).  The raw events come in another order than hostsim_fusions'; the caller reduces both.
#include "../../tophat_amd/csrc/thj_fusion_block.h"
struct FusBlockSim {
    simt::Block* blk; int tid, lane;
    void sync() { blk->barrier(); }
    unsigned long long ballot(bool q) { const uint32_t* a = blk->exchange(tid, q ? 1u : 0u); unsigned long long m = 0; for (int i = 0; i < 64; ++i) m |= (unsigned long long)(a[i] & 1u) << i; return m; }
    int shfl(int v, int src) { return (int)blk->exchange(tid, (uint32_t)v)[src & 63]; }
    int shfl_up(int v, int d) { const uint32_t* a = blk->exchange(tid, (uint32_t)v); return lane >= d ? (int)a[lane - d] : v; }
    uint32_t atomic_add(uint32_t* q, uint32_t v) { const uint32_t o = *q; *q = o + v; return o; }
    void atomic_or(uint32_t* q, uint32_t v) { *q |= v; }
    template <class Out>
    void detect_now(const Genome& g, const Params& p, const u64* rp, int W, int rl, bool rc, const Hit& lh, const Hit& rh, int dir, Out& out) { ++n_now; detect_fusion(g, p, rp, W, rl, rc, lh, rh, dir, out); }
    static int64_t n_now;
};
int64_t FusBlockSim::n_now = 0;
struct FusBlockCollect {
    std::vector<thj_fusion> v;
    std::vector<uint32_t> ign;
    bool ignored(uint32_t ref) const { for (uint32_t x : ign) if (x == ref) return true; return false; }
    void fusion(uint32_t r1, uint32_t r2, uint32_t l, uint32_t r, uint32_t dir, uint32_t ed) { v.push_back({r1, r2, l, r, dir, 1u, ed, 0u}); }
    unsigned long long reserve(unsigned long long n) { const unsigned long long at = v.size(); v.resize(v.size() + (size_t)n); return at; }
    void put(unsigned long long pos, uint32_t r1, uint32_t r2, uint32_t l, uint32_t r, uint32_t dir, uint32_t ed) { v[(size_t)pos] = thj_fusion{r1, r2, l, r, dir, 1u, ed, 0u}; }
};
struct FusSimBatch {
    const thj_seg_batch* b; int W;
    ReadView view(int r) const {
        ReadView v;
        v.hits = (const Hit*)b->hits;
        v.so = b->seg_off + (int64_t)r * b->nseg;
        v.nseg = b->nseg; v.W = b->words_per_plane;
        v.rp = (const u64*)b->read_planes + (int64_t)r * 3 * v.W;
        v.rl = b->read_len[r];
        v.mate = nullptr; v.n_mate = 0; v.slots = nullptr; v.mscan = nullptr;
        if (b->mate_off) { v.mate = (const Hit*)b->mate_hits + b->mate_off[r]; v.n_mate = (int)(b->mate_off[r + 1] - b->mate_off[r]); }
        return v;
    }
    int read_len(uint32_t r) const { return b->read_len[r]; }
    const u64* planes(uint32_t r) const { return (const u64*)b->read_planes + (int64_t)r * 3 * b->words_per_plane; }
};
extern "C" int hostsim_fusions_block(const thj_params* tp, const uint64_t* blocks, const uint32_t* contig_blk, const int32_t* contig_len,
                                     int32_t n_contigs, const thj_seg_batch* b, const uint32_t* ignore, int32_t n_ignore, int32_t n_blocks,
                                     thj_fusion** out, int64_t* n_out, int64_t* n_evaluated_in_place) {
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    Params p;
    memcpy(&p, tp, sizeof p);
    FusBlockCollect c;
    c.ign.assign(ignore, ignore + n_ignore);
    FusSimBatch sb{b, b->words_per_plane};
    FusBlockSim::n_now = 0;
    if (n_blocks < 1) n_blocks = 1;
    for (int blk_i = 0; blk_i < n_blocks; ++blk_i) {
        FusBlockShared sh;
        simt::run_block(256, [&](simt::Block& blk, int tid) {
            FusBlockSim x{&blk, tid, tid & 63};
            fusion_block(x, g, p, sb, b->n_reads, blk_i, n_blocks, c, sh);
        }, 512 * 1024);
    }
    *n_evaluated_in_place = FusBlockSim::n_now;
    *n_out = (int64_t)c.v.size();
    *out = (thj_fusion*)malloc(sizeof(thj_fusion) * (c.v.size() + 1));
    memcpy(*out, c.v.data(), sizeof(thj_fusion) * c.v.size());
    return 0;
}

// ---------------------------------------------------------------- long_spanning_reads
#include "../../tophat_amd/csrc/thj_span_core.h"

struct VecSink {
    std::vector<OutAln>* v;
    void emit(const OutAln& o) { v->push_back(o); }
    void emit_words(const uint32_t* w) { OutAln o; memcpy(&o, w, sizeof o); v->push_back(o); }
    void set_count(uint32_t, int) {}
};

static int64_t g_wave_reads = 0;            // reads the packed tier finished since the last call of hostsim_wave_reads
extern "C" int64_t hostsim_wave_reads() { const int64_t n = g_wave_reads; g_wave_reads = 0; return n; }
static int64_t g_chain_groups = 0;          // multihit reads whose chains travelled as chain entries (thj_k_chains)
extern "C" int64_t hostsim_chain_groups() { const int64_t n = g_chain_groups; g_chain_groups = 0; return n; }
static int64_t g_chain_deferred = 0;        // ... of them, the ones whose join searched a closure
extern "C" int64_t hostsim_chain_deferred() { const int64_t n = g_chain_deferred; g_chain_deferred = 0; return n; }
static int64_t g_chain_reads = 0;           // reads that travelled as chain entries (tier 0 -> join -> finish) since the last call
extern "C" int64_t hostsim_chain_reads() { const int64_t n = g_chain_reads; g_chain_reads = 0; return n; }
// the wave operations span_pack_wave is written against, over simt.h's fibers (one wave = 64 fibers)
struct WaveSimX {
    simt::Block* b; int tid, lane;
    uint64_t ballot(bool p) { const uint32_t* a = b->exchange(tid, p ? 1u : 0u); uint64_t m = 0; for (int i = 0; i < 64; ++i) m |= (uint64_t)(a[i] & 1u) << i; return m; }
    uint32_t bcast(uint32_t v, int src) { return b->exchange(tid, v)[src & 63]; }
    uint32_t shfl(uint32_t v, int src) { return b->exchange(tid, v)[src & 63]; }
    uint32_t incl_scan(uint32_t v) { const uint32_t* a = b->exchange(tid, v); uint32_t s = 0; for (int i = 0; i <= lane; ++i) s += a[i]; return s; }
    uint32_t wmax(uint32_t v) { const uint32_t* a = b->exchange(tid, v); uint32_t m = 0; for (int i = 0; i < 64; ++i) m = a[i] > m ? a[i] : m; return m; }
    void wsync() { b->exchange(tid, 0); }
    unsigned long long clock() { return 0; }
};

// the packed multihit tier over `list` (64 entries per wave at a time, as thj_k_stitch_pack draws them): records into outs[read],
// the entries it hands on into `fwd`
template <int MS, int MAXROOTS, int MAXHITS, int CL>
static int run_pack(const Genome& g, const Params& p, const SpanSets& S, int32_t nseg, int32_t W, const uint32_t* seg_off, const void* hits,
                    const uint64_t* planes, const uint16_t* read_len, const uint8_t* quals, int32_t qual_stride,
                    const std::vector<uint32_t>& list, std::vector<std::vector<OutAln>>& outs, std::vector<uint32_t>& fwd) {
    static PackLds<MS, MAXHITS, CL> lds;
    for (size_t i0 = 0; i0 < list.size(); i0 += 64) {
        std::vector<OutAln> lane_out[64];
        bool lane_fwd[64];
        simt::run_block(64, [&](simt::Block& blk, int tid) {
            WaveSimX x{&blk, tid, tid};
            VecSink ls{&lane_out[tid]};
            const bool has = i0 + (size_t)tid < list.size();
            lane_fwd[tid] = span_pack_wave<MS, MAXROOTS, MAXHITS, CL>(x, g, p, S, (const SpanHit*)hits, (const SpanHitHead*)nullptr, seg_off, nseg, (const u64*)planes, W,
                                                                      read_len, quals, qual_stride, has ? list[i0 + (size_t)tid] : 0u, has, lds, ls);
        }, 256 * 1024);
        for (int l = 0; l < 64; ++l) {
            if (i0 + (size_t)l >= list.size()) { if (lane_fwd[l]) return -22; continue; }
            if (lane_fwd[l]) fwd.push_back(list[i0 + (size_t)l]); else ++g_wave_reads;
        }
        for (int l = 0; l < 64; ++l)
            for (auto& o : lane_out[l]) outs[o.read_idx].push_back(o);
    }
    for (uint32_t r : fwd) if (!outs[r].empty()) return -23;       // a read that goes on has emitted nothing here
    return 0;
}

extern "C" int hostsim_spanning(const thj_params* tp, const uint64_t* blocks, const uint32_t* contig_blk,
                                const int32_t* contig_len, int32_t n_contigs,
                                int32_t n_reads, int32_t nseg, int32_t W, const uint32_t* seg_off, const void* hits,
                                const uint64_t* planes, const uint16_t* read_len, const uint8_t* quals, int32_t qual_stride,
                                const thj_junction* juncs, int64_t n_juncs,
                                const uint32_t* ins /* 4 u32 each: ref,left,len,seq3 */, int64_t n_ins,
                                int32_t mode /* 0 = the four tiers as the kernels run them, 1 = generic only, 2 = as 0 without the packed multihit tier, 3 = as 0 with the packed tier's limits (roots, hits per round) made tiny: many rounds, many hand-overs */,
                                void** out, int64_t* n_out, int64_t* status_counts /* [5] */) {
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    Params p;
    memcpy(&p, tp, sizeof p);
    std::vector<u64> jk((size_t)n_juncs), ik((size_t)n_ins);
    std::vector<uint32_t> iseq((size_t)n_ins);
    for (int64_t i = 0; i < n_juncs; ++i) jk[i] = junc_key(g, juncs[i].ref_id, juncs[i].left, juncs[i].right, juncs[i].antisense != 0);
    for (int64_t i = 0; i < n_ins; ++i) { ik[i] = ins_key(g, ins[4 * i], ins[4 * i + 1], (int)ins[4 * i + 2]); iseq[i] = ins[4 * i + 3]; }
    for (int64_t i = 1; i < n_juncs; ++i) if (jk[i] <= jk[i - 1]) return -10;   // must already be sorted unique
    for (int64_t i = 1; i < n_ins; ++i) if (ik[i] <= ik[i - 1]) return -11;
    SpanSets S{jk.data(), n_juncs, ik.data(), iseq.data(), n_ins, nullptr, 0};
    // mode 0 runs as the kernels do: with the coarse bucket index over the junction keys
    std::vector<uint32_t> bucket;
    if (mode != 1) {
        int64_t n_blocks = 0;
        for (int32_t k = 0; k < n_contigs; ++k) { int64_t e = (int64_t)contig_blk[k] + (contig_len[k] + 63) / 64 + 2; if (e > n_blocks) n_blocks = e; }
        const int64_t nb = ((n_blocks * 64 + 2) >> JUNC_BUCKET_SHIFT) + 1;
        bucket.resize((size_t)nb + 1);
        for (int64_t bk = 0; bk <= nb; ++bk)
            bucket[(size_t)bk] = (uint32_t)lower_bound_u64(jk.data(), n_juncs, (u64)bk << (JUNC_BUCKET_SHIFT + 30));
        bucket[(size_t)nb] = (uint32_t)n_juncs;
        S.junc_bucket = bucket.data(); S.n_buckets = nb;
    }
    std::vector<std::vector<OutAln>> outs((size_t)n_reads);
    std::vector<uint32_t> multi, gen;
    status_counts[0] = status_counts[1] = status_counts[2] = status_counts[3] = status_counts[4] = 0;
    for (int32_t r = 0; r < n_reads; ++r) {
        VecSink sink{&outs[(size_t)r]};
        int st = SPAN_NEED_GENERIC;
        if (mode != 1) {
            // modes 0 and 3: plain one-hit-per-segment reads travel as chain entries (tier 0 -> thj_k_join -> thj_k_finish); mode 2 keeps
            // every such read on span_read_lean
            ChainEntry ent;
            st = span_read_contig(g, p, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                                  read_len[r], quals + (int64_t)r * qual_stride, (uint32_t)r, sink, (mode == 0 || mode == 3) ? &ent : nullptr);
            if ((st & 0xFF) == SPAN_NEED_CHAIN) {
                status_counts[4]++; ++g_chain_reads;
                RAln res;
                SpanHit ch[CHAIN_MAXSEG];
                for (int s = 0; s < CHAIN_MAXSEG; ++s) ch[s] = ((const SpanHit*)hits)[ent.hit[s]];
                // as thj_k_join runs it: the abutting chains' pass first, the closure search only for what that pass defers
                int jr = chain_join<true>(g, p, S, ch, ent.meta, (const u64*)planes + (int64_t)ent.read * 3 * W, W, res);
                if (jr == LJ_DEFER) { jr = chain_join<false>(g, p, S, ch, ent.meta, (const u64*)planes + (int64_t)ent.read * 3 * W, W, res); ++g_chain_deferred; }
                if (jr == LJ_PUNT) { gen.push_back((uint32_t)r); continue; }       // more cigar ops than LEAN_C: thj_k_join hands the read to the general tier
                if (jr == LJ_OK) {
                    Q16 ja, jb, jc;
                    joined_pack(res, ent.read, chain_nsegs(ent.meta) == 1, chain_q(ent.meta), chain_k(ent.meta), ja, jb, jc);
                    joined_finish(g, p, ja, jb, jc, (const u64*)planes, W, read_len, quals, qual_stride, 0, sink);
                }
                st = SPAN_OK;
            }
            if ((st & 0xFF) == SPAN_NEED_LEAN) {
                status_counts[4]++;
                SpanHitHead stage[SPAN_MAXSEG];
                st = span_read_lean(g, p, S, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                                    read_len[r], quals + (int64_t)r * qual_stride, (uint32_t)r, stage, sink);
                if (st == SPAN_INCOMPAT) st = SPAN_OK;
            }
        }
        if (st == SPAN_NEED_GENERIC) (((mode == 0 || mode == 3) && nseg <= SPAN_MIDSEG) ? multi : gen).push_back((uint32_t)r);      // (reads of more than eight segments skip the packed tier, as in thj_span_run_async)
        else status_counts[st]++;
    }
    if (!multi.empty() && mode == 0 && nseg <= CHAIN_MAXSEG) {          // (mode 3 keeps every multihit read on the packed tier)
        // thj_k_chains: the multihit reads whose chains are known without a search travel as chain entries, each with its rank
        // among the read's chains; thj_k_finish numbers the records by counting the lower-ranked siblings that were reported
        std::vector<uint32_t> rest;
        for (uint32_t r : multi) {
            const uint32_t* so = seg_off + (int64_t)r * nseg;
            int nsegs = 0;
            while (nsegs < nseg && so[nsegs + 1] > so[nsegs]) ++nsegs;
            bool ok = nsegs > 0 && (((const SpanHit*)hits)[so[nsegs - 1]].meta & SH_END) != 0;
            if (ok && p.bowtie2) for (int s2 = 0; s2 < nsegs; ++s2) ok = ok && !((int)(so[s2 + 1] - so[s2]) > p.max_seg_multihits);
            if (!ok) { status_counts[SPAN_OK]++; continue; }              // the worker's early outs (:2777-2785, :2625-2632): nothing for this read
            struct HostTab {
                const SpanHit* h0;
                SpanHitHead head(int j) const {
                    const SpanHit& x = h0[j];
                    int right = x.left; int n = (int)(x.meta >> 24); if (n > 5) n = 5;
                    for (int i = 0; i < n; ++i) { const int op = cig_op(x.cigar[i]); if (op == OP_MATCH || op == OP_REF_SKIP || op == OP_DEL) right += (int)cig_len(x.cigar[i]); }
                    return SpanHitHead{x.ref_id, x.left, x.meta, (uint32_t)right};
                }
            } tab{(const SpanHit*)hits + so[0]};
            uint32_t off[CHAIN_MAXSEG + 1];
            for (int s2 = 0; s2 <= nsegs && s2 <= CHAIN_MAXSEG; ++s2) off[s2] = so[s2] - so[0];
            uint32_t sel[CHAINS_MAX]; int q[CHAINS_MAX];
            const int k = nsegs <= CHAIN_MAXSEG ? chains_discover(p, tab, off, nsegs, sel, q) : CHAINS_DECLINE;
            if (k == CHAINS_DECLINE) { rest.push_back(r); continue; }
            ++g_chain_groups;
            VecSink sink{&outs[(size_t)r]};
            int order = 0;
            bool punt = false;
            std::vector<OutAln> keep = outs[(size_t)r];
            for (int rank = 0; rank < k && !punt; ++rank)
                for (int c = 0; c < k; ++c) {
                    if (q[c] != rank) continue;
                    SpanHit ch[CHAIN_MAXSEG];
                    for (int s2 = 0; s2 < CHAIN_MAXSEG; ++s2) ch[s2] = ((const SpanHit*)hits)[so[0] + ((sel[c] >> (4 * (s2 < nsegs ? s2 : 0))) & 15u)];
                    const uint32_t meta = chain_meta(nsegs, rank, k, read_len[r]);
                    RAln res;
                    int jr = chain_join<true>(g, p, S, ch, meta, (const u64*)planes + (int64_t)r * 3 * W, W, res);
                    if (jr == LJ_DEFER) jr = chain_join<false>(g, p, S, ch, meta, (const u64*)planes + (int64_t)r * 3 * W, W, res);
                    if (jr == LJ_PUNT) { punt = true; break; }
                    if (jr != LJ_OK) continue;
                    Q16 ja, jb, jc;
                    joined_pack(res, r, nsegs == 1, rank, k, ja, jb, jc);
                    if (joined_finish(g, p, ja, jb, jc, (const u64*)planes, W, read_len, quals, qual_stride, order, sink)) ++order;
                }
            if (punt) { outs[(size_t)r] = keep; gen.push_back(r); continue; }     // (the kernels: a chain that needs more cigar ops sends the read to the general tier; see thj_k_join)
            status_counts[SPAN_OK]++;
        }
        multi.swap(rest);
    }
    if (!multi.empty()) {          // tier 2: the multihit list in batches of 64 entries, lanes as fibers
        int rc;
        if (mode == 3) rc = run_pack<SPAN_MIDSEG, 6, 20, 64>(g, p, S, nseg, W, seg_off, hits, planes, read_len, quals, qual_stride, multi, outs, gen);
        else if (nseg <= 4) rc = run_pack<4, 64, 256, 128>(g, p, S, nseg, W, seg_off, hits, planes, read_len, quals, qual_stride, multi, outs, gen);
        else rc = run_pack<SPAN_MIDSEG, 64, 256, 128>(g, p, S, nseg, W, seg_off, hits, planes, read_len, quals, qual_stride, multi, outs, gen);
        if (rc) return rc;
        status_counts[SPAN_OK] += (int64_t)(multi.size() - gen.size());
        std::sort(gen.begin(), gen.end());
    }
    for (uint32_t r : gen) {
        VecSink sink{&outs[(size_t)r]};
        int st = SPAN_NEED_GENERIC;
        if (mode != 1) {          // tier 3, first attempt: DFS over global memory, lean joins
            SpanHit stage[SPAN_MAXSEG];
            st = span_read_multi<48>(g, p, S, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                                     read_len[r], quals + (int64_t)r * qual_stride, r, stage, sink);
        }
        if (st == SPAN_NEED_GENERIC) {                       // tier 3: the general arrays
            status_counts[3]++;
            st = span_read(g, p, S, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                           read_len[r], quals + (int64_t)r * qual_stride, r, sink);
            // thj_k_stitch_huge: a read with more joined alignments than a thread's own array holds is done again with a workspace of
            // 2 * cap records (the list and the merge sort's scratch) -- THJ_HOSTSIM_HUGE_CAP=cap; without it the read is only counted
            const int huge_cap = getenv("THJ_HOSTSIM_HUGE_CAP") ? atoi(getenv("THJ_HOSTSIM_HUGE_CAP")) : 0;
            if (st == SPAN_TOO_MANY_JOINED && huge_cap > 0) {
                std::vector<Aln> ws((size_t)2 * huge_cap);
                st = span_read(g, p, S, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                               read_len[r], quals + (int64_t)r * qual_stride, r, sink, ws.data(), huge_cap);
            }
        }
        status_counts[st]++;
    }
    // read order; a read's records in rank order (the packed tier's lanes emit them in lane order)
    std::vector<OutAln> res;
    for (auto& v : outs) {
        std::stable_sort(v.begin(), v.end(), [](const OutAln& a, const OutAln& b) { return a.order < b.order; });
        for (auto& o : v) res.push_back(o);
    }
    *n_out = (int64_t)res.size();
    *out = malloc(sizeof(OutAln) * (res.size() + 1));
    memcpy(*out, res.data(), sizeof(OutAln) * res.size());
    return 0;
}

// ---- --fusion-search: tier 0 as the kernel runs it, then thj_span_fusion.h for every read it does not finish
// THJ_HOSTSIM_QR_VERIFY=1: dfs_seg_hits' quick "no" (fus_quick_reject) is only recorded, the whole pair test runs, and a pair it would have
// passed over that the whole test accepts is counted (hostsim_qr_counts: how often it said no, how often wrongly)
#define THJ_QR_VERIFY
static bool thj_qr_verify = false;
static int64_t thj_qr_said_no = 0, thj_qr_wrong = 0;
#include "../../tophat_amd/csrc/thj_span_fusion.h"
extern "C" void hostsim_qr_counts(int64_t* said_no, int64_t* wrong) { *said_no = thj_qr_said_no; *wrong = thj_qr_wrong; thj_qr_said_no = thj_qr_wrong = 0; }

struct FusWaveSim {
    simt::Block* b; int tid, lane;
    void sync() { b->barrier(); }
    uint32_t atomic_add(uint32_t* q, uint32_t v) { const uint32_t o = *q; *q = o + v; return o; }
    void mark(int, int) {}
    void counts(uint32_t, uint32_t) {}
    unsigned long long ballot(bool q) { const uint32_t* a = b->exchange(tid, q ? 1u : 0u); unsigned long long m = 0; for (int i = 0; i < 64; ++i) m |= (unsigned long long)(a[i] & 1u) << i; return m; }
};

extern "C" int hostsim_spanning_fusion(const thj_params* tp, const uint64_t* blocks, const uint32_t* contig_blk,
                                       const int32_t* contig_len, int32_t n_contigs,
                                       int32_t n_reads, int32_t nseg, int32_t W, const uint32_t* seg_off, const void* hits,
                                       const uint64_t* planes, const uint16_t* read_len, const uint8_t* quals, int32_t qual_stride,
                                       const thj_junction* juncs, int64_t n_juncs, const uint32_t* ins, int64_t n_ins,
                                       const thj_span_fusion* fus, int64_t n_fus, int32_t skip_tier0,
                                       void** out, int64_t* n_out, int64_t* status_counts /* [5] */) {
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    Params p;
    memcpy(&p, tp, sizeof p);
    std::vector<u64> jk((size_t)n_juncs), ik((size_t)n_ins);
    std::vector<uint32_t> iseq((size_t)n_ins);
    for (int64_t i = 0; i < n_juncs; ++i) jk[i] = junc_key(g, juncs[i].ref_id, juncs[i].left, juncs[i].right, juncs[i].antisense != 0);
    for (int64_t i = 0; i < n_ins; ++i) { ik[i] = ins_key(g, ins[4 * i], ins[4 * i + 1], (int)ins[4 * i + 2]); iseq[i] = ins[4 * i + 3]; }
    for (int64_t i = 1; i < n_juncs; ++i) if (jk[i] <= jk[i - 1]) return -10;
    for (int64_t i = 1; i < n_ins; ++i) if (ik[i] <= ik[i - 1]) return -11;
    SpanSets S{jk.data(), n_juncs, ik.data(), iseq.data(), n_ins, nullptr, 0};
    FusionSet F{(const FusKey*)fus, n_fus};
    std::vector<OutAln> res;
    VecSink sink{&res};
    for (int k = 0; k < 5; ++k) status_counts[k] = 0;
    thj_qr_verify = getenv("THJ_HOSTSIM_QR_VERIFY") != nullptr;
    const int fuswave_cap = getenv("THJ_HOSTSIM_FUSWAVE") ? atoi(getenv("THJ_HOSTSIM_FUSWAVE")) : 0;      // joined alignments the wave's workspace holds
    for (int32_t r = 0; r < n_reads; ++r) {
        int st = SPAN_NEED_GENERIC;
        if (!skip_tier0)
            st = span_read_contig(g, p, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                                  read_len[r], quals + (int64_t)r * qual_stride, (uint32_t)r, sink);
        if ((st & 0xFF) == SPAN_NEED_LEAN) {            // tier 1 as the kernel runs it: a read it cannot join goes on to the fusion tier
            status_counts[4]++;
            SpanHitHead stage[SPAN_MAXSEG];
            const size_t before = res.size();
            st = span_read_lean(g, p, S, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                                read_len[r], quals + (int64_t)r * qual_stride, (uint32_t)r, stage, sink);
            (void)before;
            if (st == SPAN_INCOMPAT) st = p.fusion_search ? SPAN_NEED_GENERIC : SPAN_OK;
        }
        if (st == SPAN_NEED_GENERIC && fuswave_cap > 0) {        // thj_k_stitch_huge's way: the 64 lanes of a wave on the read (fusion_read_wave)
            status_counts[3]++;
            std::vector<OutAln> mine;
            VecSink ws_sink{&mine};
            std::vector<char> wsp(fus_wave_ws_bytes(fuswave_cap));
            FusWaveShared sh;
            int sts[64], nrec[64];
            simt::run_block(64, [&](simt::Block& blk, int tid) {
                FusWaveSim x{&blk, tid, tid};
                sts[tid] = fusion_read_wave(x, g, p, S, F, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                                            read_len[r], quals + (int64_t)r * qual_stride, (uint32_t)r, ws_sink, wsp.data(), fuswave_cap, sh, nrec[tid]);
            }, 256 * 1024);
            for (int l = 1; l < 64; ++l) if (sts[l] != sts[0] || nrec[l] != nrec[0]) return -12;      // the verdict and the count are the wave's
            if ((size_t)nrec[0] != mine.size()) return -13;
            std::stable_sort(mine.begin(), mine.end(), [](const OutAln& a, const OutAln& b) { return a.order < b.order; });
            for (size_t k = 0; k < mine.size(); ++k) { if (mine[k].order != k) return -14; res.push_back(mine[k]); }
            st = sts[0];
        } else if (st == SPAN_NEED_GENERIC) {
            status_counts[3]++;
            st = span_read_fusion(g, p, S, F, (const SpanHit*)hits, seg_off + (int64_t)r * nseg, nseg, (const u64*)planes + (int64_t)r * 3 * W, W,
                                  read_len[r], quals + (int64_t)r * qual_stride, (uint32_t)r, sink);
        }
        status_counts[st]++;
    }
    *n_out = (int64_t)res.size();
    *out = malloc(sizeof(OutAln) * (res.size() + 1));
    memcpy(*out, res.data(), sizeof(OutAln) * res.size());
    return 0;
}

// ---- coverage search (thj_cov_core.h): the kernels of thj_covsearch_impl.h as plain loops over their thread index
#include "../../tophat_amd/csrc/thj_cov_core.h"
#include <algorithm>

// One shard's coverage-search state (what thj_covsearch_device_state exposes on the device): coverage words, per-contig
// sizes, extension-table entries of the shard's hits and unmapped reads.
extern "C" int hostsim_coverage_state(const uint32_t* contig_blk, const int32_t* contig_len, int32_t n_contigs, int64_t n_blocks,
                                      const thj_hit* hits, int64_t n_hits, const uint64_t* ium_planes, const uint16_t* ium_lens, int64_t n_ium, int32_t W,
                                      uint64_t* bits /* n_blocks */, int32_t* sizes /* n_contigs */, uint32_t* keys /* n_ium: the read records' lengths */, uint64_t* vals /* their 2-bit strings */) {
    using namespace thj::cov;
    Layout L{contig_blk, contig_len, n_contigs, n_blocks};
    for (int64_t i = 0; i < n_hits; ++i)
        add_hit(L, *(const Hit*)&hits[i], [&](int64_t w, u64 m) { bits[w] |= m; }, [&](int k, int32_t sz) { if (sz > sizes[k]) sizes[k] = sz; });
    for (int64_t r = 0; r < n_ium; ++r) read_record((const u64*)ium_planes, ium_lens, W, keys, (u64*)vals, 0, r);
    return 0;
}

// The pass from (merged) state: thj_covsearch_merge_async's rule is OR of the coverage words, max of the sizes,
// concatenation of the entries -- done by the caller -- then thj_covsearch_run_async's kernels as loops.
extern "C" int hostsim_coverage_run(const uint64_t* blocks, const uint32_t* contig_blk, const int32_t* contig_len, int32_t n_contigs,
                                    int64_t n_blocks, const uint64_t* bits, const int32_t* sizes, const uint32_t* keys, const uint64_t* vals, int64_t n_ext,
                                    int32_t min_cov_length, int32_t min_intron, int32_t max_intron, int64_t max_juncs, thj_junction** out, int64_t* n_out) {
    using namespace thj::cov;
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    Layout L{contig_blk, contig_len, n_contigs, n_blocks};
    const int64_t nw = n_blocks;
    std::vector<u64> bm((size_t)nw * 8, 0);
    u64 *covb = bm.data(), *le = covb + nw, *ll = le + nw, *lr = ll + nw, *fd = lr + nw, *ra = fd + nw, *fa = ra + nw, *rd = fa + nw;
    memcpy(covb, bits, (size_t)nw * 8);
    // the table from the read records (n_ext of them): entries, ordered by seed (the device lays them out by a counting sort)
    std::vector<std::pair<uint32_t, u64>> ent;
    for (int64_t i = 0; i < n_ext; ++i) record_entries(keys[i], (u64)vals[i], [&](uint32_t k, u64 v) { ent.push_back({k, v}); });
    std::stable_sort(ent.begin(), ent.end(), [](const std::pair<uint32_t, u64>& a, const std::pair<uint32_t, u64>& b) { return a.first < b.first; });
    std::vector<uint32_t> skeys(ent.size() + 1); std::vector<u64> svals(ent.size() + 1);
    for (size_t i = 0; i < ent.size(); ++i) { skeys[i] = ent[i].first; svals[i] = ent[i].second; }
    std::vector<uint32_t> off((size_t)N_KEYS + 2);
    for (uint32_t k = 0; k <= N_KEYS; ++k) key_offset(skeys.data(), (int64_t)ent.size(), off.data(), k);
    for (int64_t w = 0; w < nw; ++w) long_enough_word(L, covb, le, min_cov_length - 1, w);
    for (int64_t w = 0; w < nw; ++w) look_word(L, le, sizes, ll, lr, w);
    for (int i = 0; i < 2 * n_contigs; ++i) drop_windows(L, sizes, ll, lr, i);
    for (int64_t w = 0; w < nw; ++w) site_word(g, L, ll, lr, fd, ra, fa, rd, w);
    Collect c;
    // the Bloom filter over the entries, deliberately small here (2^16 bits) so that false positives happen too
    const u64 fmask = (1ull << 16) - 1;
    std::vector<u64> filter((size_t)((fmask + 1) / 64), 0);
    for (size_t i = 0; i < ent.size(); ++i)
        if (skeys[i] < N_KEYS) entry_filter_bits(skeys[i], svals[i], fmask, [&](u64 b) { filter[(size_t)(b >> 6)] |= 1ull << (b & 63); });
    ExtTable et{off.data(), svals.data(), filter.data(), fmask};
    for (int64_t w = 0; w < nw; ++w) pair_word(g, L, et, fd, fa, 0, min_intron, max_intron, w, c);
    for (int64_t w = 0; w < nw; ++w) pair_word(g, L, et, ra, rd, 1, min_intron, max_intron, w, c);
    // the max_cov_juncs cut as thj_covsearch_finish makes it: the smallest max_juncs by (skip count, junction)
    auto jl = [](const thj_junction& a, const thj_junction& b) {
        if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
        if (a.left != b.left) return a.left < b.left;
        if (a.right != b.right) return a.right < b.right;
        return a.antisense < b.antisense;
    };
    std::sort(c.cov.begin(), c.cov.end(), [&](const std::pair<uint32_t, thj_junction>& a, const std::pair<uint32_t, thj_junction>& b) {
        if (a.first != b.first) return a.first < b.first;
        return jl(a.second, b.second);
    });
    if ((int64_t)c.cov.size() > max_juncs) c.cov.resize((size_t)max_juncs);
    *n_out = (int64_t)c.cov.size();
    *out = (thj_junction*)malloc(sizeof(thj_junction) * (c.cov.size() + 1));
    for (size_t i = 0; i < c.cov.size(); ++i) (*out)[i] = c.cov[i].second;
    return 0;
}

// single shard: state, then the pass
extern "C" int hostsim_coverage_search(const uint64_t* blocks, const uint32_t* contig_blk, const int32_t* contig_len, int32_t n_contigs,
                                       int64_t n_blocks, const thj_hit* hits, int64_t n_hits,
                                       const uint64_t* ium_planes, const uint16_t* ium_lens, int64_t n_ium, int32_t W,
                                       int32_t min_cov_length, int32_t min_intron, int32_t max_intron, int64_t max_juncs,
                                       thj_junction** out, int64_t* n_out) {
    std::vector<uint64_t> bits((size_t)n_blocks, 0), vals((size_t)n_ium + 1);
    std::vector<int32_t> sizes((size_t)n_contigs + 1, 0);
    std::vector<uint32_t> keys((size_t)n_ium + 1);
    hostsim_coverage_state(contig_blk, contig_len, n_contigs, n_blocks, hits, n_hits, ium_planes, ium_lens, n_ium, W, bits.data(), sizes.data(), keys.data(), vals.data());
    return hostsim_coverage_run(blocks, contig_blk, contig_len, n_contigs, n_blocks, bits.data(), sizes.data(), keys.data(), vals.data(), n_ium,
                                min_cov_length, min_intron, max_intron, max_juncs, out, n_out);
}


// ---- butterfly search: thj_butterfly_run's kernels as loops (islands -> candidate positions -> sites -> (site, extension) keys sorted and
// made distinct -> join -> the cut by (intron length, junction))
extern "C" int hostsim_butterfly_search(const uint64_t* blocks, const uint32_t* contig_blk, const int32_t* contig_len, int32_t n_contigs,
                                        int64_t n_blocks, const thj_hit* hits, int64_t n_hits,
                                        const uint64_t* ium_planes, const uint16_t* ium_lens, int64_t n_ium, int32_t W,
                                        int32_t min_intron, int32_t max_intron, int64_t max_juncs, thj_junction** out, int64_t* n_out) {
    using namespace thj::cov;
    std::vector<uint64_t> bits((size_t)n_blocks, 0), vals((size_t)n_ium + 1);
    std::vector<int32_t> sizes((size_t)n_contigs + 1, 0);
    std::vector<uint32_t> keys((size_t)n_ium + 1);
    hostsim_coverage_state(contig_blk, contig_len, n_contigs, n_blocks, hits, n_hits, ium_planes, ium_lens, n_ium, W, bits.data(), sizes.data(), keys.data(), vals.data());
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    Layout L{contig_blk, contig_len, n_contigs, n_blocks};
    const int64_t nw = n_blocks;
    std::vector<std::pair<uint32_t, u64>> ent;
    for (int64_t i = 0; i < n_ium; ++i) record_entries(keys[i], (u64)vals[i], [&](uint32_t k, u64 v) { ent.push_back({k, v}); });
    std::stable_sort(ent.begin(), ent.end(), [](const std::pair<uint32_t, u64>& a, const std::pair<uint32_t, u64>& b) { return a.first < b.first; });
    std::vector<uint32_t> skeys(ent.size() + 1); std::vector<u64> svals(ent.size() + 1);
    for (size_t i = 0; i < ent.size(); ++i) { skeys[i] = ent[i].first; svals[i] = ent[i].second; }
    std::vector<uint32_t> off((size_t)N_KEYS + 2);
    for (uint32_t k = 0; k <= N_KEYS; ++k) key_offset(skeys.data(), (int64_t)ent.size(), off.data(), k);
    ExtTable et{off.data(), svals.data(), nullptr, 0};
    std::vector<u64> bm((size_t)nw * 6, 0);
    u64 *V = bm.data(), *E = V + nw, *fd = E + nw, *ra = fd + nw, *fa = ra + nw, *rd = fa + nw;
    memcpy(V, bits.data(), (size_t)nw * 8);
    for (int k = 0; k < n_contigs; ++k) bf_drop_tail(L, V, k);
    for (int64_t w = 0; w < nw; ++w) bf_eligible_word(L, V, E, w);
    for (int64_t w = 0; w < nw; ++w) bf_site_word(g, L, E, fd, ra, fa, rd, w);
    std::vector<u64> lk, rk;
    for (int side = 0; side < 2; ++side)
        for (int o = 0; o < 2; ++o) {
            const u64* sites = side ? (o ? rd : fa) : (o ? ra : fd);
            for (int64_t w = 0; w < nw; ++w) {
                u64 b = sites[w];
                if (!b) continue;
                const int k = contig_of(L, w);
                const int64_t pos0 = (w - (int64_t)contig_blk[k]) * 64;
                while (b) {
                    const int bit = __builtin_ctzll(b);
                    b &= b - 1;
                    bf_site_keys(g, L, et, (u64)(pos0 + bit) | ((u64)k << 32) | ((u64)o << 63), side != 0, [&](u64 key) { (side ? rk : lk).push_back(key); });
                }
            }
        }
    std::sort(lk.begin(), lk.end()); lk.erase(std::unique(lk.begin(), lk.end()), lk.end());
    std::sort(rk.begin(), rk.end()); rk.erase(std::unique(rk.begin(), rk.end()), rk.end());
    Collect c;
    for (u64 key : lk) { const BfRange m = bf_match_range(L, rk.data(), (int64_t)rk.size(), key, min_intron, max_intron); bf_emit_pairs(m, rk.data(), key, c); }
    auto jl = [](const thj_junction& a, const thj_junction& b) {
        if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
        if (a.left != b.left) return a.left < b.left;
        if (a.right != b.right) return a.right < b.right;
        return a.antisense < b.antisense;
    };
    std::sort(c.cov.begin(), c.cov.end(), [&](const std::pair<uint32_t, thj_junction>& a, const std::pair<uint32_t, thj_junction>& b) {
        if (a.first != b.first) return a.first < b.first;
        return jl(a.second, b.second);
    });
    c.cov.erase(std::unique(c.cov.begin(), c.cov.end(), [&](const std::pair<uint32_t, thj_junction>& a, const std::pair<uint32_t, thj_junction>& b) { return a.first == b.first && !jl(a.second, b.second) && !jl(b.second, a.second); }), c.cov.end());
    if ((int64_t)c.cov.size() > max_juncs) c.cov.resize((size_t)max_juncs);
    *n_out = (int64_t)c.cov.size();
    *out = (thj_junction*)malloc(sizeof(thj_junction) * (c.cov.size() + 1));
    for (size_t i = 0; i < c.cov.size(); ++i) (*out)[i] = c.cov[i].second;
    return 0;
}


// ---- microexon search: the kernel logic of thj_cov_core.h (candidates, table entries, per-window pairing) as host loops around the
// product's own window merge (csrc/host/thj_mx_host.h) -- the path thj_microexon_collect / _candidates / _run take on the device
#include "../../tophat_amd/csrc/host/thj_mx_host.h"
extern "C" int hostsim_microexon(const thj_params* tp, const uint64_t* blocks, const uint32_t* contig_blk, const int32_t* contig_len, int32_t n_contigs,
                                 const thj_seg_batch* const* batches, const int32_t* sides, int32_t n_batches, int32_t min_intron, int64_t max_juncs,
                                 thj_junction** out, int64_t* n_out, int64_t* n_windows) {
    using namespace thj::cov;
    Genome g{(const u64*)blocks, contig_blk, contig_len, n_contigs};
    std::vector<thj_mx_cand> cands;
    for (int bi = 0; bi < n_batches; ++bi) {
        const thj_seg_batch* b = batches[bi];
        for (int32_t r = 0; r < b->n_reads; ++r)
            mx_read_candidates(g, (const Hit*)b->hits, b->seg_off + (int64_t)r * b->nseg, b->nseg, (const u64*)b->read_planes + (int64_t)r * 3 * b->words_per_plane,
                               b->words_per_plane, (int)b->read_len[r], tp->segment_length, tp->min_anchor_len,
                               [&](int rank, uint32_t ref, int lb, int rb, u64 str, int n) {
                                   cands.push_back(thj_mx_cand{b->ordinal_base + (uint32_t)r, (uint16_t)rank, (uint8_t)sides[bi], (uint8_t)n, ref, lb, rb, 0u, str});
                               });
    }
    std::reverse(cands.begin(), cands.end());                     // the device appends in any order: the merge must sort
    thjh::MxWindows mw = thjh::mx_merge_windows(cands);
    if (n_windows) *n_windows = (int64_t)mw.windows.size();
    std::vector<std::pair<u64, u64>> ent;
    for (size_t i = 0; i < mw.strs.size(); ++i) mx_string_entries(mw.strs[i], (int)mw.str_len[i], (u64)mw.str_window[i], [&](u64 k, u64 v) { ent.push_back({k, v}); });
    std::stable_sort(ent.begin(), ent.end(), [](const std::pair<u64, u64>& a, const std::pair<u64, u64>& b) { return a.first < b.first; });
    std::vector<u64> keys(ent.size() + 1), vals(ent.size() + 1);
    for (size_t i = 0; i < ent.size(); ++i) { keys[i] = ent[i].first; vals[i] = ent[i].second; }
    MxTable t{keys.data(), vals.data(), (int64_t)ent.size()};
    Collect c;
    for (size_t wi = 0; wi < mw.windows.size(); ++wi) {
        const thj_mx_window& w = mw.windows[wi];
        const int64_t len = g_len(g, w.ref_id);
        if (w.left < 0 || w.right >= len - 1) continue;
        const int64_t w0 = w.left >> 6;
        const int n_words = (int)mx_window_words(w.left, w.right);
        std::vector<u64> bm[4];
        for (auto& v : bm) v.assign((size_t)n_words, 0);
        for (int j = 0; j < n_words; ++j) { const MxSites s = mx_site_word(g, w.ref_id, w.left, w.right, tp->library_type, w.side, j); bm[0][j] = s.fd; bm[1][j] = s.ra; bm[2][j] = s.fa; bm[3][j] = s.rd; }
        for (int o = 0; o < 2; ++o)
            for (int j = 0; j < n_words; ++j) {
                u64 bits = bm[o][j];
                while (bits) {
                    const int b = __builtin_ctzll(bits);
                    bits &= bits - 1;
                    mx_pair_site(g, t, (u64)wi, w.ref_id, len, bm[2 + o].data(), w0, n_words, o, min_intron, (w0 + j) * 64 + b, c);
                }
            }
    }
    auto jl = [](const thj_junction& a, const thj_junction& b) {
        if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
        if (a.left != b.left) return a.left < b.left;
        if (a.right != b.right) return a.right < b.right;
        return a.antisense < b.antisense;
    };
    std::sort(c.cov.begin(), c.cov.end(), [&](const std::pair<uint32_t, thj_junction>& a, const std::pair<uint32_t, thj_junction>& b) {
        if (a.first != b.first) return a.first < b.first;
        return jl(a.second, b.second);
    });
    c.cov.erase(std::unique(c.cov.begin(), c.cov.end(), [&](const std::pair<uint32_t, thj_junction>& a, const std::pair<uint32_t, thj_junction>& b) { return a.first == b.first && !jl(a.second, b.second) && !jl(b.second, a.second); }), c.cov.end());
    if ((int64_t)c.cov.size() > max_juncs) c.cov.resize((size_t)max_juncs);
    *n_out = (int64_t)c.cov.size();
    *out = (thj_junction*)malloc(sizeof(thj_junction) * (c.cov.size() + 1));
    for (size_t i = 0; i < c.cov.size(); ++i) (*out)[i] = c.cov[i].second;
    return 0;
}
