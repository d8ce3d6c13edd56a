// thj_cpuport.cpp -- BENCH INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg): the C ABI of include/thj.h, as far as the
// executables' host-reader paths use it, answered by the CPU oracle (oracle/liborc.so, the plain-C restatement of the
// reference) on host threads.  tools/cpuport/Makefile compiles the UNCHANGED host sources of the two executables
// (tophat_amd/csrc/host/segment_juncs_main.cpp, long_spanning_reads_main.cpp) against this file instead of libthj_hip.so:
// the result is a files-in -> files-out CPU path (BGZF inflate, BAM parse, batching, the reference's algorithm on the host's
// cores, BAM encode, BGZF deflate) to time beside the GPU executables on the same files -- `cpu_baseline.files_to_files`,
// kind "port, files to files".  Nothing in the product path links, loads or calls this; the product library has no CPU path.
//
// What is answered: contexts, the genome (unpacked back to ASCII for the oracle), thj_batch_upload / thj_segjuncs_* ,
// thj_span_sets_upload / thj_span_batch_upload / thj_span_* , page-locked buffers (plain malloc).  The device-side ingest and
// the device-side BAM writer answer THJ_EFALLBACK, which is the executables' documented way onto their host readers and host
// writer; fusion / coverage / microexon search answer an error (the figure is for the default mode of configs[1]).
// Threads: THJ_CPUPORT_THREADS (default 1): a batch is cut into that many runs of reads, each run one oracle call.
#include "../../include/thj.h"
#include "../../oracle/thj_oracle.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

void thj_set_error(const char* fmt, ...);       // thj_pack.cpp

static int port_threads() {
    static const int n = [] { const char* e = getenv("THJ_CPUPORT_THREADS"); int v = e ? atoi(e) : 1; return v < 1 ? 1 : v; }();
    return n;
}

struct SegBatchHost {
    thj_seg_batch d{};
    std::vector<uint32_t> seg_off, mate_off;
    std::vector<thj_hit> hits, mate_hits;
    std::vector<uint64_t> planes;
    std::vector<uint16_t> lens;
};
struct SpanBatchHost {
    thj_span_batch d{};
    std::vector<uint32_t> seg_off;
    std::vector<thj_span_hit> hits;
    std::vector<uint64_t> planes;
    std::vector<uint16_t> lens;
    std::vector<uint8_t> quals;
};

struct thj_ctx {
    std::vector<std::string> contigs;
    std::vector<const char*> seq;
    std::vector<int64_t> len;
    orc_genome g{};
    // segment_juncs' sets (std::set orders: junctions.h:39-57; insertions.h:52-67, first inserted wins)
    std::set<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>> juncs, dels;
    std::map<std::tuple<uint32_t, uint32_t, size_t>, std::pair<uint64_t, std::string>> ins;
    thj_segjuncs_counts counts{};
    // long_spanning_reads
    std::vector<orc_junction> span_juncs;
    std::vector<orc_ins_in> span_ins;
    std::vector<thj_aln> alns;
};

extern "C" int thj_device_count(void) { return 1; }
extern "C" int thj_ctx_create(int, void*, thj_ctx** out) { *out = new thj_ctx(); return THJ_OK; }
extern "C" int thj_ctx_warm(thj_ctx*, int) { return THJ_OK; }
extern "C" void thj_ctx_destroy(thj_ctx* c) { delete c; }
extern "C" int thj_ctx_sync(thj_ctx*) { return THJ_OK; }

extern "C" int thj_genome_upload(thj_ctx* c, const uint64_t* blocks, int64_t, const uint32_t* contig_blk, const int64_t* lens, int32_t n_contigs) {
    c->contigs.assign((size_t)n_contigs, std::string());
    c->seq.assign((size_t)n_contigs, nullptr);
    c->len.assign(lens, lens + n_contigs);
    for (int32_t i = 0; i < n_contigs; ++i) {
        if (lens[i] == 0) continue;                 // an @SQ entry with no FASTA record: rt.get_seq() == NULL
        std::string& s = c->contigs[(size_t)i];
        s.resize((size_t)lens[i]);
        const uint64_t* b = blocks + (uint64_t)contig_blk[i] * 4;
        for (int64_t k = 0; k < lens[i]; ++k) {
            const uint64_t* blk = b + (k >> 6) * 4; const int bit = (int)(k & 63);
            s[(size_t)k] = ((blk[2] >> bit) & 1) ? 'N' : "ACGT"[((blk[0] >> bit) & 1) | (((blk[1] >> bit) & 1) << 1)];
        }
        c->seq[(size_t)i] = s.c_str();
    }
    c->g.n_contigs = n_contigs; c->g.seq = c->seq.data(); c->g.len = c->len.data();
    return THJ_OK;
}

// bit planes (thj_reads_pack) back to the ASCII the oracle reads
static void unpack_reads(int64_t n, int W, const uint64_t* planes, const uint16_t* lens, std::string& bases, std::vector<int64_t>& off) {
    off.assign(1, 0);
    int64_t total = 0;
    for (int64_t r = 0; r < n; ++r) total += lens[r];
    bases.resize((size_t)total);
    int64_t o = 0;
    for (int64_t r = 0; r < n; ++r) {
        const uint64_t* rp = planes + r * 3 * W;
        for (int k = 0; k < lens[r]; ++k) {
            const int w = k >> 6, b = k & 63;
            bases[(size_t)o++] = ((rp[2 * W + w] >> b) & 1) ? 'N' : "ACGT"[((rp[w] >> b) & 1) | (((rp[W + w] >> b) & 1) << 1)];
        }
        off.push_back(o);
    }
}

// ------------------------------------------------------------------------------------------------ segment_juncs
extern "C" int thj_batch_upload(thj_ctx*, const thj_seg_batch* h, int64_t n_hits, int64_t n_mate_hits, thj_seg_batch** out) {
    SegBatchHost* b = new SegBatchHost();
    const size_t n = (size_t)h->n_reads;
    b->seg_off.assign(h->seg_off, h->seg_off + n * h->nseg + 1);
    b->hits.assign(h->hits, h->hits + n_hits);
    b->planes.assign(h->read_planes, h->read_planes + n * 3 * h->words_per_plane);
    b->lens.assign(h->read_len, h->read_len + n);
    b->d = *h;
    b->d.seg_off = b->seg_off.data(); b->d.hits = b->hits.data(); b->d.read_planes = b->planes.data(); b->d.read_len = b->lens.data();
    if (h->mate_off) {
        b->mate_off.assign(h->mate_off, h->mate_off + n + 1);
        b->mate_hits.assign(h->mate_hits, h->mate_hits + n_mate_hits);
        b->d.mate_off = b->mate_off.data(); b->d.mate_hits = b->mate_hits.data();
    }
    *out = &b->d;                 // d is the first member: the descriptor's address is the batch's
    return THJ_OK;
}
extern "C" int thj_batch_free(thj_ctx*, thj_seg_batch* dev) { delete reinterpret_cast<SegBatchHost*>(dev); return THJ_OK; }
extern "C" int thj_segjuncs_reset_async(thj_ctx* c) { c->juncs.clear(); c->dels.clear(); c->ins.clear(); c->counts = thj_segjuncs_counts{}; return THJ_OK; }

extern "C" int thj_segjuncs_run_async(thj_ctx* c, const thj_params* tp, const thj_seg_batch* db) {
    static_assert(sizeof(orc_hit) == sizeof(thj_hit), "hit layouts");
    orc_params p;
    memcpy(&p, tp, sizeof p);                       // the first twelve fields of thj_params, in this order
    const int64_t n = db->n_reads;
    if (n == 0) return THJ_OK;
    std::string bases; std::vector<int64_t> read_off;
    unpack_reads(n, db->words_per_plane, db->read_planes, db->read_len, bases, read_off);
    std::vector<int64_t> seg_off((size_t)n * db->nseg + 1), mate_off;
    for (size_t i = 0; i < seg_off.size(); ++i) seg_off[i] = db->seg_off[i];
    if (db->mate_off) { mate_off.resize((size_t)n + 1); for (size_t i = 0; i < mate_off.size(); ++i) mate_off[i] = db->mate_off[i]; }
    std::vector<uint32_t> ids((size_t)n);
    for (int64_t r = 0; r < n; ++r) ids[(size_t)r] = db->ordinal_base + (uint32_t)r;
    int T = port_threads();
    if ((int64_t)T > n) T = (int)n;
    std::vector<orc_events> ev((size_t)T);
    std::vector<int> rc((size_t)T, 0);
    auto work = [&](int t) {
        const int64_t r0 = n * t / T, r1 = n * (t + 1) / T;
        orc_batch b{};
        b.n_reads = (int32_t)(r1 - r0); b.nseg = db->nseg;
        b.read_id = ids.data() + r0; b.read_off = read_off.data() + r0; b.bases = bases.data();
        b.seg_off = seg_off.data() + r0 * db->nseg; b.hits = (const orc_hit*)db->hits;
        if (db->mate_off) { b.mate_off = mate_off.data() + r0; b.mate_hits = (const orc_hit*)db->mate_hits; }
        rc[(size_t)t] = orc_segjuncs_batch(&p, &c->g, &b, &ev[(size_t)t]);
    };
    if (T > 1) { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
    else work(0);
    for (int t = 0; t < T; ++t) {                   // the runs' sets in read order: the order the reference inserts in
        if (rc[(size_t)t]) { thj_set_error("cpu port: orc_segjuncs_batch failed"); return THJ_ENOMEM; }
        orc_events& e = ev[(size_t)t];
        for (int64_t i = 0; i < e.n_juncs; ++i) c->juncs.emplace(e.juncs[i].ref_id, e.juncs[i].left, e.juncs[i].right, e.juncs[i].antisense);
        for (int64_t i = 0; i < e.n_deletions; ++i) c->dels.emplace(e.deletions[i].ref_id, e.deletions[i].left, e.deletions[i].right, e.deletions[i].antisense);
        const uint64_t base = ((uint64_t)db->ordinal_base + (uint64_t)(n * t / T)) << 24;
        for (int64_t i = 0; i < e.n_insertions; ++i) {
            const orc_insertion& x = e.insertions[i];
            const std::string s(x.seq, strnlen(x.seq, sizeof x.seq));
            const auto key = std::make_tuple(x.ref_id, x.left, s.size());
            const uint64_t prio = base + x.prio;
            auto it = c->ins.find(key);
            if (it == c->ins.end()) c->ins.emplace(key, std::make_pair(prio, s));
            else if (prio < it->second.first) it->second = std::make_pair(prio, s);
        }
        c->counts.n_windows += e.n_windows; c->counts.n_indel_pairs += e.n_indel_pairs; c->counts.n_rescue_pairs += e.n_rescue_pairs;
        orc_events_free(&e);
    }
    c->counts.n_hits_read += db->seg_off[(size_t)n * db->nseg];
    return THJ_OK;
}
extern "C" int thj_segjuncs_finish(thj_ctx* c, thj_segjuncs_counts* counts) {
    c->counts.n_juncs = (int64_t)c->juncs.size(); c->counts.n_deletions = (int64_t)c->dels.size(); c->counts.n_insertions = (int64_t)c->ins.size();
    *counts = c->counts;
    return THJ_OK;
}
extern "C" int thj_segjuncs_download(thj_ctx* c, thj_junction* j, thj_junction* d, thj_insertion* ins) {
    for (const auto& x : c->juncs) *j++ = thj_junction{std::get<0>(x), std::get<1>(x), std::get<2>(x), std::get<3>(x)};
    for (const auto& x : c->dels) *d++ = thj_junction{std::get<0>(x), std::get<1>(x), std::get<2>(x), std::get<3>(x)};
    for (const auto& x : c->ins) {
        thj_insertion o{};
        o.ref_id = std::get<0>(x.first); o.left = std::get<1>(x.first); o.prio = x.second.first;
        strncpy(o.seq, x.second.second.c_str(), sizeof o.seq - 1);
        *ins++ = o;
    }
    return THJ_OK;
}

// ------------------------------------------------------------------------------------------------ long_spanning_reads
extern "C" int thj_span_sets_upload(thj_ctx* c, const thj_junction* juncs, int64_t n_juncs, const uint32_t* ins, int64_t n_ins) {
    static_assert(sizeof(orc_junction) == sizeof(thj_junction), "junction layouts");
    c->span_juncs.assign((const orc_junction*)juncs, (const orc_junction*)juncs + n_juncs);
    c->span_ins.clear();
    for (int64_t i = 0; i < n_ins; ++i) {           // rows {ref_id, left, length, bases 3 bits each (A,C,G,T,N = 0..4)}
        orc_ins_in x{};
        x.ref_id = ins[4 * i]; x.left = ins[4 * i + 1];
        const uint32_t len = ins[4 * i + 2], bits = ins[4 * i + 3];
        for (uint32_t k = 0; k < len && k < sizeof x.seq - 1; ++k) x.seq[k] = "ACGTN"[(bits >> (3 * k)) & 7u];
        c->span_ins.push_back(x);
    }
    return THJ_OK;
}
extern "C" int thj_span_batch_upload(thj_ctx*, const thj_span_batch* h, int64_t n_hits, thj_span_batch** out) {
    SpanBatchHost* b = new SpanBatchHost();
    const size_t n = (size_t)h->n_reads;
    b->seg_off.assign(h->seg_off, h->seg_off + n * h->nseg + 1);
    b->hits.assign(h->hits, h->hits + n_hits);
    b->planes.assign(h->read_planes, h->read_planes + n * 3 * h->words_per_plane);
    b->lens.assign(h->read_len, h->read_len + n);
    b->quals.assign(h->quals, h->quals + n * h->qual_stride);
    b->d = *h;
    b->d.seg_off = b->seg_off.data(); b->d.hits = b->hits.data(); b->d.read_planes = b->planes.data(); b->d.read_len = b->lens.data();
    b->d.quals = b->quals.data(); b->d.hit_heads = nullptr;
    *out = &b->d;
    return THJ_OK;
}
extern "C" int thj_span_batch_free(thj_ctx*, thj_span_batch* dev) { delete reinterpret_cast<SpanBatchHost*>(dev); return THJ_OK; }
extern "C" int thj_span_reset_async(thj_ctx* c) { c->alns.clear(); return THJ_OK; }

extern "C" int thj_span_run_async(thj_ctx* c, const thj_params* tp, const thj_span_batch* db) {
    static_assert(sizeof(orc_span_hit) == sizeof(thj_span_hit), "span hit layouts");
    if (tp->fusion_search) { thj_set_error("cpu port: --fusion-search is not part of this figure"); return THJ_EINVAL; }
    orc_span_params p{};
    p.segment_length = tp->segment_length; p.max_insertion_length = tp->max_insertion_length; p.max_deletion_length = tp->max_deletion_length;
    p.min_report_intron = tp->min_report_intron; p.max_report_intron = tp->max_report_intron; p.max_seg_multihits = tp->max_seg_multihits;
    p.read_mismatches = tp->read_mismatches; p.read_gap_length = tp->read_gap_length; p.read_edit_dist = tp->read_edit_dist;
    p.bowtie2 = tp->bowtie2; p.bowtie2_max_penalty = tp->bowtie2_max_penalty; p.bowtie2_min_penalty = tp->bowtie2_min_penalty;
    p.bowtie2_penalty_for_N = tp->bowtie2_penalty_for_N; p.bowtie2_read_gap_open = tp->bowtie2_read_gap_open;
    p.bowtie2_read_gap_cont = tp->bowtie2_read_gap_cont; p.bowtie2_ref_gap_open = tp->bowtie2_ref_gap_open; p.bowtie2_ref_gap_cont = tp->bowtie2_ref_gap_cont;
    const int64_t n = db->n_reads;
    if (n == 0) return THJ_OK;
    std::string bases; std::vector<int64_t> read_off;
    unpack_reads(n, db->words_per_plane, db->read_planes, db->read_len, bases, read_off);
    std::string quals(bases.size(), '!');
    for (int64_t r = 0; r < n; ++r) memcpy(&quals[(size_t)read_off[(size_t)r]], db->quals + (size_t)r * db->qual_stride, (size_t)db->read_len[r]);
    std::vector<int64_t> seg_off((size_t)n * db->nseg + 1);
    for (size_t i = 0; i < seg_off.size(); ++i) seg_off[i] = db->seg_off[i];
    int T = port_threads();
    if ((int64_t)T > n) T = (int)n;
    std::vector<std::vector<thj_aln>> part((size_t)T);
    std::vector<int> rc((size_t)T, 0);
    const uint32_t idx_base = 0;                    // read_idx = index of the read in its batch (one batch per pass in the executables)
    auto work = [&](int t) {
        const int64_t r0 = n * t / T, r1 = n * (t + 1) / T;
        orc_span_batch b{};
        b.n_reads = (int32_t)(r1 - r0); b.nseg = db->nseg; b.read_off = read_off.data() + r0; b.bases = bases.data(); b.quals = quals.data();
        b.seg_off = seg_off.data() + r0 * db->nseg; b.hits = (const orc_span_hit*)db->hits;
        orc_aln* out = nullptr; int64_t no = 0;
        orc_long_md_reset();
        rc[(size_t)t] = orc_spanning_batch(&p, &c->g, &b, c->span_juncs.data(), (int64_t)c->span_juncs.size(), c->span_ins.data(), (int64_t)c->span_ins.size(), &out, &no);
        if (rc[(size_t)t]) return;
        std::vector<thj_aln>& v = part[(size_t)t];
        v.resize((size_t)no);
        uint32_t prev = 0xFFFFFFFFu; uint16_t order = 0;
        for (int64_t i = 0; i < no; ++i) {
            const orc_aln& a = out[i];
            thj_aln& o = v[(size_t)i];
            memset(&o, 0, sizeof o);
            o.read_idx = idx_base + (uint32_t)(r0 + a.read_idx); o.ref_id = a.ref_id; o.left = a.left;
            o.flags = (uint8_t)((a.antisense ? THJ_HIT_ANTISENSE : 0) | (a.antisense_splice ? THJ_HIT_ANTISENSE_SPLICE : 0));
            o.mismatches = a.mismatches; o.edit_dist = a.edit_dist;
            if (a.n_cigar > 16) { rc[(size_t)t] = -1; break; }
            o.n_cigar = (uint8_t)a.n_cigar;
            memcpy(o.cigar, a.cigar, sizeof(uint32_t) * (size_t)a.n_cigar);
            o.AS = (int16_t)a.AS; o.XM = (uint8_t)a.XM; o.XO = (uint8_t)a.XO; o.XG = (uint8_t)a.XG;
            const size_t ml = a.md[0] == '\x01' ? sizeof o.md + 1 : strnlen(a.md, sizeof a.md);
            if (ml > sizeof o.md) o.md_len = THJ_MD_ON_HOST;          // the executable rebuilds it (thj_md_string), as for a device record
            else { o.md_len = (uint8_t)ml; memcpy(o.md, a.md, ml); }
            order = o.read_idx == prev ? (uint16_t)(order + 1) : 0; prev = o.read_idx; o.order = order;
        }
        orc_free(out);
    };
    if (T > 1) { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
    else work(0);
    for (int t = 0; t < T; ++t) {
        if (rc[(size_t)t]) { thj_set_error("cpu port: orc_spanning_batch failed"); return THJ_ENOMEM; }
        c->alns.insert(c->alns.end(), part[(size_t)t].begin(), part[(size_t)t].end());
    }
    return THJ_OK;
}
extern "C" int thj_span_finish(thj_ctx* c, int64_t* n_alns) { *n_alns = (int64_t)c->alns.size(); return THJ_OK; }
extern "C" int thj_span_download(thj_ctx* c, thj_aln* out) { if (!c->alns.empty()) memcpy(out, c->alns.data(), c->alns.size() * sizeof(thj_aln)); return THJ_OK; }

// ------------------------------------------------------------------------------------------------ the rest
extern "C" void* thj_pinned_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void thj_pinned_free(void* p) { free(p); }
extern "C" void thj_pinned_drain(void) {}
extern "C" void thj_ingest_timing_report(void) {}
extern "C" void thj_cpuport_error(const char* what, int declined) {       // thj_cpuport_stubs.c
    thj_set_error(declined ? "cpu port: %s runs on the host" : "cpu port: %s is not part of the CPU figure", what);
}
