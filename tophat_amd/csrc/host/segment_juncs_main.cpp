// segment_juncs -- MI355X-native drop-in for TopHat's segment_juncs (same argv + files; tophat.py:3097-3112,
// parsed like segment_juncs.cpp:5186-5364).  Host C++ over the C ABI in include/thj.h; all per-read work runs in
// the HIP kernels of libthj_hip.so.  Split-segment search, small indels, the paired-end rescue, --fusion-search and the
// coverage search, the microexon search and the (opt-in) butterfly search are supported.
//
// One process drives every visible GPU.  The reads are cut into contiguous read-id shards with the
// reference's own planner (calculate_offsets over the inputs' .index files, utils.cpp:22-170; segment_juncs.cpp:4756-4810);
// host workers ingest the shards in parallel, shard k runs on GPU k mod n, and when all are done the per-GPU event sets
// are united by ONE RCCL all-gather (thj_events_allgather_async) -- segment_juncs.cpp:4911-4922 across GPUs.  GPU 0's
// sets are written.  The result does not depend on the number of shards or GPUs.
#include <sys/stat.h>
#include "thj_hostio.h"
#include "thj_mx_host.h"

using namespace thjh;

static std::atomic<long long> g_host_ingest_shards{0};      // shards the device-side ingest declined (the host readers took them)
static void print_usage() {
    fprintf(stderr, "Usage:   segment_juncs <ref.fa> <segment.juncs> <segment.insertions> <segment.deletions> <segment.fusions> "
                    "<left_reads.fq> <left_reads.bwtout> <left_seg1.bwtout,...,segN.bwtout> "
                    "[right_reads.fq right_reads.bwtout right_seg1.bwtout,...,right_segN.bwtout]\n");
}

static PhaseTimer g_timer;
static WorkClock g_work;

struct SideInput {
    std::string reads, map;
    std::vector<std::string> segs;
};

// one GPU: its context (created on a side thread while the first shards are parsed) and the lock that serialises the
// device calls of the host workers feeding it
#ifndef THJ_DEFAULT_CTX_PER_GPU
#define THJ_DEFAULT_CTX_PER_GPU 2
#endif
struct Gpu {
    int device = 0;
    thj_ctx* ctx = nullptr;
    std::future<thj_ctx*> fut;
    std::mutex mu;
};

struct Shard {
    uint64_t begin_id = 0, end_id = ~0ull;
    int64_t read_off = 0;
    std::vector<int64_t> seg_off;                   // one per segment map
    int64_t partner_off = 0, seg_partner_off = 0;   // the mate's whole-read map / last segment map
    // where the next shard starts in the same files (-1: this is the last one) -- the device-side ingest hands whole pieces over
    int64_t read_end = -1, partner_end = -1, seg_partner_end = -1;
    std::vector<int64_t> seg_end;
};

// BAM inputs mapped once for the device-side ingest (key: file name); a file that is not a BAM, or whose header does not parse,
// is simply absent and the shards that need it take the host readers
static std::map<std::string, std::unique_ptr<BamFile>> g_bam;
static std::mutex g_stage_mu; static std::condition_variable g_stage_cv; static int g_stage_free = 4;      // STAGE_SLOTS (THJ_STAGE_SLOTS)
static const BamFile* bam_of(const std::string& fn) { auto it = g_bam.find(fn); return it == g_bam.end() ? nullptr : it->second.get(); }

// the reference's shard plan for one side: calculate_offsets over {reads, segment maps} + calculate_offsets_from_ids for the
// mate's two maps (segment_juncs.cpp:4756-4775); one shard when an input has no usable index
static std::vector<Shard> plan_side(const SideInput& in, const SideInput* mate, int want) {
    std::vector<Shard> out(1);
    out[0].seg_off.assign(in.segs.size(), 0);
    if (want < 2 || in.segs.empty()) return out;
    std::vector<IndexList> lists(1 + in.segs.size());
    load_index(in.reads, want * 4, lists[0]);
    for (size_t s = 0; s < in.segs.size(); ++s) load_index(in.segs[s], want * 4, lists[1 + s]);
    std::vector<uint64_t> ids; std::vector<std::vector<int64_t>> offs;
    size_t smallest = ~(size_t)0;
    for (auto& l : lists) smallest = std::min(smallest, l.size());
    if ((size_t)want > smallest) want = (int)smallest;
    if (!calculate_offsets(lists, want, ids, offs)) return out;
    std::vector<int64_t> po, spo;
    if (mate) {
        IndexList l;
        if (!mate->map.empty()) { load_index(mate->map, want * 4, l); calculate_offsets_from_ids(l, ids, po); }
        l.clear();
        if (!mate->segs.empty()) { load_index(mate->segs.back(), want * 4, l); calculate_offsets_from_ids(l, ids, spo); }
    }
    out.assign((size_t)want, Shard());
    for (int i = 0; i < want; ++i) {
        Shard& sh = out[(size_t)i];
        if (i == 0) sh.seg_off.assign(in.segs.size(), 0);
        else {
            sh.begin_id = ids[(size_t)i - 1];
            sh.read_off = offs[(size_t)i - 1][0];
            sh.seg_off.assign(offs[(size_t)i - 1].begin() + 1, offs[(size_t)i - 1].end());
            if (!po.empty()) sh.partner_off = po[(size_t)i - 1];
            if (!spo.empty()) sh.seg_partner_off = spo[(size_t)i - 1];
        }
        sh.end_id = i + 1 < want ? ids[(size_t)i] : ~0ull;
    }
    // ends of the shards' shares of every file (whole pieces go to the device-side ingest)
    IndexList l_full, l_last;
    if (mate && !mate->map.empty()) load_index(mate->map, want * 4, l_full);
    if (mate && !mate->segs.empty()) load_index(mate->segs.back(), want * 4, l_last);
    for (int i = 0; i < want; ++i) {
        Shard& sh = out[(size_t)i];
        sh.seg_end.assign(in.segs.size(), -1);
        if (i + 1 == want) continue;
        sh.read_end = shard_end_offset(lists[0], sh.end_id);
        for (size_t s = 0; s < in.segs.size(); ++s) sh.seg_end[s] = shard_end_offset(lists[1 + s], sh.end_id);
        sh.partner_end = shard_end_offset(l_full, sh.end_id);
        sh.seg_partner_end = shard_end_offset(l_last, sh.end_id);
    }
    return out;
}

// One shard of one side (SegmentSearchWorker, segment_juncs.cpp:4548-4672): walk the nseg id-sorted segment maps in
// increasing id order -- the visiting order of look_for_hit_group (segment_juncs.cpp:3823-4123; derivation in
// tophat_amd/batch.py) -- join the mate's maps by id (find_gaps :3321-3348), batch, run.
// `ordinal` = first-inserted-wins priority of the shard's first read (std::set<Insertion>, insertions.h:52-67): the reference
// inserts all left reads before all right reads and, inside a side, in read-id order (thread sets are merged in thread
// order, :4911-4916).  A shard starts at its begin_id -- ids are distinct and increasing, so begin_id + k never exceeds the
// id of the shard's k-th visited read and every shard's ordinals stay below the next shard's.
static constexpr uint32_t RIGHT_ORDINAL_BASE = 1u << 28;
static void run_shard(const std::function<thj_ctx*(Gpu&)>& device_ready, Gpu& gpu, Opts& o, RefTable& rt, const SideInput& in,
                      const SideInput* mate, int read_side, const Shard& sh, uint32_t ordinal, uint32_t ordinal_limit, size_t batch_reads) {
    const int nseg = (int)in.segs.size();
    // one segment map: no segment search (segment_juncs.cpp:4752 `size() > 1`), but its hits still belong to the coverage
    // map (all_segmap_fnames :4929-4935)
    if (nseg < 1 || (nseg == 1 && !o.cov_state)) return;
    const long long t_shard = WorkClock::now();
    struct AtExit { long long t; ~AtExit() { g_work.add(0, t); } } at_exit{t_shard};
    // ---- device-side ingest (thj_ingest_seg_batch): every input a mapped BAM -> the host only points at compressed bytes
    static const bool host_ingest = getenv("THJ_HOST_INGEST") != nullptr;
    if (!host_ingest) {
        std::vector<thj_bam_piece> segp;
        bool ok = true;
        for (int s = 0; s < nseg && ok; ++s) { const BamFile* bf = bam_of(in.segs[(size_t)s]); if (!bf) ok = false; else segp.push_back(bf->piece(sh.seg_off[(size_t)s], sh.seg_end.empty() ? -1 : sh.seg_end[(size_t)s])); }
        const BamFile* rf = ok ? bam_of(in.reads) : nullptr;
        if (!rf) ok = false;
        thj_bam_piece mf{}, ml{}; bool have_mf = false, have_ml = false;
        if (ok && mate && !mate->segs.empty()) {
            if (!mate->map.empty()) { const BamFile* bf = bam_of(mate->map); if (bf) { mf = bf->piece(sh.partner_off, sh.partner_end); have_mf = true; } else ok = false; }
            const BamFile* bl = ok ? bam_of(mate->segs.back()) : nullptr;
            if (bl) { ml = bl->piece(sh.seg_partner_off, sh.seg_partner_end); have_ml = true; } else ok = false;
        }
        if (ok) {
            thj_bam_piece rp = rf->piece(sh.read_off, sh.read_end);
            thj_params p = o.p;
            p.read_side = read_side;
            const uint32_t b_id = sh.begin_id > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sh.begin_id, e_id = sh.end_id > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sh.end_id;
            // the shard's compressed pieces into one page-locked buffer, outside the GPU's lock (stage_pieces)
            std::vector<std::pair<const BamFile*, thj_bam_piece*>> to_stage;
            for (int s = 0; s < nseg; ++s) to_stage.emplace_back(bam_of(in.segs[(size_t)s]), &segp[(size_t)s]);
            to_stage.emplace_back(rf, &rp);
            if (have_mf) to_stage.emplace_back(bam_of(mate->map), &mf);
            if (have_ml) to_stage.emplace_back(bam_of(mate->segs.back()), &ml);
            // at most STAGE_SLOTS shards hold a buffer at a time (the others wait here instead of at the GPU's lock): locking pages costs
            // 0.2 s per GB and stalls the device calls of the other threads while it runs, so the pool must stay at a handful of blocks
            // that go round -- sixteen workers each locking their own 60 MB made the stage twice as slow as reading the mappings
            struct Staged {
                uint8_t* p = nullptr;
                Staged() { std::unique_lock<std::mutex> lk(g_stage_mu); g_stage_cv.wait(lk, [] { return g_stage_free > 0; }); --g_stage_free; }
                ~Staged() { thj_pinned_free(p); { std::lock_guard<std::mutex> lk(g_stage_mu); ++g_stage_free; } g_stage_cv.notify_one(); }
            } staged;
            staged.p = stage_pieces(to_stage);
            const long long tw = WorkClock::now();
            std::lock_guard<std::mutex> lk(gpu.mu);
            g_work.add(1, tw);
            const long long td = WorkClock::now();
            thj_ctx* ctx = device_ready(gpu);
            thj_seg_batch* dev = nullptr;
            int64_t n = 0;
            const int rc = thj_ingest_seg_batch(ctx, &p, nseg, segp.data(), have_mf ? &mf : nullptr, have_ml ? &ml : nullptr, &rp, b_id, e_id,
                                                (o.fusion_search || o.cov_state) ? 1 : 0, ordinal, &dev, &n);
            if (rc == THJ_OK) {
                if ((uint64_t)ordinal + (uint64_t)n > ordinal_limit)
                    die("Error: too many reads on the %s side for the device's read ordinals (read ids must stay below %u)\n", read_side == 1 ? "left" : "right", RIGHT_ORDINAL_BASE);
                if (dev) {
                    if (nseg > 1 && thj_segjuncs_run_async(ctx, &p, dev)) die("Error: %s\n", thj_last_error());
                    if (nseg > 1 && o.fusion_search && thj_fusion_run_async(ctx, &p, dev)) die("Error: %s\n", thj_last_error());
                    if (o.cov_state && thj_covsearch_add_hits_async(ctx, dev)) die("Error: %s\n", thj_last_error());
                    if (!o.no_microexon_search && thj_microexon_collect(ctx, &p, dev, read_side)) die("Error: %s\n", thj_last_error());
                    if (thj_batch_free(ctx, dev)) die("Error: %s\n", thj_last_error());
                }
                g_work.add(2, td);
                return;
            }
            if (rc != THJ_EFALLBACK) die("Error: %s\n", thj_last_error());
            g_work.add(2, td);
            static std::atomic<bool> told{false};
            g_host_ingest_shards.fetch_add(1);
            if (!told.exchange(true)) fprintf(stderr, "\tdevice-side ingest not possible (%s); reading on the host\n", thj_last_error());
        }
    }
    std::vector<HitStream> st((size_t)nseg);
    for (int s = 0; s < nseg; ++s)
        if (!st[(size_t)s].open(in.segs[(size_t)s], rt, o.p, false, sh.seg_off[(size_t)s], sh.begin_id, sh.end_id))
            die("Error opening SAM file %s\n", in.segs[(size_t)s].c_str());
    HitStream mate_full, mate_last;
    bool have_mate = false;
    if (mate && !mate->segs.empty()) {
        bool a = !mate->map.empty() && mate_full.open(mate->map, rt, o.p, false, sh.partner_off, sh.begin_id, sh.end_id);
        bool b = mate_last.open(mate->segs.back(), rt, o.p, false, sh.seg_partner_off, sh.begin_id, sh.end_id);
        have_mate = a || b;
    }
    ReadStream reads;
    if (!reads.open(in.reads, o.zpacker, sh.read_off)) die("Error: cannot open %s for reading\n", in.reads.c_str());

    thj_params p = o.p;
    p.read_side = read_side;
    std::vector<uint32_t> seg_off, mate_off;
    std::vector<thj_hit> hits, mate_hits;
    std::vector<int64_t> read_off;
    std::string bases;
    size_t max_len = 0;
    auto reset = [&]() { seg_off.assign(1, 0); mate_off.assign(1, 0); hits.clear(); mate_hits.clear(); read_off.assign(1, 0); bases.clear(); max_len = 0; };
    auto flush = [&]() {
        int64_t n = (int64_t)read_off.size() - 1;
        if (n == 0) return;
        if ((uint64_t)ordinal + (uint64_t)n > ordinal_limit)
            die("Error: too many reads on the %s side for the device's read ordinals (read ids must stay below %u)\n", read_side == 1 ? "left" : "right",
                RIGHT_ORDINAL_BASE);
        int W = (int)((max_len + 63) / 64); if (W < 1) W = 1;
        std::vector<uint64_t> planes((size_t)n * 3 * W);
        std::vector<uint16_t> lens((size_t)n);
        if (thj_reads_pack(n, read_off.data(), bases.data(), W, planes.data(), lens.data())) die("Error: %s\n", thj_last_error());
        thj_seg_batch hb{};
        hb.n_reads = (int32_t)n; hb.nseg = nseg; hb.words_per_plane = W;
        hb.seg_off = seg_off.data(); hb.hits = hits.data(); hb.read_planes = planes.data(); hb.read_len = lens.data();
        if (have_mate) { hb.mate_off = mate_off.data(); hb.mate_hits = mate_hits.data(); }
        hb.ordinal_base = ordinal;
        {
            const long long tw = WorkClock::now();
            std::lock_guard<std::mutex> lk(gpu.mu);
            g_work.add(1, tw);
            const long long td = WorkClock::now();
            thj_ctx* ctx = device_ready(gpu);
            thj_seg_batch* dev = nullptr;
            if (thj_batch_upload(ctx, &hb, (int64_t)hits.size(), (int64_t)mate_hits.size(), &dev)) die("Error: %s\n", thj_last_error());
            if (nseg > 1 && thj_segjuncs_run_async(ctx, &p, dev)) die("Error: %s\n", thj_last_error());
            if (nseg > 1 && o.fusion_search && thj_fusion_run_async(ctx, &p, dev)) die("Error: %s\n", thj_last_error());
            if (o.cov_state && thj_covsearch_add_hits_async(ctx, dev)) die("Error: %s\n", thj_last_error());
            if (!o.no_microexon_search && thj_microexon_collect(ctx, &p, dev, read_side)) die("Error: %s\n", thj_last_error());
            if (thj_batch_free(ctx, dev)) die("Error: %s\n", thj_last_error());
            g_work.add(2, td);
        }
        ordinal += (uint32_t)n;
        reset();
    };
    reset();
    std::vector<std::vector<Hit>> grp((size_t)nseg);
    std::vector<Hit> mg;
    for (;;) {
        uint32_t id = 0;
        for (int s = 0; s < nseg; ++s) { uint32_t g = st[(size_t)s].next_group_id(); if (g && (id == 0 || g < id)) id = g; }
        if (id == 0) break;
        int top = -1;
        for (int s = 0; s < nseg; ++s) {
            grp[(size_t)s].clear();
            if (st[(size_t)s].next_group_id() == id) { st[(size_t)s].next_group(grp[(size_t)s]); top = s; }
        }
        // only find_fusions runs for reads whose highest mapped segment is the first (segment_juncs.cpp:3994-4028);
        // they are event-neutral for the gap / indel finders, so with --fusion-search they simply ride along
        // ... and with the coverage search on their hits belong to the coverage map (build_coverage_map :4140-4176 walks
        // every record of every segment map)
        if (top < 0 || (top == 0 && !o.fusion_search && !o.cov_state)) continue;
        Read rd;
        if (!reads.get(id, rd)) die("Error: could not get read# %d from stream!", (int)id);
        for (int s = 0; s < nseg; ++s) { for (auto& h : grp[(size_t)s]) hits.push_back(h.h16); seg_off.push_back((uint32_t)hits.size()); }
        if (have_mate) {
            mg.clear();
            while (mate_full.next_group_id() && mate_full.next_group_id() < id) mate_full.skip_group();
            if (mate_full.next_group_id() == id) mate_full.next_group(mg);
            else {
                while (mate_last.next_group_id() && mate_last.next_group_id() < id) mate_last.skip_group();
                if (mate_last.next_group_id() == id) mate_last.next_group(mg);
            }
            for (auto& h : mg) mate_hits.push_back(h.h16);
            mate_off.push_back((uint32_t)mate_hits.size());
        }
        bases += rd.seq;
        read_off.push_back((int64_t)bases.size());
        if (rd.seq.size() > max_len) max_len = rd.seq.size();
        if (read_off.size() - 1 >= batch_reads) flush();
    }
    flush();
}

static int real_main(int argc, char** argv) {
    if (getenv("THJ_STAGE_SLOTS") && atoi(getenv("THJ_STAGE_SLOTS")) >= 1) g_stage_free = atoi(getenv("THJ_STAGE_SLOTS"));
    fprintf(stderr, "segment_juncs (MI355X-native, %s)\n---------------------------\n", thj_version());
    Opts o;
    int rc = parse_options(argc, argv, o, print_usage);
    if (rc) return rc;
    std::vector<std::string> pos;
    for (int i = optind; i < argc; ++i) pos.push_back(argv[i]);
    if (pos.size() < 8 || (pos.size() > 8 && pos.size() < 11)) { print_usage(); return 1; }
    if (o.color) die("Error: colour-space reads are not supported by this build\n");
    if (o.ium_reads.empty()) { o.no_coverage_search = true; o.butterfly_search = false; }      // no unmapped reads: segment_juncs.cpp:4978-4982
    // the coverage map and the extension table are what both the coverage search and the butterfly search work from (:4955)
    o.cov_state = !o.no_coverage_search || o.butterfly_search;
    SideInput left{pos[5], pos[6], split(pos[7], ',')}, right;
    if (pos.size() >= 11) right = SideInput{pos[8], pos[9], split(pos[10], ',')};
    if (left.segs.empty()) { fprintf(stderr, "No hits to process, exiting\n"); return 0; }      // segment_juncs.cpp:4724-4728

    FILE* fj = fopen(pos[1].c_str(), "w"); if (!fj) die("Error: cannot open %s for writing\n", pos[1].c_str());
    FILE* fi = fopen(pos[2].c_str(), "w"); if (!fi) die("Error: cannot open %s for writing\n", pos[2].c_str());
    FILE* fd = fopen(pos[3].c_str(), "w"); if (!fd) die("Error: cannot open %s for writing\n", pos[3].c_str());
    FILE* ff = fopen(pos[4].c_str(), "w"); if (!ff) die("Error: cannot open %s for writing\n", pos[4].c_str());

    // ---- GPUs: every visible one (THJ_GPUS caps the count, THJ_DEVICE picks a single device).  HIP start-up runs beside the
    // FASTA load and the first shards' ingest: each context is created on its own thread and picked up -- with the genome
    // going up then -- by the first worker that needs the device (under the GPU's lock).
    // the reference is read on its own thread(s) while the HIP runtime starts (the device count below is its first call, ~50 ms)
    RefTable rt;
    rt.load_sam_header(o.sam_header);
    fprintf(stderr, "Loading reference sequences...\n");
    std::future<void> fasta_loaded = std::async(std::launch::async, [&rt, &pos]() { rt.load_reference(pos[0], pos[1]); });
    std::vector<std::unique_ptr<Gpu>> gpus;
    {
        int n_dev = 1, first = 0;
        if (getenv("THJ_DEVICE")) first = atoi(getenv("THJ_DEVICE"));
        else { n_dev = thj_device_count(); if (n_dev < 1) die("Error: %s\n", thj_last_error()); if (getenv("THJ_GPUS") && atoi(getenv("THJ_GPUS")) >= 1) n_dev = std::min(n_dev, atoi(getenv("THJ_GPUS"))); }
        // THJ_CTX_PER_GPU=k: k contexts (streams, arenas, tables) on every device, each a rank of its own -- a shard's host-to-device
        // copies and stream round trips then overlap another shard's kernels on the same GPU
        // (default: 2 on a single GPU -- measured 2.4 -> 2.0 s for segment_juncs on 8 M pairs -- and 1 per device on several: a
        // communicator is either all-RCCL or all-loopback)
        int per = getenv("THJ_CTX_PER_GPU") ? atoi(getenv("THJ_CTX_PER_GPU")) : (n_dev > 1 ? 1 : THJ_DEFAULT_CTX_PER_GPU);
        if (n_dev > 1) per = 1;
        if (per < 1) per = 1;
        if (per > 8) per = 8;
        for (int d = 0; d < n_dev * per; ++d) {
            gpus.emplace_back(new Gpu());
            Gpu& g = *gpus.back();
            g.device = first + d / per;
            g.fut = std::async(std::launch::async, [dev = g.device]() {
                thj_ctx* c = nullptr;
                if (thj_ctx_create(dev, nullptr, &c)) die("Error: %s\n", thj_last_error());
                // (thj_ctx_warm here was measured at nothing: the runtime's start-up on this thread is what the first shard waits for, and
                // the code objects loaded behind it only make that longer; long_spanning_reads, with three of them, gains 0.05 s)
                if (getenv("THJ_WARM") && thj_ctx_warm(c, THJ_WARM_SEGJUNCS | THJ_WARM_INGEST)) die("Error: %s\n", thj_last_error());
                if (getenv("THJ_TIMING")) fprintf(stderr, "[timing] a device context ready after       %8.3f s of the process\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count() - g_timer.wall0);
                return c;
            });
        }
    }
    const int n_gpus = (int)gpus.size();

    fasta_loaded.get();
    for (const SideInput* sd : {&left, &right}) { register_targets(sd->map, rt); for (auto& f : sd->segs) register_targets(f, rt); }
    if (!getenv("THJ_HOST_INGEST"))                                   // map the BAM inputs for the device-side ingest
        for (const SideInput* sd : {&left, &right}) {
            std::vector<std::string> fns = sd->segs;
            fns.push_back(sd->reads); if (!sd->map.empty()) fns.push_back(sd->map);
            for (auto& fn : fns) if (!fn.empty() && !g_bam.count(fn)) { std::unique_ptr<BamFile> bf(new BamFile()); if (bf->open(fn, rt)) g_bam[fn] = std::move(bf); }
        }
    std::vector<uint32_t> fusion_ignore_ids;
    if (o.fusion_search && !o.fusion_ignore.empty())                   // segment_juncs.cpp:3214-3219
        for (auto& nm : split(o.fusion_ignore, ',')) if (!nm.empty()) fusion_ignore_ids.push_back(rt.get_id(nm));
    rt.freeze();
    g_timer.lap("options + reference FASTA");

    std::function<thj_ctx*(Gpu&)> device_ready = [&](Gpu& g) -> thj_ctx* {          // call with g.mu held
        if (g.ctx) return g.ctx;
        g.ctx = g.fut.get();
        rt.upload(g.ctx);
        if (thj_segjuncs_reset_async(g.ctx)) die("Error: %s\n", thj_last_error());
        if (o.fusion_search && thj_fusion_reset_async(g.ctx)) die("Error: %s\n", thj_last_error());
        if (o.cov_state && thj_covsearch_reset_async(g.ctx)) die("Error: %s\n", thj_last_error());
        if (!fusion_ignore_ids.empty() && thj_fusion_set_ignored(g.ctx, fusion_ignore_ids.data(), (int32_t)fusion_ignore_ids.size())) die("Error: %s\n", thj_last_error());
        return g.ctx;
    };
    size_t batch_reads = getenv("THJ_BATCH_READS") ? (size_t)atoll(getenv("THJ_BATCH_READS")) : (size_t)1 << 20;
    // ---- coverage search, part 1 (segment_juncs.cpp:4955-4982): the extension table of the initially unmapped reads
    // (index_read_mers :548-571 -- the first 32 bases of every read) is fed to the device on its own thread while the
    // two sides are being ingested (chunks go round the GPUs; the exchange step concatenates the tables anyway)
    std::thread ium_thread;
    if (o.cov_state)
        ium_thread = std::thread([&]() {
            const size_t CH = (size_t)1 << 19;
            size_t turn = 0;
            for (auto& fn : split(o.ium_reads, ',')) {
                if (fn.empty()) continue;
                // an unaligned BAM (what tophat.py passes): runs of whole BGZF members go up compressed and are inflated and parsed on the
                // device (thj_covsearch_add_reads_bam); the host stream below takes over -- past the records the device took -- when a
                // piece is declined (records that straddle members)
                long long skip = 0;
                {
                    BamFile bf;
                    static const bool host_ingest = getenv("THJ_HOST_INGEST") != nullptr;
                    if (!host_ingest && bf.open(fn, rt)) {
                        const size_t MEMBERS = 8192;                       // 512 MB of inflated records a call at most
                        size_t at = (size_t)(bf.first_rec_voff >> 16);
                        int64_t from = bf.first_rec_voff;
                        bool declined = false, reserved = false;
                        // inflated bytes of the members in [a, e): their ISIZE fields
                        auto inflated = [&bf](size_t a, size_t e) {
                            double t = 0;
                            while (a < e) { const uint32_t bs = BamFile::member_size(bf.data + a, bf.size - a); if (!bs || a + bs > bf.size) break; uint32_t isz; memcpy(&isz, bf.data + a + bs - 4, 4); t += isz; a += bs; }
                            return t;
                        };
                        while (at < bf.size && !declined) {
                            size_t e = at, k = 0;
                            while (e < bf.size && k < MEMBERS) { const uint32_t bs = BamFile::member_size(bf.data + e, bf.size - e); if (!bs) { e = bf.size; break; } e += bs; ++k; }
                            thj_bam_piece pc = bf.piece(from, e < bf.size ? (int64_t)e << 16 : -1);
                            std::vector<std::pair<const BamFile*, thj_bam_piece*>> st{{&bf, &pc}};
                            uint8_t* staged = stage_pieces(st);
                            int64_t n = 0;
                            int rc2;
                            {
                                Gpu& g = *gpus[turn++ % gpus.size()];
                                std::lock_guard<std::mutex> lk(g.mu);
                                rc2 = thj_covsearch_add_reads_bam(device_ready(g), &pc, &n);
                            }
                            thj_pinned_free(staged);
                            if (rc2 == THJ_EFALLBACK) { declined = true; break; }
                            if (rc2) die("Error: %s\n", thj_last_error());
                            skip += n;
                            if (!reserved && n > 0 && e < bf.size) {
                                // the file's first piece says how many reads a byte holds: room for the rest of the file in every context's table now
                                // (the pieces go round the contexts), instead of a table that grows by half -- and is copied -- whenever it is full
                                reserved = true;
                                const double here = inflated(at, e), rest = inflated(e, bf.size);
                                if (here > 0) {
                                    const int64_t per_ctx = (int64_t)(rest * ((double)n / here) * 1.3 / (double)gpus.size()) + 65536;
                                    for (auto& gp : gpus) { std::lock_guard<std::mutex> lk(gp->mu); if (thj_covsearch_reserve_reads(device_ready(*gp), per_ctx)) die("Error: %s\n", thj_last_error()); }
                                }
                            }
                            at = e; from = (int64_t)e << 16;
                        }
                        if (!declined) continue;
                        g_host_ingest_shards.fetch_add(1);
                    }
                }
                ReadStream rs;
                if (!rs.open(fn, o.zpacker)) { fprintf(stderr, "Can't open file %s for reading, skipping...\n", fn.c_str()); continue; }
                std::string bases; std::vector<int64_t> off(1, 0);
                auto push = [&]() {
                    const int64_t n = (int64_t)off.size() - 1;
                    if (!n) return;
                    std::vector<uint64_t> planes((size_t)n * 3); std::vector<uint16_t> lens((size_t)n);
                    if (thj_reads_pack(n, off.data(), bases.data(), 1, planes.data(), lens.data())) die("Error: %s\n", thj_last_error());
                    {
                        Gpu& g = *gpus[turn++ % gpus.size()];
                        std::lock_guard<std::mutex> lk(g.mu);
                        if (thj_covsearch_add_reads(device_ready(g), n, 1, planes.data(), lens.data(), 0)) die("Error: %s\n", thj_last_error());
                    }
                    bases.clear(); off.assign(1, 0);
                };
                Read rd;
                while (rs.next_direct(rd)) {
                    if (skip > 0) { --skip; continue; }
                    if (rd.qc_fail) continue;                 // reads.cpp:556 (the device path gives such a record length 0: thj_k_ium_planes)
                    bases.append(rd.seq, 0, rd.seq.size() < 32 ? rd.seq.size() : 32);      // count_read_mers / store_read_mers :425, :520
                    off.push_back((int64_t)bases.size());
                    if (off.size() - 1 >= CH) push();
                }
                push();
            }
        });
    fprintf(stderr, ">> Performing segment-search:\n");
    // ---- shards and host workers.  THJ_WORKERS host workers (default: half the usable CPUs -- every worker also keeps one reader
    // thread per input file busy) take (side, shard) items; -p N asks for at least N shards per side, as in the reference.
    const int hw = effective_cpus();
    int workers = getenv("THJ_WORKERS") ? atoi(getenv("THJ_WORKERS")) : std::max(1, std::min(32, hw * 3 / 4));
    if (workers < 1) workers = 1;
    int want = getenv("THJ_SHARDS") ? atoi(getenv("THJ_SHARDS")) : std::max(std::max(workers, o.num_threads), n_gpus);
    {   // ... and a shard's inputs stay below 384 MB of BGZF: the device-side ingest addresses a shard's members with 16 bits (~1 GB at
        // BAM's compression) and keeps 64 KiB of inflated bytes and a token stream per member
        auto fsize = [](const std::string& f) -> uint64_t { struct stat st; return (!f.empty() && stat(f.c_str(), &st) == 0) ? (uint64_t)st.st_size : 0; };
        auto side_bytes = [&](const SideInput& a, const SideInput& b) { uint64_t n = fsize(a.reads); for (auto& f : a.segs) n += fsize(f); n += fsize(b.map); if (!b.segs.empty()) n += fsize(b.segs.back()); return n; };
        const uint64_t by_size = (std::max(side_bytes(left, right), side_bytes(right, left)) >> 20) / 384 + 1;
        if (!getenv("THJ_SHARDS") && by_size > (uint64_t)want) want = (int)std::min<uint64_t>(by_size, 1 << 16);
    }
    struct Item { const SideInput* in; const SideInput* mate; int side; Shard sh; uint32_t ordinal, limit; int gpu; };
    std::vector<Item> items;
    {
        const bool paired = !right.segs.empty();
        std::vector<Shard> ls = plan_side(left, paired ? &right : nullptr, want);
        std::vector<Shard> rs = paired ? plan_side(right, &left, want) : std::vector<Shard>();
        // both sides of a read pair have the same id: interleave the sides so that the sets a GPU builds grow evenly
        for (size_t k = 0; k < std::max(ls.size(), rs.size()); ++k) {
            if (k < ls.size()) {
                if (ls[k].begin_id >= RIGHT_ORDINAL_BASE) die("Error: read ids must stay below %u\n", RIGHT_ORDINAL_BASE);
                items.push_back({&left, paired ? &right : nullptr, 1, ls[k], (uint32_t)ls[k].begin_id, paired ? RIGHT_ORDINAL_BASE : (1u << 29) - 1, 0});
            }
            if (k < rs.size()) {
                if (rs[k].begin_id >= RIGHT_ORDINAL_BASE) die("Error: read ids must stay below %u\n", RIGHT_ORDINAL_BASE);
                items.push_back({&right, &left, 2, rs[k], RIGHT_ORDINAL_BASE + (uint32_t)rs[k].begin_id, (1u << 29) - 1, 0});
            }
        }
        for (size_t k = 0; k < items.size(); ++k) items[k].gpu = (int)(k % (size_t)n_gpus);
        fprintf(stderr, "\t%d left + %d right read-id shards, %d host workers, %d GPU%s\n", (int)ls.size(), (int)rs.size(), workers, n_gpus, n_gpus > 1 ? "s" : "");
    }
    {
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= items.size()) return;
                const Item& it = items[k];
                run_shard(device_ready, *gpus[(size_t)it.gpu], o, rt, *it.in, it.mate, it.side, it.sh, it.ordinal, it.limit, batch_reads);
            }
        };
        const int nthr = (int)std::min<size_t>((size_t)workers, items.size());
        std::vector<std::thread> th;
        for (int t = 1; t < nthr; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    }
    for (auto& g : gpus) { std::lock_guard<std::mutex> lk(g->mu); device_ready(*g); }
    g_timer.lap("device start-up + ingest + pack + upload + launch (all shards)");
    fprintf(stderr, "\tshards read on the host because the device-side ingest declined them: %lld\n", g_host_ingest_shards.load());
    if (o.cov_state) ium_thread.join();            // the unmapped reads went up beside the segment search

    // ---- the exchange step and the end of the pass, one host thread per GPU (each rank's collective calls come from its
    // own thread, include/thj.h).  With one GPU the same calls run without a communicator.
    std::vector<thj_comm*> comms((size_t)n_gpus, nullptr);
    if (n_gpus > 1) {
        std::vector<thj_ctx*> cs;
        for (auto& g : gpus) cs.push_back(g->ctx);
        if (thj_comm_create_local(cs.data(), n_gpus, comms.data())) die("Error: %s\n", thj_last_error());
    }
    std::vector<int64_t> n_cov((size_t)n_gpus, -1), n_fus((size_t)n_gpus, 0);
    std::vector<thj_segjuncs_counts> cnt((size_t)n_gpus);
    // microexon search (segment_juncs.cpp:3737-3941): the candidate windows every GPU found among its reads come down, merge on the
    // host in the order the reference visits the reads (add_to_microexon_windows, thj_mx_host.h), and GPU 0 searches the windows
    // before the exchange step spreads what it found
    MxWindows mxw;
    int64_t n_mx = -1;
    if (!o.no_microexon_search) {
        std::vector<thj_mx_cand> all;
        for (auto& g : gpus) {
            thj_mx_cand* h = nullptr; int64_t nc = 0;
            if (thj_microexon_candidates(g->ctx, &h, &nc)) die("Error: %s\n", thj_last_error());
            all.insert(all.end(), h, h + nc);
            free(h);
        }
        mxw = mx_merge_windows(std::move(all));
    }
    auto end_of_pass = [&](int r) {
        thj_ctx* ctx = gpus[(size_t)r]->ctx;
        thj_comm* cm = comms[(size_t)r];
        if (o.cov_state && cm && thj_covsearch_allgather(ctx, cm)) die("Error: %s\n", thj_last_error());
        if (!o.no_coverage_search) {
            if (r == 0) fprintf(stderr, ">> Performing coverage-search:\n");
            int mcl = 20; if (mcl > o.p.segment_length - 2) mcl = o.p.segment_length - 2;          // :62, :5350
            if (thj_covsearch_run_async(ctx, mcl, o.min_coverage_intron, o.max_coverage_intron)) die("Error: %s\n", thj_last_error());
            if (thj_covsearch_finish(ctx, 5000000, &n_cov[(size_t)r])) die("Error: %s\n", thj_last_error());        // max_cov_juncs :56
        }
        if (o.butterfly_search && r == 0) {
            // pair_covered_sites (segment_juncs.cpp:4998-5012) over the gathered coverage map and extension table: one rank does it, the
            // exchange step below hands its junctions to the others
            int64_t n_bf = 0;
            fprintf(stderr, ">> Performing butterfly-search: \n");
            if (thj_butterfly_run(ctx, o.min_coverage_intron, o.max_coverage_intron, 5000000, &n_bf)) die("Error: %s\n", thj_last_error());
            fprintf(stderr, "\tfound %d potential junctions\n", (int)n_bf);
        }
        if (!o.no_microexon_search && r == 0) {
            fprintf(stderr, ">> Performing microexon-search: \n");
            fprintf(stderr, "Aligning %d microexon segments in %lu windows\n", (int)mxw.strs.size(), (unsigned long)mxw.windows.size());
            if (thj_microexon_run(ctx, mxw.windows.data(), (int64_t)mxw.windows.size(), mxw.strs.data(), mxw.str_len.data(), mxw.str_window.data(), (int64_t)mxw.strs.size(),
                                  o.min_coverage_intron, o.p.library_type, 5000000, &n_mx)) die("Error: %s\n", thj_last_error());
            fprintf(stderr, "\tfound %d potential junctions\n", (int)n_mx);
        }
        if (cm && thj_events_allgather_async(ctx, cm)) die("Error: %s\n", thj_last_error());
        if (thj_segjuncs_finish(ctx, &cnt[(size_t)r])) die("Error: %s\n", thj_last_error());
        if (o.fusion_search) {
            if (thj_fusion_finish(ctx, &n_fus[(size_t)r])) die("Error: %s\n", thj_last_error());
            if (cm && thj_fusion_allgather(ctx, cm, &n_fus[(size_t)r])) die("Error: %s\n", thj_last_error());
        }
    };
    {
        std::vector<std::thread> th;
        for (int r = 1; r < n_gpus; ++r) th.emplace_back(end_of_pass, r);
        end_of_pass(0);
        for (auto& t : th) t.join();
    }
    if (!o.no_coverage_search) {
        fprintf(stderr, "\tfound %d potential junctions\n", (int)n_cov[0]);
        g_timer.lap("coverage search (wait for the unmapped reads + device pass)");
    }
    thj_ctx* ctx = gpus[0]->ctx;
    const thj_segjuncs_counts& n = cnt[0];
    std::vector<thj_junction> j((size_t)n.n_juncs + 1), d((size_t)n.n_deletions + 1);
    std::vector<thj_insertion> ins((size_t)n.n_insertions + 1);
    if (thj_segjuncs_download(ctx, j.data(), d.data(), ins.data())) die("Error: %s\n", thj_last_error());
    g_timer.lap("exchange step + device finish + download");
    fprintf(stderr, "\tfound %ld potential split-segment junctions\n", (long)n.n_juncs);
    fprintf(stderr, "\tfound %ld potential small deletions\n", (long)n.n_deletions);
    fprintf(stderr, "\tfound %ld potential small insertions\n", (long)n.n_insertions);
    // writers: segment_juncs.cpp:5035-5095
    for (int64_t i = 0; i < n.n_juncs; ++i)
        fprintf(fj, "%s\t%d\t%d\t%c\n", rt.names[j[(size_t)i].ref_id - 1].c_str(), (int)j[(size_t)i].left, (int)j[(size_t)i].right, j[(size_t)i].antisense ? '-' : '+');
    for (int64_t i = 0; i < n.n_deletions; ++i)
        fprintf(fd, "%s\t%d\t%d\n", rt.names[d[(size_t)i].ref_id - 1].c_str(), (int)d[(size_t)i].left + 1, (int)d[(size_t)i].right);
    for (int64_t i = 0; i < n.n_insertions; ++i)
        fprintf(fi, "%s\t%d\t%d\t%s\n", rt.names[ins[(size_t)i].ref_id - 1].c_str(), (int)ins[(size_t)i].left, (int)ins[(size_t)i].left, ins[(size_t)i].seq);
    if (o.fusion_search) {
        // fusion writer with its neighbour filter (segment_juncs.cpp:5048-5054, :5096-5182)
        const int64_t nf = n_fus[0];
        std::vector<thj_fusion> f((size_t)nf + 1);
        if (thj_fusion_download(ctx, f.data())) die("Error: %s\n", thj_last_error());
        std::vector<std::pair<uint32_t, int>> coords;          // SpliceJunctionCoord(refid, coord)
        for (int64_t i = 0; i < n.n_juncs; ++i) {
            coords.emplace_back(j[(size_t)i].ref_id, (int)j[(size_t)i].left);
            coords.emplace_back(j[(size_t)i].ref_id, (int)j[(size_t)i].right);
        }
        std::sort(coords.begin(), coords.end());
        std::vector<char> lc((size_t)nf + 1, 0), rc2((size_t)nf + 1, 0), skip((size_t)nf + 1, 0);
        for (int64_t i = 0; i < nf; ++i) {
            lc[(size_t)i] = std::binary_search(coords.begin(), coords.end(), std::make_pair(f[(size_t)i].ref_id1, (int)f[(size_t)i].left));
            rc2[(size_t)i] = std::binary_search(coords.begin(), coords.end(), std::make_pair(f[(size_t)i].ref_id2, (int)f[(size_t)i].right));
        }
        for (int64_t i = 0; i < nf; ++i) {
            const thj_fusion& a = f[(size_t)i];
            for (int64_t k = i + 1; k < nf; ++k) {
                const thj_fusion& b = f[(size_t)k];
                int left_diff = abs((int)a.left - (int)b.left);
                if (!(a.ref_id1 == b.ref_id1 && a.ref_id2 == b.ref_id2 && left_diff < 10)) break;
                if (a.dir == b.dir && left_diff == abs((int)a.right - (int)b.right)) {
                    if (b.count > a.count) skip[(size_t)i] = 1;
                    else if (b.count == a.count) {
                        int cc = lc[(size_t)i] + rc2[(size_t)i], nc = lc[(size_t)k] + rc2[(size_t)k];
                        if (cc < nc) skip[(size_t)i] = 1; else skip[(size_t)k] = 1;
                    } else skip[(size_t)k] = 1;
                }
            }
            if (skip[(size_t)i] && !o.fusion_do_not_resolve_conflicts) continue;
            const char* dir = a.dir == THJ_FUSION_FR ? "fr" : a.dir == THJ_FUSION_RF ? "rf" : a.dir == THJ_FUSION_RR ? "rr" : "ff";
            fprintf(ff, "%s\t%d\t%s\t%d\t%s\n", rt.names[a.ref_id1 - 1].c_str(), (int)a.left, rt.names[a.ref_id2 - 1].c_str(), (int)a.right, dir);
        }
    }
    close_output(fj, "the junctions file"); close_output(fi, "the insertions file"); close_output(fd, "the deletions file"); close_output(ff, "the fusions file");
    fprintf(stderr, "Reported %d total potential splices\n", (int)n.n_juncs);
    g_timer.lap("write outputs");
    g_timer.report();
    thj_ingest_timing_report();
    { static const char* const nm[4] = {"shards (ingest + merge + pack + device)", "  waiting for the GPU's lock", "  device calls (upload, launch, free)", "  -"}; g_work.report(nm); }
    // Everything is written and closed.  Leave without running the exit handlers or freeing the contexts: tearing the HIP
    // runtime (and RCCL) down after use takes tenths of a second that nobody is waiting for.
    rt.finish_cache();                 // (the packed-genome cache's writer, when this process was the one to pack the reference)
    finish_outputs_complete(0);
}

int main(int argc, char** argv) { return run_with_handoff(argc, argv, real_main); }
