// long_spanning_reads -- MI355X-native drop-in for TopHat's long_spanning_reads (same argv + files;
// tophat.py:3160-3191, parsed like long_spanning_reads.cpp:3151-3329).  Host C++ over include/thj.h.
// Contig segment maps (with any CIGAR, incl. N/I/D) are supported; junction-db ("spliced") segment maps
// (SplicedBAMHitFactory) and --fusion-search are refused loudly.
#include "thj_hostio.h"

using namespace thjh;

static void print_usage() {
    fprintf(stderr, "Usage:   long_spanning_reads <reference.fasta> <reads.fq> <possible_juncs1,...,possible_juncsN> "
                    "<possible_insertions1,...,possible_insertionsN> <possible_deletions1,...,possible_deletionsN> "
                    "<possible_fusions1,...,possible_fusionsN> <out.bam> <seg1.bwtout,...,segN.bwtout> [spliced_seg1.bwtout,...,spliced_segN.bwtout]\n");
}

static const char OPCH[16] = {'?', 'M', 'm', 'I', 'i', 'D', 'd', '?', '?', '?', '?', 'N', 'n', 'S', 'H', 'P'};

static PhaseTimer g_timer;

int main(int argc, char** argv) {
    fprintf(stderr, "long_spanning_reads (MI355X-native, %s)\n--------------------------------------------\n", thj_version());
    Opts o;
    int rc = parse_options(argc, argv, o, print_usage);
    if (rc) return rc;
    std::vector<std::string> pos;
    for (int i = optind; i < argc; ++i) pos.push_back(argv[i]);
    if (pos.size() < 8) { print_usage(); return 1; }
    if (o.color) die("Error: colour-space reads are not supported by this build\n");
    if (o.fusion_search) die("Error: --fusion-search is not supported by this build yet\n");
    std::vector<std::string> spliced_segs;
    if (pos.size() >= 9) spliced_segs = split(pos[8], ',');
    std::vector<std::string> segs = split(pos[7], ',');
    if (segs.empty()) { fprintf(stderr, "No hits to process, exiting\n"); return 0; }           // long_spanning_reads.cpp:2883-2887

    RefTable rt;
    rt.load_sam_header(o.sam_header);
    fprintf(stderr, "Loading reference sequences...\n");
    rt.load_fasta(pos[0]);
    fprintf(stderr, "        reference sequences loaded.\n");
    g_timer.lap("options + reference FASTA");

    // ---- junctions + deletions -> std::set<Junction> (long_spanning_reads.cpp:2897-2944)
    std::vector<thj_junction> juncs;
    for (auto& fn : split(pos[2], ',')) {
        FILE* f = fopen(fn.c_str(), "r");
        if (!f) { fprintf(stderr, "Warning: cannot open %s\n", fn.c_str()); continue; }   // :3245-3251
        char buf[2048];
        while (fgets(buf, sizeof buf, f)) {
            char name[256]; int l, r; char ori;
            if (sscanf(buf, "%255s %d %d %c", name, &l, &r, &ori) != 4) continue;
            juncs.push_back({rt.get_id(name), (uint32_t)l, (uint32_t)r, ori == '-' ? 1u : 0u});
        }
        fclose(f);
    }
    for (auto& fn : split(pos[4], ',')) {
        FILE* f = fopen(fn.c_str(), "r");
        if (!f) continue;
        char buf[2048];
        while (fgets(buf, sizeof buf, f)) {
            char* nl = strrchr(buf, '\n'); if (nl) *nl = 0;
            std::vector<std::string> t = split(buf, '\t');
            if (t.size() < 3) die("Error: malformed deletion coordinate record\n");
            juncs.push_back({rt.get_id(t[0]), (uint32_t)atoi(t[1].c_str()) - 1u, (uint32_t)atoi(t[2].c_str()), 0u});
        }
        fclose(f);
    }
    auto jl = [](const thj_junction& a, const thj_junction& b) {       // junctions.h:39-57
        if (a.ref_id != b.ref_id) return a.ref_id < b.ref_id;
        if (a.left != b.left) return a.left < b.left;
        if (a.right != b.right) return a.right < b.right;
        return a.antisense < b.antisense;
    };
    std::sort(juncs.begin(), juncs.end(), jl);
    juncs.erase(std::unique(juncs.begin(), juncs.end(), [&](const thj_junction& a, const thj_junction& b) { return !jl(a, b) && !jl(b, a); }), juncs.end());
    // ---- insertions -> std::set<Insertion>: first inserted wins among equal (ref,left,len) (:2952-2980, insertions.h:52-67)
    struct InsRow { uint32_t ref, left, len, seq; size_t order; };
    std::vector<InsRow> ins;
    for (auto& fn : split(pos[3], ',')) {
        FILE* f = fopen(fn.c_str(), "r");
        if (!f) continue;
        char buf[2048];
        while (fgets(buf, sizeof buf, f)) {
            char* nl = strrchr(buf, '\n'); if (nl) *nl = 0;
            std::vector<std::string> t = split(buf, '\t');
            if (t.size() < 4) die("Error: malformed insertion coordinate record\n");
            uint32_t code = 0;
            if (t[3].size() > 6) die("Error: insertion longer than 6 bases is not supported by this build\n");
            for (size_t k = 0; k < t[3].size(); ++k) {
                uint32_t c = 4;
                switch (t[3][k]) { case 'A': case 'a': c = 0; break; case 'C': case 'c': c = 1; break; case 'G': case 'g': c = 2; break; case 'T': case 't': c = 3; break; }
                code |= c << (3 * k);
            }
            ins.push_back({rt.get_id(t[0]), (uint32_t)atoi(t[1].c_str()), (uint32_t)t[3].size(), code, ins.size()});
        }
        fclose(f);
    }
    std::stable_sort(ins.begin(), ins.end(), [](const InsRow& a, const InsRow& b) {
        if (a.ref != b.ref) return a.ref < b.ref;
        if (a.left != b.left) return a.left < b.left;
        return a.len < b.len;
    });
    std::vector<uint32_t> ins_tab;
    for (size_t i = 0; i < ins.size(); ++i) {
        if (i && ins[i].ref == ins[i - 1].ref && ins[i].left == ins[i - 1].left && ins[i].len == ins[i - 1].len) continue;
        ins_tab.insert(ins_tab.end(), {ins[i].ref, ins[i].left, ins[i].len, ins[i].seq});
    }

    g_timer.lap("junction / indel lists");
    // HIP start-up (0.15-0.25 s) runs beside the first batch's ingest: the context is created on its own thread and
    // picked up -- with the genome and the sets going up then -- the first time the device is needed.
    int device = getenv("THJ_DEVICE") ? atoi(getenv("THJ_DEVICE")) : 0;
    thj_ctx* ctx = nullptr;
    std::future<thj_ctx*> ctx_future = std::async(std::launch::async, [device]() {
        thj_ctx* c = nullptr;
        if (thj_ctx_create(device, nullptr, &c)) die("Error: %s\n", thj_last_error());
        return c;
    });
    auto ensure_device = [&]() {
        if (ctx) return;
        ctx = ctx_future.get();
        g_timer.lap("device context (what was not hidden by ingest)");
        rt.upload(ctx);
        g_timer.lap("genome pack + upload");
        // junctions on contigs the device genome does not know cannot be closed anyway: drop them
        std::vector<thj_junction> keep;
        for (auto& j : juncs) if (j.ref_id <= rt.names.size() && (j.right - j.left) < (1u << 29)) keep.push_back(j);
        juncs.swap(keep);
        if (thj_span_sets_upload(ctx, juncs.data(), (int64_t)juncs.size(), ins_tab.data(), (int64_t)ins_tab.size() / 4)) die("Error: %s\n", thj_last_error());
        if (thj_span_reset_async(ctx)) die("Error: %s\n", thj_last_error());
    };

    const int nseg = (int)segs.size();
    std::vector<HitStream> st((size_t)nseg);
    for (int s = 0; s < nseg; ++s)
        if (!st[(size_t)s].open(segs[(size_t)s], rt, o.p)) die("Error opening SAM file %s\n", segs[(size_t)s].c_str());
    // junction-db ("spliced") segment maps: SplicedBAMHitFactory streams, one per segment (:3110-3123)
    std::vector<HitStream> sst(spliced_segs.size());
    for (size_t s = 0; s < spliced_segs.size(); ++s)
        if (!sst[s].open(spliced_segs[s], rt, o.p, true)) die("Error opening SAM file %s\n", spliced_segs[s].c_str());
    ReadStream reads;
    if (!reads.open(pos[1], o.zpacker)) die("Error: cannot open %s for reading\n", pos[1].c_str());

    // ---- BAM output (print_bamhit, bwt_map.cpp:1888-2093).  One output file: every batch's records are written as soon
    // as they come back (worker threads encode and deflate).  -p N => <base>{0..N-1}.bam (long_spanning_reads.cpp:
    // 3056-3064): the records are kept and cut into N files at read boundaries at the end.
    const std::string out = pos[6];
    const int parts = o.num_threads > 1 ? o.num_threads : 1;
    BamWriter bw1;
    if (parts == 1 && !bw1.open(out, rt, out + ".index")) die("Error: could not create BAM file %s!\n", out.c_str());
    std::vector<Read> all_reads;                 // reads of the current batch (parts == 1) or of the run (parts > 1)
    std::vector<thj_aln> alns;
    auto write_range = [&](BamWriter& bw, const std::vector<thj_aln>& alns, const std::vector<Read>& all_reads, size_t a0, size_t a1) {
        bw.write_records(a1 - a0, [&](size_t i, std::vector<uint8_t>& d) -> long {
            const thj_aln& a = alns[a0 + i];
            const Read& rd = all_reads[a.read_idx];
            int rlen = 0, indel = 0; bool spliced = false;
            for (int k = 0; k < a.n_cigar; ++k) {
                uint32_t op = a.cigar[k] >> 28, len = a.cigar[k] & 0x0FFFFFFF;
                if (op == 1 || op == 2 || op == 3 || op == 4 || op == 13) rlen += (int)len;
                if (op >= 3 && op <= 6) indel += (int)len;
                if (op == 11 || op == 12) spliced = true;
            }
            std::string seq = rd.seq, qual = rd.qual;
            seq.resize((size_t)rlen); qual.resize((size_t)rlen);
            uint32_t flag = 0;
            if (a.flags & THJ_HIT_ANTISENSE) { flag |= 0x10; reverse_complement(seq); std::reverse(qual.begin(), qual.end()); }
            std::vector<std::string> aux;
            aux.push_back("AS:i:" + std::to_string((int)a.AS));
            aux.push_back("XM:i:" + std::to_string((int)a.XM));
            aux.push_back("XO:i:" + std::to_string((int)a.XO));
            aux.push_back("XG:i:" + std::to_string((int)a.XG));
            aux.push_back("MD:Z:" + std::string(a.md, a.md_len));
            aux.push_back("NM:i:" + std::to_string((int)a.mismatches + indel));
            if (spliced) aux.push_back(std::string("XS:A:") + ((a.flags & THJ_HIT_ANTISENSE_SPLICE) ? '-' : '+'));
            bw.encode(d, rd.name, flag, rt.names[a.ref_id - 1], a.left + 1, a.cigar, a.n_cigar, seq, qual, aux);
            return atol(rd.name.c_str());
        });
    };

    // one output file: a writer thread encodes / deflates / writes batch k while the main thread ingests batch k+1
    std::thread writer;
    std::vector<Read> w_reads; std::vector<thj_aln> w_alns;
    auto writer_join = [&]() { if (writer.joinable()) writer.join(); };

    size_t batch_reads = getenv("THJ_BATCH_READS") ? (size_t)atoll(getenv("THJ_BATCH_READS")) : (size_t)1 << 19;
    std::vector<uint32_t> seg_off; std::vector<thj_span_hit> hits; std::vector<int64_t> read_off; std::string bases, quals;
    size_t max_len = 0; size_t batch_first = 0;
    auto reset = [&]() { seg_off.assign(1, 0); hits.clear(); read_off.assign(1, 0); bases.clear(); quals.clear(); max_len = 0; batch_first = all_reads.size(); };
    auto flush = [&]() {
        int64_t n = (int64_t)read_off.size() - 1;
        if (n == 0) return;
        g_timer.lap("ingest (parse + merge by id)");
        ensure_device();
        int W = (int)((max_len + 63) / 64); if (W < 1) W = 1;
        std::vector<uint64_t> planes((size_t)n * 3 * W);
        std::vector<uint16_t> lens((size_t)n);
        if (thj_reads_pack(n, read_off.data(), bases.data(), W, planes.data(), lens.data())) die("Error: %s\n", thj_last_error());
        int stride = (int)((max_len + 3) / 4 * 4);
        std::vector<uint8_t> q((size_t)n * stride, 0);
        for (int64_t r = 0; r < n; ++r) memcpy(q.data() + (size_t)r * stride, quals.data() + read_off[(size_t)r], (size_t)(read_off[(size_t)r + 1] - read_off[(size_t)r]));
        thj_span_batch hb{};
        hb.n_reads = (int32_t)n; hb.nseg = nseg; hb.words_per_plane = W; hb.qual_stride = stride;
        hb.seg_off = seg_off.data(); hb.hits = hits.data(); hb.read_planes = planes.data(); hb.read_len = lens.data(); hb.quals = q.data();
        thj_span_batch* dev = nullptr;
        if (thj_span_batch_upload(ctx, &hb, (int64_t)hits.size(), &dev)) die("Error: %s\n", thj_last_error());
        if (thj_span_reset_async(ctx)) die("Error: %s\n", thj_last_error());
        if (thj_span_run_async(ctx, &o.p, dev)) die("Error: %s\n", thj_last_error());
        int64_t na = 0;
        if (thj_span_finish(ctx, &na)) die("Error: %s\n", thj_last_error());
        size_t base = alns.size();
        alns.resize(base + (size_t)na);
        if (na && thj_span_download(ctx, alns.data() + base)) die("Error: %s\n", thj_last_error());
        for (size_t k = base; k < alns.size(); ++k) alns[k].read_idx += (uint32_t)batch_first;      // -> index into all_reads
        if (thj_span_batch_free(ctx, dev)) die("Error: %s\n", thj_last_error());
        g_timer.lap("pack + upload + stitch + download");
        if (parts == 1) {
            writer_join();                          // batch k-1 is on disk (time spent here = the writer is the bottleneck)
            w_alns.swap(alns); w_reads.swap(all_reads);
            alns.clear(); all_reads.clear();
            writer = std::thread([&]() { write_range(bw1, w_alns, w_reads, 0, w_alns.size()); });
            g_timer.lap("wait for the BAM writer thread");
        }
        reset();
    };
    reset();
    // the worker iterates over first-segment groups (long_spanning_reads.cpp:2706-2765); segments to the right are
    // looked up by id (look_right_for_hit_group :87-163; the kernel stops at the first empty segment as it does)
    std::vector<Hit> g;
    for (;;) {
        // first-segment groups of the contig and the spliced stream, merged by id (:2706-2765)
        uint32_t id = st[0].next_group_id();
        if (!sst.empty()) { uint32_t sid = sst[0].next_group_id(); if (sid && (id == 0 || sid < id)) id = sid; }
        if (id == 0) break;
        Read rd;
        if (!reads.get(id, rd)) die("Error: could not get read # %d from stream\n", (int)id);
        for (int s = 0; s < nseg; ++s) {
            g.clear();
            if (s > 0) while (st[(size_t)s].next_group_id() && st[(size_t)s].next_group_id() < id) st[(size_t)s].skip_group();
            if (st[(size_t)s].next_group_id() == id) st[(size_t)s].next_group(g);
            if ((size_t)s < sst.size()) {            // spliced hits are appended after the contig hits (:125-147, :2738-2744)
                while (sst[(size_t)s].next_group_id() && sst[(size_t)s].next_group_id() < id) sst[(size_t)s].skip_group();
                if (sst[(size_t)s].next_group_id() == id) sst[(size_t)s].next_group(g);
            }
            for (auto& h : g) hits.push_back(h.h32);
            seg_off.push_back((uint32_t)hits.size());
        }
        bases += rd.seq; quals += rd.qual;
        read_off.push_back((int64_t)bases.size());
        if (rd.seq.size() > max_len) max_len = rd.seq.size();
        all_reads.push_back(std::move(rd));
        if (read_off.size() - 1 >= batch_reads) flush();
    }
    flush();
    g_timer.lap("ingest (parse + merge by id)");

    if (parts == 1) { writer_join(); bw1.close(); g_timer.lap("wait for the BAM writer thread"); }
    else {
        std::vector<size_t> cut((size_t)parts + 1, alns.size());
        cut[0] = 0;
        for (int k = 1; k < parts; ++k) {
            size_t c = alns.size() * (size_t)k / (size_t)parts;
            while (c > 0 && c < alns.size() && alns[c].read_idx == alns[c - 1].read_idx) ++c;      // never split a read
            cut[(size_t)k] = c;
        }
        for (int k = 0; k < parts; ++k) {
            std::string fn = out.substr(0, out.size() >= 4 ? out.size() - 4 : out.size()) + std::to_string(k) + ".bam";
            BamWriter bw;
            if (!bw.open(fn, rt, fn + ".index")) die("Error: could not create BAM file %s!\n", fn.c_str());
            write_range(bw, alns, all_reads, cut[(size_t)k], cut[(size_t)k + 1]);
            bw.close();
        }
        g_timer.lap("BAM output (encode + BGZF)");
    }
    if (!ctx) ctx = ctx_future.get();             // nothing to process: the context was never needed
    thj_ctx_destroy(ctx);
    g_timer.lap("teardown");
    g_timer.report();
    (void)OPCH;
    // Everything is written and closed.  Leave without running the exit handlers: tearing the HIP runtime down after a
    // context has been used takes ~0.2 s that nobody is waiting for.
    fflush(nullptr);
    _exit(0);
}
