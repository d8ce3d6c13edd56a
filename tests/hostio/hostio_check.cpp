// hostio_check.cpp -- TEST-ONLY: exercises the host I/O layer of the drop-in executables (tophat_amd/csrc/host/
// thj_hostio.h) without a GPU: the threaded BAM writer and the threaded record readers.
#include "../../tophat_amd/csrc/host/thj_hostio.h"

using namespace thjh;

static uint64_t mix(uint64_t& s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::string mode = argv[1];
    if (mode == "write" && argc >= 5) {
        // hostio_check write <out.bam> <n_records> <batch> [planned]
        // planned: the batches go through BamWriter::plan / compress / commit (all planned first, deflated in reverse order, committed in order)
        // planned-device[-fail]: as planned, and every other batch arrives the way the device path delivers it -- members deflated elsewhere
        // (here: zlib), cut by plan_cuts_closed, no record bytes (BamWriter::plan_device / wrap_member)
        const std::string variant = argc >= 6 ? argv[5] : "";
        const bool device = variant == "planned-device" || variant == "planned-device-fail";
        const bool fail = variant == "planned-fail" || variant == "planned-device-fail";      // a batch in the middle reports a member that did not fit
        const bool planned = fail || device || variant == "planned";
        size_t n_batches = 0;
        std::vector<BamWriter::Prepared> prepared; std::vector<uint8_t> carry;
        const size_t n = (size_t)atoll(argv[3]), batch = (size_t)atoll(argv[4]);
        RefTable rt;
        rt.header_text = "@HD\tVN:1.0\tSO:unsorted\n@SQ\tSN:chr1\tLN:5000000\n@SQ\tSN:chr2\tLN:700000\n";
        rt.sq = {{"chr1", 5000000u}, {"chr2", 700000u}};
        rt.get_id("chr1"); rt.get_id("chr2");
        BamWriter bw;
        std::string out = argv[2];
        if (!bw.open(out, rt, out + ".index")) return 3;
        struct Rec { std::string name, seq, qual; uint32_t flag; int ref, pos; std::vector<uint32_t> cig; int as; };
        std::vector<Rec> recs;
        uint64_t s = 88172645463325252ull;
        auto flush = [&]() {
            if (planned) {
                BamWriter::Encoded e;
                for (const Rec& r : recs) {
                    std::vector<std::string> aux = {"AS:i:" + std::to_string(r.as), "XM:i:1", "MD:Z:" + std::to_string(r.seq.size()), "NM:i:300"};
                    if (r.cig.size() > 1) aux.push_back("XS:A:+");
                    const size_t before = e.bytes.size();
                    bw.encode(e.bytes, r.name, r.flag, rt.names[(size_t)r.ref], r.pos, r.cig.data(), (int)r.cig.size(), r.seq, r.qual, aux);
                    e.size.push_back((uint32_t)(e.bytes.size() - before)); e.rid.push_back(atol(r.name.c_str()));
                }
                prepared.emplace_back();
                if (device && (n_batches++ & 1)) {
                    BamWriter::Prepared& p = prepared.back();
                    p.device = true;
                    BamWriter::plan_cuts_closed(e.size, p.cuts);
                    p.members.resize(p.cuts.size());
                    for (size_t m = 0; m < p.cuts.size(); ++m) {
                        const size_t a = m ? p.cuts[m - 1] : 0, len = p.cuts[m] - a;
                        std::vector<uint8_t> c(len + 1024);
                        z_stream zs; memset(&zs, 0, sizeof zs);
                        deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
                        zs.next_in = e.bytes.data() + a; zs.avail_in = (uInt)len; zs.next_out = c.data(); zs.avail_out = (uInt)c.size();
                        if (deflate(&zs, Z_FINISH) != Z_STREAM_END) return;
                        const size_t clen = zs.total_out;
                        deflateEnd(&zs);
                        BamWriter::wrap_member(c.data(), clen, (uint32_t)crc32(crc32(0L, Z_NULL, 0), e.bytes.data() + a, (uInt)len), (uint32_t)len, p.members[m]);
                    }
                    p.size = std::move(e.size); p.rid = std::move(e.rid);
                    BamWriter::plan_device(carry, p);
                } else
                BamWriter::plan(carry, std::move(e), prepared.back());
                recs.clear();
                return;
            }
            bw.write_records(recs.size(), [&](size_t i, std::vector<uint8_t>& d) -> long {
                const Rec& r = recs[i];
                std::vector<std::string> aux = {"AS:i:" + std::to_string(r.as), "XM:i:1", "MD:Z:" + std::to_string(r.seq.size()), "NM:i:300"};
                if (r.cig.size() > 1) aux.push_back("XS:A:+");
                bw.encode(d, r.name, r.flag, rt.names[(size_t)r.ref], r.pos, r.cig.data(), (int)r.cig.size(), r.seq, r.qual, aux);
                return atol(r.name.c_str());
            });
            recs.clear();
        };
        for (size_t i = 0; i < n; ++i) {
            Rec r;
            r.name = std::to_string(1 + (i / 2) * 3);             // every id twice: the index rule needs an id CHANGE
            int len = 50 + (int)(mix(s) % 101);
            r.seq.resize((size_t)len); r.qual.resize((size_t)len);
            // half of the records carry incompressible-looking qualities, half very regular ones
            for (int k = 0; k < len; ++k) { r.seq[(size_t)k] = "ACGTN"[mix(s) % 5]; r.qual[(size_t)k] = (char)(33 + ((i & 1) ? mix(s) % 42 : 40)); }
            r.flag = (mix(s) & 1) ? 16u : 0u;
            r.ref = (int)(mix(s) % 2);
            r.pos = 1 + (int)(mix(s) % 600000);
            if (mix(s) % 3 == 0) { int a = 10 + (int)(mix(s) % (uint64_t)(len - 20)); r.cig = {(1u << 28) | (uint32_t)a, (11u << 28) | (uint32_t)(50 + mix(s) % 5000), (1u << 28) | (uint32_t)(len - a)}; }
            else r.cig = {(1u << 28) | (uint32_t)len};
            r.as = -(int)(mix(s) % 40000);
            recs.push_back(std::move(r));
            if (recs.size() >= batch) flush();
        }
        flush();
        for (size_t k = prepared.size(); k-- > 0;) BamWriter::compress(prepared[k]);
        if (fail && prepared.size() > 2) prepared[prepared.size() / 2].ok = false;
        for (auto& pr : prepared) bw.commit(pr);
        bw.close();
        return 0;
    }
    if (mode == "sam2bam" && argc >= 5) {
        // hostio_check sam2bam <hdr.sam> <records.sam> <out.bam>: every line of records.sam (QNAME FLAG RNAME POS MAPQ CIGAR SEQ QUAL tags...,
        // the trimmed form of tests/golden/*/expected.span_*.sam) through BamWriter::encode, the encoder of the executables' general path
        RefTable rt;
        rt.load_sam_header(argv[2]);
        BamWriter bw;
        if (!bw.open(argv[4], rt, std::string(argv[4]) + ".index")) return 3;
        std::vector<std::vector<std::string>> lines;
        {
            FILE* f = fopen(argv[3], "r"); if (!f) return 3;
            char* ln = nullptr; size_t cap = 0; ssize_t n;
            while ((n = getline(&ln, &cap, f)) > 0) { std::string l(ln, (size_t)n); while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back(); if (!l.empty() && l[0] != '@') lines.push_back(split(l, '\t')); }
            free(ln); fclose(f);
        }
        bw.write_records(lines.size(), [&](size_t i, std::vector<uint8_t>& d) -> long {
            const auto& c = lines[i];
            std::vector<uint32_t> cig;
            for (size_t k = 0; k < c[5].size();) { uint32_t v = 0; while (k < c[5].size() && isdigit((unsigned char)c[5][k])) v = v * 10 + (uint32_t)(c[5][k++] - '0');
                const char o = c[5][k++]; const uint32_t op = o == 'M' ? 1u : o == 'I' ? 3u : o == 'D' ? 5u : o == 'N' ? 11u : o == 'S' ? 13u : 15u; cig.push_back(op << 28 | v); }
            std::vector<std::string> aux(c.begin() + 8, c.end());
            bw.encode(d, c[0], (uint32_t)atoi(c[1].c_str()), c[2], atoi(c[3].c_str()), cig.data(), (int)cig.size(), c[6], c[7], aux);
            return atol(c[0].c_str());
        });
        bw.close();
        return 0;
    }
    if (mode == "fdz" && argc >= 4) {
        // hostio_check fdz <in> <out> [bench]: <in> = (u32 length, bytes)*; <out> = (u32 compressed length | 0xFFFFFFFF = declined, bytes)*
        // from thj_fastdeflate.h with the room a BGZF member has; bench: MB/s over ten passes on stderr
        FILE* fi = fopen(argv[2], "rb"); FILE* fo = fopen(argv[3], "wb");
        if (!fi || !fo) return 3;
        std::vector<std::vector<uint8_t>> ins;
        for (;;) { uint32_t n; if (fread(&n, 4, 1, fi) != 1) break; std::vector<uint8_t> b(n); if (n && fread(b.data(), 1, n, fi) != n) return 4; ins.push_back(std::move(b)); }
        std::vector<uint8_t> out(70000);
        size_t tin = 0, tout = 0;
        for (auto& b : ins) {
            size_t cl = 0;
            const bool ok = fdz::deflate_fast(b.data(), b.size(), out.data(), 65536 - 18 - 8, &cl);
            uint32_t w = ok ? (uint32_t)cl : 0xFFFFFFFFu;
            fwrite(&w, 4, 1, fo);
            if (ok) { fwrite(out.data(), 1, cl, fo); tin += b.size(); tout += cl; }
        }
        fclose(fi); fclose(fo);
        if (argc >= 5) {
            auto t0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 10; ++rep) for (auto& b : ins) { size_t cl = 0; fdz::deflate_fast(b.data(), b.size(), out.data(), 65536 - 18 - 8, &cl); }
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            fprintf(stderr, "fdz: %.1f MB/s, ratio %.3f\n", 10.0 * (double)tin / dt / 1e6, tin ? (double)tout / (double)tin : 0.0);
        }
        return 0;
    }
    if (mode == "hits" && argc >= 3) {
        // hostio_check hits <map.sam|bam>  -> "<groups> <hits> <checksum>"
        RefTable rt;
        thj_params p;
        thj_params_default(&p);
        HitStream hs;
        if (!hs.open(argv[2], rt, p)) return 3;
        std::vector<Hit> g;
        uint64_t groups = 0, hits = 0, sum = 0;
        uint32_t last = 0;
        while (hs.next_group_id()) {
            g.clear();
            uint32_t id = hs.next_group(g);
            if (id == last) return 4;                 // a group must hold all consecutive records of its id
            last = id;
            ++groups; hits += g.size();
            for (auto& h : g) sum = sum * 1000003ull + h.insert_id * 31ull + (uint64_t)h.h16.left * 7ull + h.h16.flags + (uint64_t)h.h16.right;
        }
        printf("%llu %llu %llu\n", (unsigned long long)groups, (unsigned long long)hits, (unsigned long long)sum);
        return 0;
    }
    if (mode == "fasta" && argc >= 3) {
        // hostio_check fasta <ref.fa>  -> one "<name> <length> <checksum>" line per contig
        RefTable rt;
        rt.load_fasta(argv[2]);
        for (size_t i = 0; i < rt.names.size(); ++i) {
            uint64_t sum = 0;
            for (char c : rt.seqs[i]) sum = sum * 131ull + (unsigned char)c;
            printf("%s %zu %llu\n", rt.names[i].c_str(), rt.seqs[i].size(), (unsigned long long)sum);
        }
        return 0;
    }
    if (mode == "merge" && argc >= 5) {
        // hostio_check merge <reads.fq> <mate_map|-> <seg1,seg2,...>: the ingest loop of segment_juncs without the device
        RefTable rt;
        thj_params p;
        thj_params_default(&p);
        std::vector<std::string> segs = split(argv[4], ',');
        const int nseg = (int)segs.size();
        std::vector<HitStream> st((size_t)nseg);
        for (int s2 = 0; s2 < nseg; ++s2) if (!st[(size_t)s2].open(segs[(size_t)s2], rt, p)) return 3;
        HitStream mate;
        bool have_mate = std::string(argv[3]) != "-" && mate.open(argv[3], rt, p);
        ReadStream reads;
        if (!reads.open(argv[2], "")) return 3;
        std::vector<std::vector<Hit>> grp((size_t)nseg);
        std::vector<Hit> mg;
        std::vector<thj_hit> hits, mate_hits; std::vector<uint32_t> seg_off(1, 0), mate_off(1, 0); std::string bases; std::vector<int64_t> read_off(1, 0);
        uint64_t nreads = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            uint32_t id = 0;
            for (int s2 = 0; s2 < nseg; ++s2) { uint32_t g = st[(size_t)s2].next_group_id(); if (g && (id == 0 || g < id)) id = g; }
            if (id == 0) break;
            int top = -1;
            for (int s2 = 0; s2 < nseg; ++s2) {
                grp[(size_t)s2].clear();
                if (st[(size_t)s2].next_group_id() == id) { st[(size_t)s2].next_group(grp[(size_t)s2]); top = s2; }
            }
            if (top <= 0) continue;
            Read rd;
            if (!reads.get(id, rd)) return 5;
            for (int s2 = 0; s2 < nseg; ++s2) { for (auto& h : grp[(size_t)s2]) hits.push_back(h.h16); seg_off.push_back((uint32_t)hits.size()); }
            if (have_mate) {
                mg.clear();
                while (mate.next_group_id() && mate.next_group_id() < id) mate.skip_group();
                if (mate.next_group_id() == id) mate.next_group(mg);
                for (auto& h : mg) mate_hits.push_back(h.h16);
                mate_off.push_back((uint32_t)mate_hits.size());
            }
            bases += rd.seq;
            read_off.push_back((int64_t)bases.size());
            ++nreads;
            if (nreads % (1 << 19) == 0) { hits.clear(); mate_hits.clear(); seg_off.assign(1, 0); mate_off.assign(1, 0); bases.clear(); read_off.assign(1, 0); }
        }
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%llu reads in %.3f s = %.3f us/read\n", (unsigned long long)nreads, dt, 1e6 * dt / (double)(nreads ? nreads : 1));
        return 0;
    }
    if (mode == "shardmerge" && argc >= 6) {
        // hostio_check shardmerge <n_shards> <reads> <mate_map|-> <seg1,seg2,...>: the ingest loop of segment_juncs over the
        // reference's shard plan (calculate_offsets over the .index files / probed text offsets), shard after shard.
        // Prints "<shards used> <reads> <hits> <mate hits> <checksum>" -- the last four must not depend on n_shards.
        RefTable rt;
        thj_params p;
        thj_params_default(&p);
        int want = atoi(argv[2]);
        const std::string reads_fn = argv[3], mate_fn = argv[4];
        std::vector<std::string> segs = split(argv[5], ',');
        const int nseg = (int)segs.size();
        for (auto& f : segs) register_targets(f, rt);
        if (mate_fn != "-") register_targets(mate_fn, rt);
        rt.freeze();
        struct Sh { uint64_t b = 0, e = ~0ull; int64_t ro = 0, po = 0; std::vector<int64_t> so; };
        std::vector<Sh> shards(1);
        shards[0].so.assign((size_t)nseg, 0);
        if (want > 1) {
            std::vector<IndexList> lists(1 + (size_t)nseg);
            load_index(reads_fn, want * 4, lists[0]);
            for (int k = 0; k < nseg; ++k) load_index(segs[(size_t)k], want * 4, lists[1 + (size_t)k]);
            size_t smallest = ~(size_t)0;
            for (auto& l : lists) smallest = std::min(smallest, l.size());
            if ((size_t)want > smallest) want = (int)smallest;
            std::vector<uint64_t> ids; std::vector<std::vector<int64_t>> offs;
            if (calculate_offsets(lists, want, ids, offs)) {
                std::vector<int64_t> po;
                if (mate_fn != "-") { IndexList l; load_index(mate_fn, want * 4, l); calculate_offsets_from_ids(l, ids, po); }
                shards.assign((size_t)want, Sh());
                for (int i = 0; i < want; ++i) {
                    Sh& sh = shards[(size_t)i];
                    sh.so.assign((size_t)nseg, 0);
                    if (i > 0) { sh.b = ids[(size_t)i - 1]; sh.ro = offs[(size_t)i - 1][0]; sh.so.assign(offs[(size_t)i - 1].begin() + 1, offs[(size_t)i - 1].end()); if (!po.empty()) sh.po = po[(size_t)i - 1]; }
                    sh.e = i + 1 < want ? ids[(size_t)i] : ~0ull;
                }
            }
        }
        uint64_t nreads = 0, nhits = 0, nmate = 0, sum = 0;
        for (auto& sh : shards) {
            std::vector<HitStream> st((size_t)nseg);
            for (int k = 0; k < nseg; ++k) if (!st[(size_t)k].open(segs[(size_t)k], rt, p, false, sh.so[(size_t)k], sh.b, sh.e)) return 3;
            HitStream mate;
            bool have_mate = mate_fn != "-" && mate.open(mate_fn, rt, p, false, sh.po, sh.b, sh.e);
            ReadStream reads;
            if (!reads.open(reads_fn, "", sh.ro)) return 3;
            std::vector<Hit> g, mg;
            for (;;) {
                uint32_t id = 0;
                for (int k = 0; k < nseg; ++k) { uint32_t x = st[(size_t)k].next_group_id(); if (x && (id == 0 || x < id)) id = x; }
                if (id == 0) break;
                Read rd;
                if (!reads.get(id, rd)) return 5;
                sum = sum * 1000003ull + id;
                for (char c : rd.seq) sum = sum * 131ull + (unsigned char)c;
                for (int k = 0; k < nseg; ++k) {
                    g.clear();
                    if (st[(size_t)k].next_group_id() == id) st[(size_t)k].next_group(g);
                    nhits += g.size();
                    for (auto& h : g) sum = sum * 1000003ull + (uint64_t)h.h16.left * 7ull + h.h16.flags + (uint64_t)h.h16.right + (uint64_t)k;
                }
                if (have_mate) {
                    mg.clear();
                    while (mate.next_group_id() && mate.next_group_id() < id) mate.skip_group();
                    if (mate.next_group_id() == id) mate.next_group(mg);
                    nmate += mg.size();
                    for (auto& h : mg) sum = sum * 1000003ull + (uint64_t)h.h16.left;
                }
                ++nreads;
            }
        }
        printf("%zu %llu %llu %llu %llu\n", shards.size(), (unsigned long long)nreads, (unsigned long long)nhits, (unsigned long long)nmate, (unsigned long long)sum);
        return 0;
    }
    if (mode == "reads" && argc >= 3) {
        // hostio_check reads <reads.fq> <id> [<id> ...]  -> one "<id> <seq> <qual>" line per request
        ReadStream rs;
        if (!rs.open(argv[2], "")) return 3;
        for (int i = 3; i < argc; ++i) {
            Read r;
            if (!rs.get((uint32_t)atoi(argv[i]), r)) { printf("%s MISSING\n", argv[i]); continue; }
            printf("%u %s %s\n", r.id, r.seq.c_str(), r.qual.c_str());
        }
        return 0;
    }
    return 2;
}
