// thj_junctions -- the junction consensus of tophat_reports as a program: reads the alignments
// tophat_reports would report from BAM files (the spanning BAMs of long_spanning_reads, whole-read maps, an accepted_hits
// file ...), reduces their REF_SKIPs to the JunctionSet on the device (thj_juncbed_*, include/thj.h) and prints junctions.bed
// exactly as print_junctions does (junctions.cpp:100-120, :330-350).
//
//   thj_junctions [--min-anchor N] [--sam-header hdr.sam] <ref.fa> <junctions.bed> <in1.bam[,in2.bam,...]>
//
// Not tophat_reports: the choice among a read's alignments (read_best_alignments, pair grading, realign_reads) is not made
// here -- every record of the inputs counts as reported.  Records without a REF_SKIP cannot touch the result and are skipped
// while reading.
#include "thj_hostio.h"

using namespace thjh;

static void usage() { fprintf(stderr, "Usage:   thj_junctions [--min-anchor N] [--sam-header hdr.sam] <ref.fa> <junctions.bed> <alignments1.bam[,alignments2.bam,...]>\n"); }

static int real_main(int argc, char** argv) {
    Opts o;
    int rc = parse_options(argc, argv, o, usage);
    if (rc) return rc;
    std::vector<std::string> pos;
    for (int i = optind; i < argc; ++i) pos.push_back(argv[i]);
    if (pos.size() < 3) { usage(); return 1; }
    std::future<thj_ctx*> fut = std::async(std::launch::async, []() {
        thj_ctx* c = nullptr;
        if (thj_ctx_create(getenv("THJ_DEVICE") ? atoi(getenv("THJ_DEVICE")) : 0, nullptr, &c)) die("Error: %s\n", thj_last_error());
        return c;
    });
    RefTable rt;
    rt.load_sam_header(o.sam_header);
    rt.load_fasta(pos[0]);
    std::vector<std::string> inputs = split(pos[2], ',');
    for (auto& f : inputs) register_targets(f, rt);
    rt.freeze();
    // ---- spliced records -> thj_aln (only the fields the reduce reads).  BGZF members inflate independently, so every input is cut
    // into runs of members, one per host thread; a run must end on a record boundary (true of every BAM written through
    // bgzf_flush_try: samtools 0.1.18's writer, this build's) -- if one does not, that input is read again by the sequential reader.
    std::vector<std::vector<thj_aln>> recs(inputs.size());
    // One record -> thj_aln, or nothing.  A fusion alignment comes as two records that both carry the whole alignment in an XF:Z tag
    // ("1|2 <contig1>-<contig2> <pos> <cigar with an F op> <bases> <qualities>", print_bamhit, bwt_map.cpp:2047-2083): the first one is
    // rebuilt from the tag the way BAMHitFactory::get_hit_from_buf does (bwt_map.cpp:1208-1318 -- lower-case ops for pieces that run down
    // the genome, F = position on the second contig + 1, its direction FF / FR / RF / RR from the ops around it), the second is dropped.
    const int max_report_intron = o.p.max_report_intron;
    auto parse_record = [&rt, max_report_intron](const uint8_t* d, int32_t bs, const std::vector<uint32_t>& tid2ref, std::vector<thj_aln>& out) {
        static const uint32_t OPS[9] = {THJ_CIG_MATCH, THJ_CIG_INS, THJ_CIG_DEL, THJ_CIG_REF_SKIP, THJ_CIG_SOFT_CLIP, 14u, 15u, THJ_CIG_MATCH, THJ_CIG_MATCH};
        int32_t tid, p0; uint32_t bin_mq_nl, flag_nc; int32_t l_seq;
        memcpy(&tid, d, 4); memcpy(&p0, d + 4, 4); memcpy(&bin_mq_nl, d + 8, 4); memcpy(&flag_nc, d + 12, 4); memcpy(&l_seq, d + 16, 4);
        if (!bam_record_shape_ok(d, bs)) die("Error: malformed BAM record (its header does not fit its %d bytes)\n", (int)bs);
        const uint32_t l_rn = bin_mq_nl & 0xFF, n_cig = flag_nc & 0xFFFF;
        if (tid < 0 || ((flag_nc >> 16) & 4)) return;
        thj_aln a; memset(&a, 0, sizeof a);
        bool spliced = false;
        size_t pp = 32 + l_rn;
        for (uint32_t i = 0; i < n_cig; ++i) { uint32_t c; memcpy(&c, d + pp, 4); pp += 4; const uint32_t op = (c & 0xF) < 9 ? OPS[c & 0xF] : 15u; if (op == THJ_CIG_REF_SKIP) spliced = true; if (i < 16) a.cigar[i] = (op << 28) | (c >> 4); }
        pp += (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
        char xs = 0; const char* xf = nullptr;
        while (pp + 3 <= (size_t)bs) {                        // XS:A, XF:Z
            const char t0 = (char)d[pp], t1 = (char)d[pp + 1], ty = (char)d[pp + 2];
            pp += 3;
            if (bam_aux_fixed_size(ty) > (size_t)bs - pp) die("Error: malformed BAM record (a tag runs past its end)\n");
            switch (ty) {
            case 'A': if (t0 == 'X' && t1 == 'S') xs = (char)d[pp]; pp += 1; break;
            case 'c': case 'C': pp += 1; break;
            case 's': case 'S': pp += 2; break;
            case 'i': case 'I': case 'f': pp += 4; break;
            case 'd': pp += 8; break;
            case 'Z': case 'H': { const size_t at = pp; while (pp < (size_t)bs && d[pp]) ++pp; if (pp < (size_t)bs && ty == 'Z' && t0 == 'X' && t1 == 'F') xf = (const char*)d + at; ++pp; break; }
            case 'B': { char st = (char)d[pp]; int32_t cnt; memcpy(&cnt, d + pp + 1, 4); if (cnt < 0 || (int64_t)cnt > (int64_t)bs) die("Error: malformed BAM record (an array tag runs past its end)\n");
                        pp += 5 + (size_t)cnt * ((st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4); break; }
            default: pp = (size_t)bs; break;
            }
        }
        a.flags = (uint8_t)(xs == '-' ? THJ_HIT_ANTISENSE_SPLICE : 0);
        if (xf) {
            if (xf[0] == '2') return;                          // "ignore the second part of a fusion alignment" (:1214-1216)
            std::vector<std::string> f = split(xf, ' ');
            if (f.size() < 4) return;
            std::vector<std::string> cs = split(f[1], '-');
            // fewer than two names in the contig field: the reference keeps the record's own target as the first contig and an empty name as
            // the second (bwt_map.cpp:1219-1225: text_name / text_name2 stay as they were) and goes on; so does this
            const uint32_t r1 = cs.size() >= 2 ? rt.get_id(cs[0]) : ((size_t)tid < tid2ref.size() ? tid2ref[(size_t)tid] : 0), r2 = cs.size() >= 2 ? rt.get_id(cs[1]) : rt.get_id(std::string());
            if (!r1 || !r2) return;
            int n = 0; bool spl = false;
            uint32_t op[16];
            for (const char* q = f[3].c_str(); *q;) {
                char* t; const long len0 = strtol(q, &t, 10); long len = len0;
                if (len <= 0) return;
                uint32_t code;
                switch (*t) {
                case 'M': code = 1; break; case 'm': code = 2; break; case 'I': code = 3; break; case 'i': code = 4; break;
                case 'D': code = 5; break; case 'd': code = 6; break;
                case 'N': case 'n': if (len > max_report_intron) return; code = *t == 'N' ? 11 : 12; spl = true; break;
                case 'F': code = 7; len = len - 1; break;
                case 'S': code = 13; break; case 'H': code = 14; break; case 'P': code = 15; break;
                default: return;
                }
                q = t + 1;
                if (n >= 15) die("Error: a fusion alignment of more than 15 CIGAR operations (XF:Z:%s)\n", xf);
                op[n++] = code << 28 | ((uint32_t)len & 0x0FFFFFFFu);
                if (n >= 3 && (op[n - 2] >> 28) == 7) {         // the direction of the fusion from the pieces around it (:1283-1301)
                    auto up = [](uint32_t c) { c >>= 28; return c == 1 || c == 5 || c == 3 || c == 11; };
                    const bool i1 = up(op[n - 3]), i2 = up(op[n - 1]);
                    const uint32_t dir = (i1 && !i2) ? 8u : (!i1 && i2) ? 9u : (!i1 && !i2) ? 10u : 7u;
                    op[n - 2] = dir << 28 | (op[n - 2] & 0x0FFFFFFFu);
                }
            }
            if (!spl) return;
            memset(a.cigar, 0, sizeof a.cigar);
            for (int i = 0; i < n; ++i) a.cigar[i] = op[i];
            a.cigar[15] = r2;
            a.ref_id = r1; a.left = atoi(f[2].c_str()) - 1; a.n_cigar = (uint8_t)n;
            out.push_back(a);
            return;
        }
        if (!spliced || n_cig < 3) return;
        // a spliced alignment the consensus cannot hold must not vanish from the support counts silently
        if (n_cig > 16) die("Error: a spliced alignment of %u CIGAR operations (at most 16 are supported)\n", n_cig);
        a.ref_id = (size_t)tid < tid2ref.size() ? tid2ref[(size_t)tid] : 0;
        if (!a.ref_id) return;
        a.left = p0; a.n_cigar = (uint8_t)n_cig;
        out.push_back(a);
    };
    auto read_parallel = [&](const std::string& fn, std::vector<thj_aln>& out) -> bool {
        BamFile bf;
        if (getenv("THJ_SEQUENTIAL_READ") || !bf.open(fn, rt)) return false;
        std::vector<size_t> moff;
        for (size_t off = (size_t)(bf.first_rec_voff >> 16); off < bf.size;) {
            const uint32_t bs = BamFile::member_size(bf.data + off, bf.size - off);
            if (!bs || off + bs > bf.size) return false;
            moff.push_back(off); off += bs;
        }
        const size_t nm = moff.size();
        if (!nm) return true;
        const size_t T = std::min<size_t>((size_t)std::max(1, host_threads()), nm);
        std::vector<std::vector<thj_aln>> part(T);
        std::vector<char> bad(T, 0);
        std::vector<std::thread> th;
        for (size_t t = 0; t < T; ++t) th.emplace_back([&, t]() {
            std::vector<uint8_t> buf; size_t have = 0;
            z_stream zs; memset(&zs, 0, sizeof zs);
            if (inflateInit2(&zs, -15) != Z_OK) { bad[t] = 1; return; }
            for (size_t m = nm * t / T; m < nm * (t + 1) / T && !bad[t]; ++m) {
                const uint8_t* d = bf.data + moff[m];
                const uint32_t bs = BamFile::member_size(d, bf.size - moff[m]), xlen = d[10] | (d[11] << 8);
                uint32_t isz; memcpy(&isz, d + bs - 4, 4);
                if (buf.size() < have + isz) buf.resize(have + isz + 65536);
                inflateReset(&zs);
                zs.next_in = const_cast<uint8_t*>(d + 12 + xlen); zs.avail_in = bs - 12 - xlen - 8;
                zs.next_out = buf.data() + have; zs.avail_out = isz;
                if (inflate(&zs, Z_FINISH) != Z_STREAM_END && isz) { bad[t] = 1; break; }
                size_t end = have + isz;
                size_t pos = m == 0 ? (size_t)(bf.first_rec_voff & 0xFFFF) : 0;      // the first member may still hold the end of the header
                if (m == 0 && pos > end) { bad[t] = 1; break; }
                while (pos + 4 <= end) {
                    int32_t rs; memcpy(&rs, buf.data() + pos, 4);
                    if (rs < 32) { bad[t] = 1; break; }
                    if (pos + 4 + (size_t)rs > end) break;
                    parse_record(buf.data() + pos + 4, rs, bf.tid2ref, part[t]);
                    pos += 4 + (size_t)rs;
                }
                have = end - pos;
                if (have) memmove(buf.data(), buf.data() + pos, have);
            }
            inflateEnd(&zs);
            if (have) bad[t] = 1;                                  // a record straddles the end of this run
        });
        for (auto& x : th) x.join();
        for (char b : bad) if (b) return false;
        size_t tot = 0;
        for (auto& v : part) tot += v.size();
        out.reserve(tot);
        for (auto& v : part) out.insert(out.end(), v.begin(), v.end());
        return true;
    };
    std::vector<std::thread> th;
    for (size_t k = 0; k < inputs.size(); ++k) th.emplace_back([&, k]() {
        if (read_parallel(inputs[k], recs[k])) return;
        recs[k].clear();
        AlnReader rd;
        if (!rd.open(inputs[k])) die("Error: cannot open %s\n", inputs[k].c_str());
        if (!rd.is_bam()) die("Error: %s: BAM input expected\n", inputs[k].c_str());
        std::vector<uint32_t> tid2ref;
        for (auto& t : rd.targets()) tid2ref.push_back(rt.get_id(t));
        int32_t bs = 0;
        while (const uint8_t* d = rd.next_raw(bs)) parse_record(d, bs, tid2ref, recs[k]);
    });
    for (auto& t : th) t.join();
    thj_ctx* ctx = fut.get();
    rt.upload(ctx);
    int64_t cap = 0;
    for (int attempt = 0;; ++attempt) {
        if (cap && thj_juncbed_configure(ctx, cap)) die("Error: %s\n", thj_last_error());
        if (thj_juncbed_reset_async(ctx)) die("Error: %s\n", thj_last_error());
        for (auto& v : recs) if (thj_juncbed_add_records(ctx, v.data(), (int64_t)v.size(), 0)) die("Error: %s\n", thj_last_error());
        int64_t n = 0;
        rc = thj_juncbed_finish(ctx, o.p.min_anchor_len, &n);
        if (rc == THJ_EOVERFLOW && attempt < 6) { cap = cap ? cap * 4 : (int64_t)1 << 22; continue; }     // more distinct junctions than the table holds
        if (rc) die("Error: %s\n", thj_last_error());
        std::vector<thj_juncstat> js((size_t)n + 1);
        if (thj_juncbed_download(ctx, js.data())) die("Error: %s\n", thj_last_error());
        FILE* f = fopen(pos[1].c_str(), "w");
        if (!f) die("Error: cannot open %s for writing\n", pos[1].c_str());
        fprintf(f, "track name=junctions description=\"TopHat junctions\"\n");
        for (int64_t i = 0; i < n; ++i) {
            const thj_juncstat& j = js[(size_t)i];
            const int start = (int)j.left + 1 - (int)j.left_extent, end = (int)j.right + (int)j.right_extent;
            fprintf(f, "%s\t%d\t%d\tJUNC%08d\t%d\t%c\t%d\t%d\t255,0,0\t2\t%d,%d\t0,%d\n", rt.names[j.ref_id - 1].c_str(), start, end, (int)(i + 1), (int)j.support,
                    j.antisense ? '-' : '+', start, end, (int)j.left_extent, (int)j.right_extent, (int)j.right - start);
        }
        close_output(f, "junctions.bed");
        break;
    }
    finish_outputs_complete(0);
}

int main(int argc, char** argv) { return run_with_handoff(argc, argv, real_main); }
