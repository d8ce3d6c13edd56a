// thj_gen -- BENCH / TEST INFRASTRUCTURE: writes, at scale and fast, the files tophat.py hands to segment_juncs and
// long_spanning_reads for a synthetic paired-end run of the BASELINE config shapes: reference
// FASTA + SAM header, the reads as unaligned BAM (integer qnames, what prep_reads leaves), the whole-read maps and the
// per-segment maps as read-id-sorted BAM, every BAM with its `.index` side file (GBamWriter, common.h:562-606).  bowtie's
// part is emulated the way tophat_amd/synth.py:make_case does: a read or a segment maps where it was taken from when it lies
// inside one exon (or overhangs a junction by <= 3 bases) with <= 2 mismatches, and is absent from the map otherwise.
//
// Pair i is a pure function of (seed, i): `--pairs M` writes exactly the first M pairs of any larger run, so a small sample
// of a big case can be written again in text form (--text: FASTQ + SAM) for the CPU oracle.
//
// SURVEY 8(d)'s mix, with the rules of tophat_amd/synth.py:make_device_workload (what bench.py's resident-data line runs):
//   --multihit-frac F --max-copies C   the first 1/96 of contig 1 stands C times back to back (a repeat family: its copies lie further
//                                      apart than the longest intron); F of the pairs come from the genes inside the first copy and
//                                      both their reads report every hit at the first c copies, c = 2 for 85 % of them, 3..8 for 12 %,
//                                      9..40 for 2.7 %, 41 for 0.3 % (capped at C); the other pairs come from the genes behind the family
//   --indel-frac F                     F of the other pairs get a left read with 1..3 reference bases missing, one to three bases before
//                                      the end of a segment: that segment mapped ungapped with the mismatches its last bases then show
//                                      (absent with more than two), the following segments further on, no whole-read hit
//
//   thj_gen --out DIR --pairs N [--genome-len L | --contigs l1,l2,...] [--introns K] [--intron-max M] [--exon-len E]
//           [--read-len R] [--seed S] [--err 0.01] [--multihit-frac F --max-copies C] [--indel-frac F] [--text] [--threads T]
#include <cmath>

#include "../tophat_amd/csrc/host/thj_hostio.h"

using namespace thjh;

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull) { next(); next(); }
    uint64_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s * 0x2545F4914F6CDD1Dull; }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return next() % n; }
    double gauss() { double u = uni(), v = uni(); if (u < 1e-300) u = 1e-300; return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }
};

struct Gene { int contig; int64_t e1, d0, a1; };          // exon 1 = [e1, d0), intron = [d0, a1), exon 2 = [a1, a1 + exon_len)

static inline char comp(char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } }

static void md_nm(const char* ref, const char* rs, int n, int& nm, std::string& md) {
    md.clear(); nm = 0; int run = 0;
    for (int i = 0; i < n; ++i) {
        if (ref[i] == rs[i]) ++run;
        else { md += std::to_string(run); md.push_back(ref[i]); run = 0; ++nm; }
    }
    md += std::to_string(run);
}

int main(int argc, char** argv) {
    // the files stand for what tophat.py hands over: BAM written through samtools' bgzf.c, i.e. zlib at its default level -- not this
    // build's own fast DEFLATE, which the executables' OUTPUT uses (inflate kernels see different match statistics)
    setenv("THJ_BGZF_LEVEL", "-1", 0);
    std::string out; int64_t pairs = 100000, genome_len = 64444167; std::vector<int64_t> contigs;
    int introns = 20000, intron_max = 200000, exon_len = 300, read_len = 100, seg_len = 25, threads = host_threads();
    uint64_t seed = 1; double err = 0.01, drop = 0.03, multihit_frac = 0.0, indel_frac = 0.0; bool text = false; int max_copies = 41;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) die("thj_gen: %s needs a value\n", a.c_str()); return argv[++i]; };
        if (a == "--out") out = val();
        else if (a == "--pairs") pairs = atoll(val());
        else if (a == "--genome-len") genome_len = atoll(val());
        else if (a == "--contigs") { for (auto& t : split(val(), ',')) contigs.push_back(atoll(t.c_str())); }
        else if (a == "--introns") introns = atoi(val());
        else if (a == "--intron-max") intron_max = atoi(val());
        else if (a == "--exon-len") exon_len = atoi(val());
        else if (a == "--read-len") read_len = atoi(val());
        else if (a == "--seed") seed = (uint64_t)atoll(val());
        else if (a == "--err") err = atof(val());
        else if (a == "--drop") drop = atof(val());
        else if (a == "--multihit-frac") multihit_frac = atof(val());
        else if (a == "--max-copies") max_copies = atoi(val());
        else if (a == "--indel-frac") indel_frac = atof(val());
        else if (a == "--threads") threads = atoi(val());
        else if (a == "--text") text = true;
        else die("thj_gen: unknown option %s\n", a.c_str());
    }
    if (out.empty()) die("thj_gen: --out DIR is required\n");
    if (contigs.empty()) contigs.push_back(genome_len);
    const int nseg = std::max(1, read_len / seg_len);              // the last segment takes the remainder (tophat.py:2948)
    if (2 * exon_len < 2 * read_len + 150) die("thj_gen: exons too short for the fragments\n");

    // ---- genome with planted two-exon genes (as tophat_amd/synth.py:make_scale_genome: introns log-uniform, GT-AG 90 % /
    // GC-AG 7 % / AT-AC 3 %, both strands)
    std::vector<std::string> seqs(contigs.size()), names(contigs.size());
    std::vector<Gene> genes;
    double total = 0; for (auto l : contigs) total += (double)l;
    for (size_t ci = 0; ci < contigs.size(); ++ci) {
        names[ci] = contigs.size() == 1 ? "chr20" : "chr" + std::to_string(ci + 1);
        std::string& s = seqs[ci];
        s.resize((size_t)contigs[ci]);
        {
            const int T = std::max(1, threads);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t]() {
                Rng r(seed * 1000003 + ci * 131 + (uint64_t)t);
                const size_t a = s.size() * (size_t)t / (size_t)T, b = s.size() * (size_t)(t + 1) / (size_t)T;
                for (size_t i = a; i < b;) { uint64_t x = r.next(); for (int k = 0; k < 32 && i < b; ++k, ++i, x >>= 2) s[i] = "ACGT"[x & 3]; }
            });
            for (auto& x : th) x.join();
        }
        Rng r(seed * 7919 + ci);
        const int want = std::max(1, (int)llround(introns * (double)contigs[ci] / total));
        int64_t p = 1000;
        for (int k = 0; k < want; ++k) {
            const int64_t il = (int64_t)exp(log(70.0) + r.uni() * (log((double)intron_max) - log(70.0)));
            if (p + 2 * exon_len + il + 1000 > contigs[ci]) break;
            const int64_t d0 = p + exon_len, a1 = d0 + il;
            const double u = r.uni();
            const char* don = u < 0.90 ? "GT" : (u < 0.97 ? "GC" : "AT");
            const char* acc = u < 0.97 ? "AG" : "AC";
            if (r.uni() < 0.5) { s[(size_t)d0] = don[0]; s[(size_t)d0 + 1] = don[1]; s[(size_t)a1 - 2] = acc[0]; s[(size_t)a1 - 1] = acc[1]; }
            else { s[(size_t)d0] = comp(acc[1]); s[(size_t)d0 + 1] = comp(acc[0]); s[(size_t)a1 - 2] = comp(don[1]); s[(size_t)a1 - 1] = comp(don[0]); }
            genes.push_back({(int)ci, p, d0, a1});
            p = a1 + exon_len + 200 + (int64_t)r.below(800);
        }
    }
    if (genes.empty()) die("thj_gen: no gene fits the genome\n");
    // the repeat family: copies of the first 1/96 of contig 1, its genes; the genes behind the last copy stay unique, those in between go
    int64_t dup_shift = 0;
    std::vector<Gene> fam, uniq;
    if (multihit_frac > 0) {
        if (max_copies < 2) die("thj_gen: --max-copies must be at least 2\n");
        dup_shift = (int64_t)seqs[0].size() / 96;
        if (dup_shift <= std::max<int64_t>(500000, intron_max + 1) + 2 * exon_len) die("thj_gen: the genome is too small for a repeat family whose copies lie further apart than the longest intron\n");
        for (int k = 1; k < max_copies; ++k) memcpy(&seqs[0][(size_t)(k * dup_shift)], &seqs[0][0], (size_t)dup_shift);
        for (const Gene& g : genes) {
            if (g.contig == 0 && g.a1 + exon_len + 1000 < dup_shift) fam.push_back(g);
            else if (g.contig != 0 || g.e1 >= (int64_t)max_copies * dup_shift + 1000) uniq.push_back(g);
        }
        if (fam.empty() || uniq.empty()) die("thj_gen: the repeat family needs genes inside the first copy and behind the last one\n");
    }
    std::string cmd = "mkdir -p '" + out + "'";
    if (system(cmd.c_str())) die("thj_gen: cannot create %s\n", out.c_str());
    // FASTA + header
    RefTable rt;
    rt.header_text = "@HD\tVN:1.0\tSO:unsorted\n";
    for (size_t ci = 0; ci < contigs.size(); ++ci) {
        rt.header_text += "@SQ\tSN:" + names[ci] + "\tLN:" + std::to_string(contigs[ci]) + "\n";
        rt.sq.emplace_back(names[ci], (uint32_t)contigs[ci]);
        rt.get_id(names[ci]);
    }
    { FILE* f = fopen((out + "/hdr.sam").c_str(), "w"); fputs(rt.header_text.c_str(), f); fclose(f); }
    {
        FILE* f = fopen((out + "/ref.fa").c_str(), "w");
        std::vector<char> buf((size_t)1 << 22);
        setvbuf(f, buf.data(), _IOFBF, buf.size());
        for (size_t ci = 0; ci < contigs.size(); ++ci) {
            fprintf(f, ">%s\n", names[ci].c_str());
            const std::string& s = seqs[ci];
            for (size_t i = 0; i < s.size(); i += 60) { fwrite(s.data() + i, 1, std::min<size_t>(60, s.size() - i), f); fputc('\n', f); }
        }
        fclose(f);
    }

    // ---- outputs: per side {reads, map, seg1..segN}
    const char* SIDES[2] = {"left", "right"};
    struct Sink { BamWriter bw; FILE* txt = nullptr; };
    std::vector<std::unique_ptr<Sink>> sinks;                    // index: side * (2 + nseg) + {0 reads, 1 map, 2 + k segment k}
    RefTable none;                                               // the reads BAM has no targets
    none.header_text = "@HD\tVN:1.0\tSO:unsorted\n";
    for (int sd = 0; sd < 2; ++sd)
        for (int f = 0; f < 2 + nseg; ++f) {
            sinks.emplace_back(new Sink());
            std::string base = out + "/" + SIDES[sd] + (f == 0 ? "_reads" : f == 1 ? "_map" : "_seg" + std::to_string(f - 1));
            if (!sinks.back()->bw.open(base + ".bam", f == 0 ? none : rt, base + ".bam.index")) die("thj_gen: cannot create %s.bam\n", base.c_str());
            if (text) {
                sinks.back()->txt = fopen((f == 0 ? out + "/" + SIDES[sd] + ".fq" : base + ".sam").c_str(), "w");
                if (f != 0) fputs(rt.header_text.c_str(), sinks.back()->txt);
            }
        }
    const int NF = 2 * (2 + nseg);
    struct Block { std::vector<BamWriter::Encoded> enc; std::vector<std::string> txt; };
    const int64_t BLK = 32768;
    const int64_t n_blocks = (pairs + BLK - 1) / BLK;
    const std::string qual_read((size_t)read_len, 'I');
    const int tx_len = 2 * exon_len;

    auto gen_block = [&](int64_t b, Block& blk) {
        blk.enc.assign((size_t)NF, BamWriter::Encoded());
        blk.txt.assign((size_t)NF, std::string());
        std::string tx((size_t)tx_len, 'N'), F, seq, piece, md, qn;
        std::vector<std::string> aux(2);
        for (int64_t i = b * BLK; i < std::min(pairs, (b + 1) * BLK); ++i) {
            const long rid = (long)i + 1;
            Rng r(seed * 0x100000001B3ull + (uint64_t)i);
            int ncopy = 1;
            const Gene* gp;
            if (dup_shift) {
                const bool is_multi = r.uni() < multihit_frac;
                gp = is_multi ? &fam[(size_t)r.below(fam.size())] : &uniq[(size_t)r.below(uniq.size())];
                const double u = r.uni();
                const int c = u < 0.85 ? 2 : u < 0.97 ? 3 + (int)r.below(6) : u < 0.997 ? 9 + (int)r.below(32) : 41;
                if (is_multi) ncopy = std::min(c, max_copies);
            } else gp = &genes[(size_t)r.below(genes.size())];
            const Gene& g = *gp;
            const bool del_pair = indel_frac > 0 && nseg >= 3 && ncopy == 1 && r.uni() < indel_frac;
            const std::string& gs = seqs[(size_t)g.contig];
            memcpy(&tx[0], gs.data() + g.e1, (size_t)exon_len);
            memcpy(&tx[(size_t)exon_len], gs.data() + g.a1, (size_t)exon_len);
            int inner = (int)llround(50.0 + 20.0 * r.gauss()); if (inner < 0) inner = 0;
            int frag = 2 * read_len + inner; if (frag > tx_len) frag = tx_len;
            const int t0 = (int)r.below((uint64_t)(tx_len - frag + 1));
            const bool flip = r.uni() < 0.5;
            for (int sd = 0; sd < 2; ++sd) {
                // FR library: one read sense at the fragment start, its mate antisense at the fragment end
                const bool anti = (sd == 0) == flip;
                const int rt0 = anti ? t0 + frag - read_len : t0;
                F.assign(tx, (size_t)rt0, (size_t)read_len);
                for (int k = 0; k < read_len; ++k) if (r.uni() < err) { char c; do c = "ACGT"[r.below(4)]; while (c == F[(size_t)k]); F[(size_t)k] = c; }
                seq = F;
                if (anti) reverse_complement(seq);
                const int base_f = sd * (2 + nseg);
                auto emit1 = [&](int f, const std::string& qname, uint32_t flag, int contig, int64_t pos0, int len, const std::string& s, bool mapped, const std::string& mdv, int nm) {
                    BamWriter::Encoded& e = blk.enc[(size_t)(base_f + f)];
                    const size_t before = e.bytes.size();
                    uint32_t cig = (1u << 28) | (uint32_t)len;
                    std::vector<std::string> ax;
                    if (mapped) { ax.push_back("NM:i:" + std::to_string(nm)); ax.push_back("MD:Z:" + mdv); }
                    sinks[(size_t)(base_f + f)]->bw.encode(e.bytes, qname, flag, mapped ? names[(size_t)contig] : std::string("*"), mapped ? (int)pos0 + 1 : 0,
                                                           &cig, mapped ? 1 : 0, s, qual_read.substr(0, (size_t)len), ax);
                    e.size.push_back((uint32_t)(e.bytes.size() - before));
                    e.rid.push_back(rid);
                    if (text) {
                        std::string& t = blk.txt[(size_t)(base_f + f)];
                        if (f == 0) t += "@" + qname + "\n" + s + "\n+\n" + qual_read + "\n";
                        else t += qname + "\t" + std::to_string(flag) + "\t" + names[(size_t)contig] + "\t" + std::to_string(pos0 + 1) + "\t255\t" + std::to_string(len) +
                                  "M\t*\t0\t0\t" + s + "\t" + qual_read.substr(0, (size_t)len) + "\tNM:i:" + std::to_string(nm) + "\tMD:Z:" + mdv + "\n";
                    }
                };
                // a read of the family reports every hit at its first ncopy copies (bowtie -k: one record each, the read's records together)
                auto emit = [&](int f, const std::string& qname, uint32_t flag, int contig, int64_t pos0, int len, const std::string& s, bool mapped, const std::string& mdv, int nm) {
                    for (int c = 0; c < (mapped ? ncopy : 1); ++c) emit1(f, qname, flag, contig, pos0 + (int64_t)c * dup_shift, len, s, mapped, mdv, nm);
                };
                if (del_pair && sd == 0) {
                    // the deletion read: forward, from the gene's first exon, dl reference bases missing at read offset x
                    const int64_t pa = g.e1 + (int64_t)r.below((uint64_t)std::max(1, exon_len - read_len - 4));
                    const int dl = 1 + (int)r.below(3), kb = 1 + (int)r.below((uint64_t)std::max(1, nseg - 2)), m = 1 + (int)r.below(3);
                    const int x = kb * seg_len - m;
                    F.resize((size_t)read_len);
                    for (int k = 0; k < read_len; ++k) F[(size_t)k] = gs[(size_t)(pa + k + (k >= x ? dl : 0))];
                    emit(0, std::to_string(rid), 4, 0, 0, read_len, F, false, "", 0);
                    for (int k = 0; k < nseg; ++k) {
                        const int s0 = k * seg_len, s1 = k == nseg - 1 ? read_len : (k + 1) * seg_len;
                        const int64_t pos = pa + s0 + (k >= kb ? dl : 0);       // the segment that holds the deletion: in the frame before it
                        int nm;
                        md_nm(gs.data() + pos, F.data() + s0, s1 - s0, nm, md);
                        if (nm > 2) continue;
                        piece.assign(F, (size_t)s0, (size_t)(s1 - s0));
                        qn = std::to_string(rid) + "|" + std::to_string(s0) + ":" + std::to_string(k) + ":" + std::to_string(nseg);
                        emit(2 + k, qn, 0u, g.contig, pos, s1 - s0, piece, true, md, nm);
                    }
                    continue;
                }
                emit(0, std::to_string(rid), 4, 0, 0, read_len, seq, false, "", 0);
                // place a transcript interval [a, a + len) contiguously: inside one exon, or overhanging the junction by <= `oh`
                auto place = [&](int a, int len, const char* bases, int oh, int64_t& pos, int& nm) -> bool {
                    if (a + len <= exon_len) pos = g.e1 + a;                               // inside exon 1
                    else if (a >= exon_len) pos = g.a1 + (a - exon_len);                   // inside exon 2
                    else if (a + len - exon_len <= oh) pos = g.e1 + a;                     // a few bases hang over into the intron
                    else if (exon_len - a <= oh) pos = g.a1 - (exon_len - a);
                    else return false;
                    if (pos < 0 || pos + len > (int64_t)gs.size()) return false;
                    md_nm(gs.data() + pos, bases, len, nm, md);
                    return nm <= 2;
                };
                for (int k = 0; k < nseg; ++k) {
                    const int s0 = k * seg_len, s1 = k == nseg - 1 ? read_len : (k + 1) * seg_len;
                    const bool dropped = r.uni() < drop;
                    if (dropped) continue;
                    const int f0 = anti ? read_len - s1 : s0, f1 = anti ? read_len - s0 : s1;
                    int64_t pos; int nm;
                    if (!place(rt0 + f0, f1 - f0, F.data() + f0, 3, pos, nm)) continue;
                    piece.assign(F, (size_t)f0, (size_t)(f1 - f0));
                    qn = std::to_string(rid) + "|" + std::to_string(s0) + ":" + std::to_string(k) + ":" + std::to_string(nseg);
                    emit(2 + k, qn, anti ? 16u : 0u, g.contig, pos, f1 - f0, piece, true, md, nm);
                }
                int64_t pos; int nm;
                if (place(rt0, read_len, F.data(), 0, pos, nm)) emit(1, std::to_string(rid), anti ? 16u : 0u, g.contig, pos, read_len, F, true, md, nm);
            }
        }
    };

    // blocks are generated by `threads` workers, written in order
    std::vector<std::unique_ptr<Block>> ready((size_t)n_blocks);
    std::mutex mu; std::condition_variable cv;
    std::atomic<int64_t> next{0};
    int64_t written = 0;
    std::vector<std::thread> th;
    for (int t = 0; t < std::max(1, threads); ++t) th.emplace_back([&]() {
        for (;;) {
            const int64_t b = next.fetch_add(1);
            if (b >= n_blocks) return;
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return b < written + 3 * (int64_t)std::max(1, threads); }); }
            std::unique_ptr<Block> blk(new Block());
            gen_block(b, *blk);
            { std::lock_guard<std::mutex> lk(mu); ready[(size_t)b] = std::move(blk); }
            cv.notify_all();
        }
    });
    for (int64_t b = 0; b < n_blocks; ++b) {
        std::unique_ptr<Block> blk;
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return ready[(size_t)b] != nullptr; }); blk = std::move(ready[(size_t)b]); }
        for (int f = 0; f < NF; ++f) {
            sinks[(size_t)f]->bw.write_encoded(blk->enc[(size_t)f]);
            if (text) fputs(blk->txt[(size_t)f].c_str(), sinks[(size_t)f]->txt);
        }
        { std::lock_guard<std::mutex> lk(mu); written = b + 1; }
        cv.notify_all();
    }
    for (auto& t : th) t.join();
    for (auto& s : sinks) { s->bw.close(); if (s->txt) fclose(s->txt); }
    printf("{\"pairs\": %lld, \"genes\": %zu, \"nseg\": %d, \"read_len\": %d, \"contigs\": %zu}\n", (long long)pairs, genes.size(), nseg, read_len, contigs.size());
    return 0;
}
