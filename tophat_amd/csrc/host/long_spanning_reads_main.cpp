// long_spanning_reads -- MI355X-native drop-in for TopHat's long_spanning_reads (same argv + files;
// tophat.py:3160-3191, parsed like long_spanning_reads.cpp:3151-3329).  Host C++ over include/thj.h.
// Contig segment maps (with any CIGAR, incl. N/I/D) and junction-db ("spliced") segment maps are supported, and so is
// --fusion-search: the .fusions list goes to the device, fused segment hits come from the fusion contigs of the junction
// database (BAM maps), fusion alignments leave as two records with XF:Z (print_bamhit / extract_partial_hits).
//
// The reads are cut into contiguous read-id shards with the reference's planner (calculate_offsets over the inputs' .index
// files, utils.cpp:22-127; long_spanning_reads.cpp:2983-3064).  Host workers ingest shards in parallel -- shard k's
// batches run on GPU k mod n -- and encode their records; with one output file a writer appends the shards' records in
// shard order (= read order), so the BAM stream and its .index do not depend on the number of shards; with -p N the N
// parts are the reference's N ranges, one file each (<base>{0..N-1}.bam).
#include <deque>

#include "thj_hostio.h"
#include <sys/stat.h>

using namespace thjh;

static std::atomic<long long> g_host_ingest_shards{0};      // shards the device-side ingest declined (the host readers took them)
static void print_usage() {
    fprintf(stderr, "Usage:   long_spanning_reads <reference.fasta> <reads.fq> <possible_juncs1,...,possible_juncsN> "
                    "<possible_insertions1,...,possible_insertionsN> <possible_deletions1,...,possible_deletionsN> "
                    "<possible_fusions1,...,possible_fusionsN> <out.bam> <seg1.bwtout,...,segN.bwtout> [spliced_seg1.bwtout,...,spliced_segN.bwtout]\n");
}

static PhaseTimer g_timer;
static WorkClock g_work;
// THJ_TRACE=1: one line per shard event on stderr (seconds since start): where a shard's time goes, for tools/lsr_trace.py
static const bool g_trace = getenv("THJ_TRACE") != nullptr;
static const long long g_trace_t0 = WorkClock::now();
static void trace(size_t shard, const char* what) { if (g_trace) fprintf(stderr, "[trace] %zu %s %.4f\n", shard, what, (double)(WorkClock::now() - g_trace_t0) * 1e-9); }

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
    int64_t read_off = 0, read_end = -1;            // read_end: where the next shard starts in the reads file (-1: the end)
    std::vector<int64_t> seg_off, spliced_off;
    std::vector<int64_t> seg_end;                   // where the next shard starts in the segment maps (-1: the end of the file)
};

// long_spanning_reads.cpp:2983-2991, :3051-3064: index files in the order {reads, spliced maps last..first, contig maps
// last..first} -- the boundaries come from the FIRST segment's contig map, the stream the worker iterates over
static std::vector<Shard> plan(const std::string& reads, const std::vector<std::string>& segs, const std::vector<std::string>& spliced, int want) {
    std::vector<Shard> out(1);
    out[0].seg_off.assign(segs.size(), 0); out[0].spliced_off.assign(spliced.size(), 0); out[0].seg_end.assign(segs.size(), -1);
    if (want < 2) return out;
    std::vector<std::string> fnames;
    fnames.push_back(reads);
    fnames.insert(fnames.end(), spliced.rbegin(), spliced.rend());
    fnames.insert(fnames.end(), segs.rbegin(), segs.rend());
    std::vector<IndexList> lists(fnames.size());
    size_t smallest = ~(size_t)0;
    for (size_t i = 0; i < fnames.size(); ++i) { load_index(fnames[i], want * 4, lists[i]); smallest = std::min(smallest, lists[i].size()); }
    if ((size_t)want > smallest) want = (int)smallest;
    std::vector<uint64_t> ids; std::vector<std::vector<int64_t>> offs;
    if (!calculate_offsets(lists, want, ids, offs)) return out;
    out.assign((size_t)want, Shard());
    for (int i = 0; i < want; ++i) {
        Shard& sh = out[(size_t)i];
        sh.seg_off.assign(segs.size(), 0); sh.spliced_off.assign(spliced.size(), 0);
        if (i > 0) {
            const std::vector<int64_t>& o = offs[(size_t)i - 1];
            sh.begin_id = ids[(size_t)i - 1];
            sh.read_off = o[0];
            for (size_t s = 0; s < spliced.size(); ++s) sh.spliced_off[s] = o[1 + (spliced.size() - 1 - s)];
            for (size_t s = 0; s < segs.size(); ++s) sh.seg_off[s] = o[1 + spliced.size() + (segs.size() - 1 - s)];
        }
        sh.end_id = i + 1 < want ? ids[(size_t)i] : ~0ull;
    }
    for (int i = 0; i < want; ++i) {
        Shard& sh = out[(size_t)i];
        sh.seg_end.assign(segs.size(), -1);
        // lists = {reads, spliced maps last..first, contig maps last..first}
        if (i + 1 < want) for (size_t s = 0; s < segs.size(); ++s) sh.seg_end[s] = shard_end_offset(lists[1 + spliced.size() + (segs.size() - 1 - s)], sh.end_id);
        if (i + 1 < want) sh.read_end = shard_end_offset(lists[0], sh.end_id);
    }
    return out;
}

// names, bases and qualities of a read from its own BAM record (bam1_t layout after block_size)
static void read_from_raw(const Read& rd, Read& out) {
    static const char nt16[] = "=ACMGRSVTWYHKDBN";
    const uint8_t* d = rd.raw;
    uint32_t w2, w3, lseq;
    memcpy(&w2, d + 8, 4); memcpy(&w3, d + 12, 4); memcpy(&lseq, d + 16, 4);
    const uint32_t l_rn = w2 & 0xFF, n_cig = w3 & 0xFFFF;
    out.id = rd.id; out.raw = rd.raw;
    out.name.assign((const char*)d + 32, l_rn ? l_rn - 1 : 0);
    const uint8_t* sq = d + 32 + l_rn + 4 * n_cig;
    const uint8_t* ql = sq + ((lseq + 1) >> 1);
    out.seq.resize(lseq); out.qual.resize(lseq);
    for (uint32_t k = 0; k < lseq; ++k) { out.seq[k] = nt16[(sq[k >> 1] >> ((k & 1) ? 0 : 4)) & 0xF]; out.qual[k] = (char)(ql[k] + 33); }
}

// print_bamhit for a plain (one-record) alignment whose read came as a BAM record: name, packed bases and qualities are copied
// (reversed / complemented nibble-wise for an antisense alignment) instead of going through strings.  Byte for byte what
// BamWriter::encode writes for the same record; false = take the general path.
static bool encode_plain_from_raw(const BamWriter& bw, const RefTable& rt, const thj_aln& a, const Read& rd, int rlen, int indel, bool spliced,
                                  std::vector<uint8_t>& d, std::vector<uint32_t>& sizes, std::vector<long>& rids) {
    static const uint8_t bamop[16] = {0, 0, 0, 1, 1, 2, 2, 0, 0, 0, 0, 3, 3, 4, 5, 6};
    static const uint8_t comp16[16] = {15, 8, 4, 15, 2, 15, 15, 15, 1, 15, 15, 15, 15, 15, 15, 15};    // A<->T, C<->G, anything else N
    const uint8_t* r = rd.raw;
    uint32_t w2, w3, lseq;
    memcpy(&w2, r + 8, 4); memcpy(&w3, r + 12, 4); memcpy(&lseq, r + 16, 4);
    const uint32_t l_rn = w2 & 0xFF, n_cig_in = w3 & 0xFFFF;
    if ((int)lseq != rlen || l_rn == 0) return false;
    const uint8_t* sq = r + 32 + l_rn + 4 * n_cig_in;
    const uint8_t* ql = sq + ((lseq + 1) >> 1);
    const bool anti = (a.flags & THJ_HIT_ANTISENSE) != 0;
    const size_t at = d.size();
    const int32_t tid = bw.tid_of(rt.names[a.ref_id - 1]);
    const int32_t pos = a.left + 1 <= 0 ? -1 : a.left;
    int end = pos;
    for (int i = 0; i < a.n_cigar; ++i) { const uint32_t c = a.cigar[i], op = bamop[c >> 28]; if (op == 0 || op == 2 || op == 3) end += (int)(c & 0x0FFFFFFF); }
    const uint32_t bin = (uint32_t)reg2bin(pos, a.n_cigar == 0 ? pos + 1 : end);
    const size_t seq_b = (lseq + 1) >> 1;
    d.resize(at + 36 + l_rn + 4 * (size_t)a.n_cigar + seq_b + lseq);
    uint8_t* o = d.data() + at;
    auto w32 = [&](size_t off, uint32_t v) { memcpy(o + off, &v, 4); };
    w32(4, (uint32_t)tid); w32(8, (uint32_t)pos); w32(12, (bin << 16) | (255u << 8) | l_rn);
    w32(16, ((anti ? 0x10u : 0u) << 16) | (uint32_t)a.n_cigar); w32(20, lseq); w32(24, (uint32_t)-1); w32(28, (uint32_t)-1); w32(32, 0);
    memcpy(o + 36, r + 32, l_rn);
    uint8_t* oc = o + 36 + l_rn;
    for (int i = 0; i < a.n_cigar; ++i) { const uint32_t v = ((a.cigar[i] & 0x0FFFFFFF) << 4) | bamop[a.cigar[i] >> 28]; memcpy(oc + 4 * i, &v, 4); }
    uint8_t* os = oc + 4 * (size_t)a.n_cigar;
    uint8_t* oq = os + seq_b;
    if (!anti) {                                      // (decoding a nibble to its letter and encoding it again is the identity)
        memcpy(os, sq, seq_b);
        if (lseq & 1) os[seq_b - 1] &= 0xF0;
        memcpy(oq, ql, lseq);
    } else {                                          // reverse_complement (reads.cpp:189-207): anything but A C G T becomes N
        memset(os, 0, seq_b);
        for (uint32_t k = 0; k < lseq; ++k) {
            const uint32_t j = lseq - 1 - k;
            const uint8_t nib = (sq[j >> 1] >> ((j & 1) ? 0 : 4)) & 0xF;
            os[k >> 1] |= (uint8_t)(comp16[nib] << ((k & 1) ? 0 : 4));
            oq[k] = ql[j];
        }
    }
    // aux: AS XM XO XG MD NM [XS] (add_aux, common.cpp:1092-1173: the smallest integer type that holds the value)
    auto put_int = [&](char t0, char t1, long long x) {
        d.push_back((uint8_t)t0); d.push_back((uint8_t)t1);
        if (x < 0) {
            if (x >= -127) { d.push_back('c'); d.push_back((uint8_t)(int8_t)x); }
            else if (x >= -32767) { d.push_back('s'); int16_t v = (int16_t)x; uint8_t b[2]; memcpy(b, &v, 2); d.insert(d.end(), b, b + 2); }
            else { d.push_back('i'); uint32_t v = (uint32_t)(int32_t)x; uint8_t b[4]; memcpy(b, &v, 4); d.insert(d.end(), b, b + 4); }
        } else {
            if (x <= 255) { d.push_back('C'); d.push_back((uint8_t)x); }
            else if (x <= 65535) { d.push_back('S'); uint16_t v = (uint16_t)x; uint8_t b[2]; memcpy(b, &v, 2); d.insert(d.end(), b, b + 2); }
            else { d.push_back('I'); uint32_t v = (uint32_t)x; uint8_t b[4]; memcpy(b, &v, 4); d.insert(d.end(), b, b + 4); }
        }
    };
    put_int('A', 'S', (int)a.AS); put_int('X', 'M', (int)a.XM); put_int('X', 'O', (int)a.XO); put_int('X', 'G', (int)a.XG);
    d.push_back('M'); d.push_back('D'); d.push_back('Z'); d.insert(d.end(), a.md, a.md + a.md_len); d.push_back(0);
    put_int('N', 'M', (int)a.mismatches + indel);
    if (spliced) { d.push_back('X'); d.push_back('S'); d.push_back('A'); d.push_back((a.flags & THJ_HIT_ANTISENSE_SPLICE) ? '-' : '+'); }
    const uint32_t bs = (uint32_t)(d.size() - at - 4);
    memcpy(d.data() + at, &bs, 4);
    sizes.push_back((uint32_t)(d.size() - at));
    long rid = 0;                                     // atol(qname)
    {
        const char* q = (const char*)r + 32;
        bool neg = false;
        size_t k = 0;
        while (k + 1 < l_rn && (q[k] == ' ' || q[k] == '\t')) ++k;
        if (q[k] == '-') { neg = true; ++k; } else if (q[k] == '+') ++k;
        for (; k + 1 < l_rn && q[k] >= '0' && q[k] <= '9'; ++k) rid = rid * 10 + (q[k] - '0');
        if (neg) rid = -rid;
    }
    rids.push_back(rid);
    return true;
}

// print_bamhit (bwt_map.cpp:1888-2093) for one alignment: one record, or -- a fusion alignment -- the two partial records of
// extract_partial_hits (:2148-2347), each carrying the whole alignment in XF:Z.  Appends (size, read id) per record.
static void encode_aln(const BamWriter& bw, const RefTable& rt, const thj_aln& a, const Read& rd, std::vector<uint8_t>& d,
                       std::vector<uint32_t>& sizes, std::vector<long>& rids) {
    int rlen = 0, indel = 0; bool spliced = false;
    int fi = -1;
    for (int k = 0; k < a.n_cigar; ++k) {
        uint32_t op = a.cigar[k] >> 28, len = a.cigar[k] & 0x0FFFFFFF;
        if (op == 1 || op == 2 || op == 3 || op == 4 || op == 13) rlen += (int)len;
        if (op >= 3 && op <= 6) indel += (int)len;
        if (op == 11 || op == 12) spliced = true;
        if (op >= THJ_CIG_FUSION_FF && op <= THJ_CIG_FUSION_RR && fi < 0) fi = k;
    }
    if (rd.raw && fi < 0 && a.md_len != THJ_MD_ON_HOST && encode_plain_from_raw(bw, rt, a, rd, rlen, indel, spliced, d, sizes, rids)) return;
    Read tmp;
    const Read& rdx = rd.raw && rd.seq.empty() ? (read_from_raw(rd, tmp), tmp) : rd;
    std::string seq = rdx.seq, qual = rdx.qual;
    seq.resize((size_t)rlen); qual.resize((size_t)rlen);
    uint32_t flag = 0;
    if (a.flags & THJ_HIT_ANTISENSE) { flag |= 0x10; reverse_complement(seq); std::reverse(qual.begin(), qual.end()); }
    const uint32_t ref_id2 = fi >= 0 ? a.cigar[15] : a.ref_id;
    std::vector<std::string> aux;
    aux.push_back("AS:i:" + std::to_string((int)a.AS));
    aux.push_back("XM:i:" + std::to_string((int)a.XM));
    aux.push_back("XO:i:" + std::to_string((int)a.XO));
    aux.push_back("XG:i:" + std::to_string((int)a.XG));
    if (a.md_len == THJ_MD_ON_HOST) {                       // longer than a device record holds: rebuilt here from the same inputs
        char md[2048];
        const std::string& ref = const_cast<RefTable&>(rt).text(a.ref_id);
        const std::string& ref2 = const_cast<RefTable&>(rt).text(ref_id2);
        const int n = fi >= 0 ? thj_md_string2(ref.data(), (int64_t)ref.size(), ref2.data(), (int64_t)ref2.size(), seq.data(), (int32_t)seq.size(), a.left,
                                               a.cigar, a.n_cigar, md, (int32_t)sizeof md)
                              : thj_md_string(ref.data(), (int64_t)ref.size(), seq.data(), (int32_t)seq.size(), a.left, a.cigar, a.n_cigar, md, (int32_t)sizeof md);
        if (n < 0) die("Error: %s\n", thj_last_error());
        aux.push_back("MD:Z:" + std::string(md, (size_t)n));
    } else aux.push_back("MD:Z:" + std::string(a.md, a.md_len));
    aux.push_back("NM:i:" + std::to_string((int)a.mismatches + indel));
    if (spliced) aux.push_back(std::string("XS:A:") + ((a.flags & THJ_HIT_ANTISENSE_SPLICE) ? '-' : '+'));
    const long rid = atol(rdx.name.c_str());
    size_t before = d.size();
    if (fi < 0) {
        bw.encode(d, rdx.name, flag, rt.names[a.ref_id - 1], a.left + 1, a.cigar, a.n_cigar, seq, qual, aux);
        sizes.push_back((uint32_t)(d.size() - before)); rids.push_back(rid);
        return;
    }
    // ---- fusion alignment
    static const char letter[16] = {0, 'M', 'm', 'I', 'i', 'D', 'd', 'F', 'F', 'F', 'F', 'N', 'n', 'S', 0, 0};
    const uint32_t fdir = a.cigar[fi] >> 28;
    std::string full;
    int right = a.left, fusion_left = -1, fusion_right = -1;
    size_t left_part_len = 0;
    for (int k = 0; k < a.n_cigar; ++k) {
        const uint32_t op = a.cigar[k] >> 28, len = a.cigar[k] & 0x0FFFFFFF;
        full += std::to_string(op >= 7 && op <= 10 ? len + 1 : len); full += letter[op];
        if (op == 1 || op == 11 || op == 5) right += (int)len;
        else if (op == 2 || op == 12 || op == 6) right -= (int)len;
        else if (op >= 7 && op <= 10) { fusion_left = (op == 7 || op == 8) ? right - 1 : right + 1; fusion_right = right = (int)len; }
        if (k < fi && (op == 1 || op == 2 || op == 3 || op == 4)) left_part_len += len;
    }
    auto upper = [](uint32_t c) { const uint32_t op = c >> 28; return (op == 2 || op == 4 || op == 6 || op == 12) ? (((op - 1) << 28) | (c & 0x0FFFFFFF)) : c; };
    uint32_t c1[16], c2[16]; int n1 = 0, n2 = 0;
    if (fdir == 7 || fdir == 8) for (int k = 0; k < fi; ++k) c1[n1++] = upper(a.cigar[k]);
    else for (int k = fi - 1; k >= 0; --k) c1[n1++] = upper(a.cigar[k]);
    if (fdir == 7 || fdir == 9) for (int k = fi + 1; k < a.n_cigar; ++k) c2[n2++] = upper(a.cigar[k]);
    else for (int k = a.n_cigar - 1; k > fi; --k) c2[n2++] = upper(a.cigar[k]);
    if (left_part_len > seq.size()) left_part_len = seq.size();
    std::string seq1 = seq.substr(0, left_part_len), qual1 = qual.substr(0, left_part_len);
    std::string seq2 = seq.substr(left_part_len), qual2 = qual.substr(left_part_len);
    if (fdir == 9 || fdir == 10) { reverse_complement(seq1); std::reverse(qual1.begin(), qual1.end()); }
    if (fdir == 8 || fdir == 10) { reverse_complement(seq2); std::reverse(qual2.begin(), qual2.end()); }
    const int left1 = (fdir == 7 || fdir == 8) ? a.left : fusion_left;
    const int left2 = (fdir == 7 || fdir == 9) ? fusion_right : right + 1;
    const std::string& n1s = rt.names[a.ref_id - 1];
    const std::string& n2s = rt.names[ref_id2 - 1];
    const std::string xf = " " + n1s + "-" + n2s + " " + std::to_string(a.left + 1) + " " + full + " " + seq + " " + qual;
    aux.push_back("XF:Z:1" + xf);
    bw.encode(d, rdx.name, flag, n1s, left1 + 1, c1, n1, seq1, qual1, aux);
    sizes.push_back((uint32_t)(d.size() - before)); rids.push_back(rid);
    before = d.size();
    aux.back() = "XF:Z:2" + xf;
    bw.encode(d, rdx.name, flag, n2s, left2 + 1, c2, n2, seq2, qual2, aux);
    sizes.push_back((uint32_t)(d.size() - before)); rids.push_back(rid);
}

static void encode_batch(const BamWriter& bw, const RefTable& rt, const thj_aln* alns, const size_t n, const std::vector<Read>& reads, int threads,
                         BamWriter::Encoded& e) {
    int T = threads;
    if ((size_t)T > n / 256 + 1) T = (int)(n / 256 + 1);
    std::vector<std::vector<uint8_t>> part((size_t)T);
    std::vector<std::vector<uint32_t>> psize((size_t)T);
    std::vector<std::vector<long>> prid((size_t)T);
    auto work = [&](int t) {
        const size_t a = n * (size_t)t / (size_t)T, b = n * (size_t)(t + 1) / (size_t)T;
        std::vector<uint8_t>& d = part[(size_t)t];
        d.reserve((b - a) * 256);
        psize[(size_t)t].reserve(b - a); prid[(size_t)t].reserve(b - a);
        for (size_t i = a; i < b; ++i) encode_aln(bw, rt, alns[i], reads[alns[i].read_idx], d, psize[(size_t)t], prid[(size_t)t]);
    };
    if (T > 1) { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
    else work(0);
    size_t total = 0, nrec = 0;
    for (auto& d : part) total += d.size();
    for (auto& v : psize) nrec += v.size();
    e.bytes.reserve(total); e.size.reserve(nrec); e.rid.reserve(nrec);
    for (size_t t = 0; t < (size_t)T; ++t) {
        e.bytes.insert(e.bytes.end(), part[t].begin(), part[t].end()); std::vector<uint8_t>().swap(part[t]);
        e.size.insert(e.size.end(), psize[t].begin(), psize[t].end());
        e.rid.insert(e.rid.end(), prid[t].begin(), prid[t].end());
    }
}

// records of one shard on their way to the writer
struct OutShard {
    std::mutex mu; std::condition_variable cv;
    std::deque<BamWriter::Prepared> q;
    bool done = false;
};

static int real_main(int argc, char** argv) {
    fprintf(stderr, "long_spanning_reads (MI355X-native, %s)\n--------------------------------------------\n", thj_version());
    Opts o;
    int rc = parse_options(argc, argv, o, print_usage);
    if (rc) return rc;
    std::vector<std::string> pos;
    for (int i = optind; i < argc; ++i) pos.push_back(argv[i]);
    if (pos.size() < 8) { print_usage(); return 1; }
    if (o.color) die("Error: colour-space reads are not supported by this build\n");
    o.p.fusion_search = o.fusion_search ? 1 : 0;
    std::vector<std::string> spliced_segs;
    if (pos.size() >= 9) spliced_segs = split(pos[8], ',');
    std::vector<std::string> segs = split(pos[7], ',');
    if (segs.empty()) { fprintf(stderr, "No hits to process, exiting\n"); return 0; }           // long_spanning_reads.cpp:2883-2887

    // the reference is read on its own thread(s) while the HIP runtime starts (the device count below is its first call, ~50 ms)
    RefTable rt;
    rt.load_sam_header(o.sam_header);
    fprintf(stderr, "Loading reference sequences...\n");
    std::future<void> fasta_loaded = std::async(std::launch::async, [&rt, &pos]() { rt.load_reference(pos[0], pos[6]); });
    std::vector<std::unique_ptr<Gpu>> gpus;
    {
        int n_dev = 1, first = 0;
        if (getenv("THJ_DEVICE")) first = atoi(getenv("THJ_DEVICE"));
        else { n_dev = thj_device_count(); if (n_dev < 1) die("Error: %s\n", thj_last_error()); if (getenv("THJ_GPUS") && atoi(getenv("THJ_GPUS")) >= 1) n_dev = std::min(n_dev, atoi(getenv("THJ_GPUS"))); }
        // THJ_CTX_PER_GPU=k: k contexts (streams, arenas, tables) on every device, each a rank of its own -- a shard's host-to-device
        // copies and stream round trips then overlap another shard's kernels on the same GPU
        // (default: 2 on a single GPU -- measured 2.4 -> 2.0 s for segment_juncs on 8 M pairs -- and 1 per device on several: a
        // communicator is either all-RCCL or all-loopback)
        // (three here: a shard's inflate launch is ~1500 members, a quarter of what the GPU holds, so three contexts' launches overlap:
        // 1.22 -> 1.09 s per side on 10 M pairs; segment_juncs, with ten times larger shards, keeps two)
        // (... when there is enough to overlap: a context costs ~50 ms to start -- its stream, its first launches and allocations -- and a
        // side of 10 M pairs, 1.4 GB of maps, is through in 0.4 s: two contexts there, 2.22-2.26 s against 2.33-2.48 for the three
        // processes; at 40 M pairs three, 4.14-4.39 s against 4.50-4.62.  tools/scratch/r05_ctx_ab.sh)
        int64_t in_bytes = 0;
        { struct stat sb; for (const std::string& f : segs) if (!stat(f.c_str(), &sb)) in_bytes += (int64_t)sb.st_size; if (!stat(pos[1].c_str(), &sb)) in_bytes += (int64_t)sb.st_size; }
        int per = getenv("THJ_CTX_PER_GPU") ? atoi(getenv("THJ_CTX_PER_GPU")) : (n_dev > 1 ? 1 : (in_bytes > (3ll << 30) ? 3 : 2));
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
                if (!getenv("THJ_NO_WARM") && thj_ctx_warm(c, THJ_WARM_SPAN | THJ_WARM_INGEST | THJ_WARM_BAMOUT)) die("Error: %s\n", thj_last_error());
                if (getenv("THJ_TIMING")) fprintf(stderr, "[timing] a device context ready after       %8.3f s of the process\n", std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count() - g_timer.wall0);
                return c;
            });
        }
    }
    const int n_gpus = (int)gpus.size();

    fasta_loaded.get();
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
    // ---- --fusion-search: the .fusions lists -> std::set<Fusion> (:2998-3040, fusions.h:44-71)
    std::vector<thj_span_fusion> fusions;
    if (o.fusion_search) {
        for (auto& fn : split(pos[5], ',')) {
            FILE* f = fopen(fn.c_str(), "r");
            if (!f) continue;
            char buf[2048];
            while (fgets(buf, sizeof buf, f)) {
                char* nl = strrchr(buf, '\n'); if (nl) *nl = 0;
                std::vector<std::string> t;                      // strsep: empty fields count
                { const char* b0 = buf; for (const char* q = buf;; ++q) if (*q == '\t' || !*q) { t.emplace_back(b0, q); if (!*q) break; b0 = q + 1; } }
                if (t.size() < 5) die("Error: malformed insertion coordinate record\n");
                uint32_t dir = THJ_CIG_FUSION_FF;
                if (t[4] == "fr") dir = THJ_CIG_FUSION_FR; else if (t[4] == "rf") dir = THJ_CIG_FUSION_RF; else if (t[4] == "rr") dir = THJ_CIG_FUSION_RR;
                fusions.push_back({rt.get_id(t[0]), rt.get_id(t[2]), (uint32_t)atoi(t[1].c_str()), (uint32_t)atoi(t[3].c_str()), dir});
            }
            fclose(f);
        }
        auto fl = [](const thj_span_fusion& a, const thj_span_fusion& b) {
            if (a.ref_id1 != b.ref_id1) return a.ref_id1 < b.ref_id1;
            if (a.ref_id2 != b.ref_id2) return a.ref_id2 < b.ref_id2;
            if (a.left != b.left) return a.left < b.left;
            if (a.right != b.right) return a.right < b.right;
            return a.dir < b.dir;
        };
        std::sort(fusions.begin(), fusions.end(), fl);
        fusions.erase(std::unique(fusions.begin(), fusions.end(), [&](const thj_span_fusion& a, const thj_span_fusion& b) { return !fl(a, b) && !fl(b, a); }), fusions.end());
    }
    for (auto& f : segs) register_targets(f, rt);
    rt.freeze();
    {   // junctions on contigs the device genome does not know cannot be closed anyway: drop them
        std::vector<thj_junction> keep;
        for (auto& j : juncs) if (j.ref_id >= 1 && j.ref_id <= rt.names.size() && (j.right - j.left) < (1u << 29)) keep.push_back(j);
        juncs.swap(keep);
    }
    g_timer.lap("junction / indel lists");

    // HIP start-up (0.15-0.25 s) runs beside the first shards' ingest: a GPU's context is picked up -- with the genome and
    // the sets going up then -- the first time a worker needs that device (under the GPU's lock).
    auto device_ready = [&](Gpu& g) -> thj_ctx* {
        if (g.ctx) return g.ctx;
        g.ctx = g.fut.get();
        rt.upload(g.ctx);
        if (thj_span_sets_upload(g.ctx, juncs.data(), (int64_t)juncs.size(), ins_tab.data(), (int64_t)ins_tab.size() / 4)) die("Error: %s\n", thj_last_error());
        if (o.fusion_search && thj_span_fusions_upload(g.ctx, fusions.data(), (int64_t)fusions.size())) die("Error: %s\n", thj_last_error());
        if (thj_span_reset_async(g.ctx)) die("Error: %s\n", thj_last_error());
        return g.ctx;
    };

    const int nseg = (int)segs.size();
    const std::string out = pos[6];
    // BAM segment maps mapped for the device-side ingest (contig maps only: junction-db maps go through the spliced hit factory
    // on the host, and then so does everything)
    std::vector<std::unique_ptr<BamFile>> bams;
    bool dev_ingest = !getenv("THJ_HOST_INGEST") && spliced_segs.empty();
    for (int s = 0; s < nseg && dev_ingest; ++s) { bams.emplace_back(new BamFile()); if (!bams.back()->open(segs[(size_t)s], rt)) dev_ingest = false; }
    // the reads file too when it is an (unaligned) BAM: its members are then inflated on the device with the maps' and the read
    // records come back ready to be copied into the output (THJ_HOST_READS=1: the host ReadStream instead)
    BamFile reads_bam;
    const bool dev_reads = dev_ingest && !getenv("THJ_HOST_READS") && reads_bam.open(pos[1], rt);
    // ... and with the reads on the device the records are built and deflated there too (thj_span_bam_encode, thj_bgzf_deflate): the
    // host wraps the members and writes them.  THJ_HOST_BAM=1 (or a zlib level asked for with THJ_BGZF_LEVEL): the host encoder.
    const bool dev_out_wanted = dev_reads && !getenv("THJ_HOST_BAM") && !getenv("THJ_BGZF_LEVEL");
    // ---- the shard plan.  -p N: the reference's N ranges, one output file each.  One output file: our own number of shards,
    // written in order.
    const int hw = effective_cpus();
    int workers = getenv("THJ_WORKERS") ? atoi(getenv("THJ_WORKERS")) : std::max(1, std::min(32, hw * 3 / 4));
    if (workers < 1) workers = 1;
    int parts = o.num_threads > 1 ? o.num_threads : 1;
    std::vector<Shard> shards;
    if (parts > 1) {
        shards = plan(pos[1], segs, spliced_segs, parts);
        if ((int)shards.size() != parts) { shards.resize(1); shards[0] = Shard(); shards[0].seg_off.assign(segs.size(), 0); shards[0].spliced_off.assign(spliced_segs.size(), 0); parts = 1; }   // not enough data: one thread (:2992-2993)
    }
    if (parts == 1) {
        // shards of ~16 MB of compressed input (~110 k reads of 100 bases with four segment maps): measured best for the pipeline below
        // (10 M pairs: 24 MB 1.02-1.15 s per side, 16 MB 0.83-0.93, 12 MB 0.80-0.92, 8 MB 0.96-1.11, 48 MB 1.49) -- and at least four per host worker
        int n_shards = getenv("THJ_SHARD_MB") ? 1 : 4 * workers;
        if (getenv("THJ_SHARDS")) n_shards = atoi(getenv("THJ_SHARDS"));
        else {
            uint64_t bytes = 0;
            struct stat st;
            for (auto& f : segs) if (stat(f.c_str(), &st) == 0) bytes += (uint64_t)st.st_size;
            if (stat(pos[1].c_str(), &st) == 0) bytes += (uint64_t)st.st_size;
            const uint64_t shard_mb = getenv("THJ_SHARD_MB") && atoi(getenv("THJ_SHARD_MB")) >= 1 ? (uint64_t)atoi(getenv("THJ_SHARD_MB")) : 16;
            const uint64_t by_size = bytes / (shard_mb << 20);
            if (by_size > (uint64_t)n_shards) n_shards = (int)std::min<uint64_t>(by_size, 4096);
        }
        shards = plan(pos[1], segs, spliced_segs, n_shards);
    }
    const size_t S = shards.size();
    fprintf(stderr, "\t%d read-id shard%s, %d host CPUs, %d GPU context%s\n", (int)S, S > 1 ? "s" : "", hw, n_gpus, n_gpus > 1 ? "s" : "");

    std::vector<std::unique_ptr<BamWriter>> bws;
    if (parts == 1) {
        bws.emplace_back(new BamWriter());
        if (!bws[0]->open(out, rt, out + ".index")) die("Error: could not create BAM file %s!\n", out.c_str());
    } else {
        for (int k = 0; k < parts; ++k) {                      // long_spanning_reads.cpp:3056-3064
            std::string fn = out.substr(0, out.size() >= 4 ? out.size() - 4 : out.size()) + std::to_string(k) + ".bam";
            bws.emplace_back(new BamWriter());
            if (!bws.back()->open(fn, rt, fn + ".index")) die("Error: could not create BAM file %s!\n", fn.c_str());
        }
    }
    std::vector<std::unique_ptr<OutShard>> outq;
    for (size_t k = 0; k < S; ++k) outq.emplace_back(new OutShard());
    std::mutex win_mu; std::condition_variable win_cv;
    size_t writer_pos = 0;                                     // shards before this one are on disk
    const size_t LOOKAHEAD = getenv("THJ_LOOKAHEAD") ? (size_t)atoll(getenv("THJ_LOOKAHEAD")) : (size_t)std::max(workers + 2, 32);      // shards in flight (memory bound)
    const size_t batch_reads = getenv("THJ_BATCH_READS") ? (size_t)atoll(getenv("THJ_BATCH_READS")) : (size_t)1 << 19;
    const int enc_threads = S == 1 ? host_threads() : 1;       // many shards: the workers are the parallelism

    // The run is a pipeline of three kinds of threads (one output file; with -p N every part writes its own file from its feeder):
    //   feeders   a few per GPU context: a shard's device work (ingest, stitch, download), shard after shard -- they wait on the
    //             GPU, not on the CPU;
    //   the pool  record encoding and BGZF deflate as independent jobs: the CPU-heavy part, never waiting for anything;
    //   writer    the main thread: appends the deflated members in shard order, computes `.index` lines.
    // Where BGZF members end depends on the bytes still open from the shard before, so shards are PLANNED in output order -- cheap,
    // record sizes only, done by whichever thread delivers the encoded shard that was missing (on_encoded) -- and DEFLATED
    // independently afterwards (BamWriter::plan / compress / commit).  Before this split every worker did all of it for its shard
    // and the run moved in waves: all workers on the GPU's lock, then all deflating while the GPU idled (THJ_TRACE, tools/lsr_trace.py).
    struct Pool {
        std::mutex mu; std::condition_variable cv; std::deque<std::function<void()>> q; bool stop = false; std::vector<std::thread> th;
        void start(int n) { for (int t = 0; t < n; ++t) th.emplace_back([this] { for (;;) { std::function<void()> f; { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return stop || !q.empty(); }); if (q.empty()) return; f = std::move(q.front()); q.pop_front(); } f(); } }); }
        void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv.notify_one(); }
        void finish() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); for (auto& t : th) t.join(); th.clear(); }
    } pool;
    struct ShardOut { BamWriter::Encoded e; bool ready = false; BamWriter::Prepared p; };
    std::vector<ShardOut> so(S);
    std::mutex plan_mu;
    size_t plan_next = 0;                                      // the first shard not planned yet
    std::vector<uint8_t> plan_carry;                           // bytes of the open block after the last planned shard
    auto deflate_shard = [&](size_t j) {
        BamWriter::compress(so[j].p);
        trace(j, "deflated");
        OutShard& oq = *outq[j];
        std::lock_guard<std::mutex> lk(oq.mu);
        oq.q.push_back(std::move(so[j].p));
        oq.done = true;
        oq.cv.notify_all();
    };
    // shard k's records are encoded (all of them: one call per shard, an empty one for a shard without records)
    auto on_encoded = [&](size_t k, BamWriter::Encoded&& e) {
        std::vector<size_t> planned;
        {
            std::lock_guard<std::mutex> lk(plan_mu);
            so[k].e = std::move(e); so[k].ready = true;
            while (plan_next < S && so[plan_next].ready) {
                if (so[plan_next].p.device) BamWriter::plan_device(plan_carry, so[plan_next].p);
                else BamWriter::plan(plan_carry, std::move(so[plan_next].e), so[plan_next].p);
                trace(plan_next, "planned");
                planned.push_back(plan_next++);
            }
        }
        for (size_t i = 1; i < planned.size(); ++i) pool.submit([&deflate_shard, j = planned[i]] { deflate_shard(j); });
        if (!planned.empty()) deflate_shard(planned[0]);
    };

    // shard k's records were encoded and deflated on the device: the planner only closes what the shard before left open
    auto on_device_shard = [&](size_t k, BamWriter::Prepared&& dp) {
        std::vector<size_t> planned;
        {
            std::lock_guard<std::mutex> lk(plan_mu);
            so[k].p = std::move(dp); so[k].ready = true;
            while (plan_next < S && so[plan_next].ready) {
                if (so[plan_next].p.device) BamWriter::plan_device(plan_carry, so[plan_next].p);
                else BamWriter::plan(plan_carry, std::move(so[plan_next].e), so[plan_next].p);
                trace(plan_next, "planned");
                planned.push_back(plan_next++);
            }
        }
        for (size_t i = 1; i < planned.size(); ++i) pool.submit([&deflate_shard, j = planned[i]] { deflate_shard(j); });
        if (!planned.empty()) deflate_shard(planned[0]);
    };
    const bool dev_out = dev_out_wanted && parts == 1;
    std::vector<int32_t> tid_of_ref(rt.names.size());
    if (dev_out) for (size_t i = 0; i < rt.names.size(); ++i) tid_of_ref[i] = bws[0]->tid_of(rt.names[i]);
    std::atomic<long long> dev_out_shards{0}, host_out_shards{0};

    // JoinSegmentsWorker (long_spanning_reads.cpp:2669-2845) for one shard
    auto run_shard = [&](size_t k) {
        const long long t_shard = WorkClock::now();
        struct AtExit { long long t; ~AtExit() { g_work.add(0, t); } } at_exit{t_shard};
        trace(k, "start");
        const Shard& sh = shards[k];
        Gpu& gpu = *gpus[k % (size_t)n_gpus];
        const BamWriter& enc_bw = *bws[parts == 1 ? 0 : k];
        // ---- device-side ingest of the segment maps (thj_ingest_span_hits): the host reads the shard's reads only
        if (dev_ingest) {
            std::vector<thj_bam_piece> segp;
            for (int s = 0; s < nseg; ++s) segp.push_back(bams[(size_t)s]->piece(sh.seg_off[(size_t)s], sh.seg_end.empty() ? -1 : sh.seg_end[(size_t)s]));
            const uint32_t b_id = sh.begin_id > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sh.begin_id, e_id = sh.end_id > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sh.end_id;
            thj_span_batch* dev = nullptr; uint32_t* ids = nullptr; int64_t n = 0;
            uint8_t* rinfl = nullptr; int64_t rinfl_bytes = 0; uint32_t* rloc = nullptr;
            int rc;
            // the shard's compressed pieces go into one page-locked buffer first -- here, beside the other feeders and outside the GPU's
            // lock: from the mapped files the copy up runs through the runtime's staging buffers at ~3 GB/s while the pool keeps the
            // CPUs busy, from page-locked memory it is DMA
            thj_bam_piece rp = dev_reads ? reads_bam.piece(sh.read_off, sh.read_end) : thj_bam_piece{};
            std::vector<std::pair<const BamFile*, thj_bam_piece*>> to_stage;
            for (int s = 0; s < nseg; ++s) to_stage.emplace_back(bams[(size_t)s].get(), &segp[(size_t)s]);
            if (dev_reads) to_stage.emplace_back(&reads_bam, &rp);
            uint8_t* stage = stage_pieces(to_stage);
            {
                const long long tw = WorkClock::now();
                std::lock_guard<std::mutex> lk(gpu.mu);
                g_work.add(1, tw);
                trace(k, "ingest_begin");
                const long long td = WorkClock::now();
                if (dev_reads && dev_out) rc = thj_ingest_span_batch(device_ready(gpu), &o.p, nseg, segp.data(), &rp, b_id, e_id, &dev, &ids, &n, nullptr, nullptr, nullptr);
                else if (dev_reads) {
                    rc = thj_ingest_span_batch(device_ready(gpu), &o.p, nseg, segp.data(), &rp, b_id, e_id, &dev, &ids, &n, &rinfl, &rinfl_bytes, &rloc);
                } else
                    rc = thj_ingest_span_hits(device_ready(gpu), &o.p, nseg, segp.data(), b_id, e_id, &dev, &ids, &n);
                g_work.add(2, td);
                trace(k, "ingest_end");
            }
            thj_pinned_free(stage);
            if (rc == THJ_OK) {
                if (dev) {
                    std::vector<Read> batch_rd((size_t)n);
                    int W = 1, stride = 0;
                    std::vector<uint64_t> planes; std::vector<uint16_t> lens; std::vector<uint8_t> q;
                    // the rows' own BAM records, where the host encoder copies names, bases and qualities from
                    auto rows_from_raw = [&]() {
                        for (int64_t r = 0; r < n; ++r) {
                            const uint32_t loc = rloc[r];
                            batch_rd[(size_t)r].raw = rinfl + ((size_t)(loc >> 16) << 16) + (loc & 0xFFFFu) + 4;
                        }
                        free(rloc); rloc = nullptr;
                    };
                    if (dev_reads) {
                        for (int64_t r = 0; r < n; ++r) batch_rd[(size_t)r].id = ids[r];
                        free(ids);
                        if (!dev_out) rows_from_raw();
                    } else {
                    ReadStream reads;
                    if (!reads.open(pos[1], o.zpacker, sh.read_off)) die("Error: cannot open %s for reading\n", pos[1].c_str());
                    std::vector<int64_t> read_off(1, 0); std::string bases, quals; size_t max_len = 0;
                    for (int64_t r = 0; r < n; ++r) {
                        Read& rd = batch_rd[(size_t)r];
                        if (!reads.get(ids[r], rd)) die("Error: could not get read # %d from stream\n", (int)ids[r]);
                        bases += rd.seq; quals += rd.qual;
                        read_off.push_back((int64_t)bases.size());
                        if (rd.seq.size() > max_len) max_len = rd.seq.size();
                    }
                    free(ids);
                    W = (int)((max_len + 63) / 64); if (W < 1) W = 1;
                    planes.resize((size_t)n * 3 * W);
                    lens.resize((size_t)n);
                    if (thj_reads_pack(n, read_off.data(), bases.data(), W, planes.data(), lens.data())) die("Error: %s\n", thj_last_error());
                    stride = (int)((max_len + 3) / 4 * 4);
                    q.assign((size_t)n * stride, 0);
                    for (int64_t r = 0; r < n; ++r) memcpy(q.data() + (size_t)r * stride, quals.data() + read_off[(size_t)r], (size_t)(read_off[(size_t)r + 1] - read_off[(size_t)r]));
                    }
                    std::vector<thj_aln> alns;
                    thj_aln* alns_pinned = nullptr; int64_t n_alns = 0;       // one output file: the records come down into a page-locked buffer
                    bool on_device = false;                                    // the shard's records were encoded and deflated on the device
                    std::vector<uint32_t> dsize; std::vector<int64_t> drid; std::vector<size_t> dcuts; std::vector<uint32_t> dclen, dcrc;
                    uint8_t* dcomp = nullptr; int64_t dcomp_bytes = 0;
                    {
                        const long long tw = WorkClock::now();
                        std::lock_guard<std::mutex> lk(gpu.mu);
                        g_work.add(1, tw);
                        trace(k, "stitch_begin");
                        const long long td = WorkClock::now();
                        thj_ctx* ctx = device_ready(gpu);
                        if (!dev_reads && thj_span_batch_attach_reads(ctx, dev, W, stride, planes.data(), lens.data(), q.data())) die("Error: %s\n", thj_last_error());
                        int64_t na = 0;
                        for (int attempt = 0;; ++attempt) {   // THJ_ERETRY: a device pool was enlarged, the pass runs again
                            if (thj_span_reset_async(ctx)) die("Error: %s\n", thj_last_error());
                            if (thj_span_run_async(ctx, &o.p, dev)) die("Error: %s\n", thj_last_error());
                            const int frc = thj_span_finish(ctx, &na);
                            if (frc == THJ_ERETRY && attempt < 4) { trace(k, "stitch_retry"); continue; }
                            if (frc) die("Error: %s\n", thj_last_error());
                            break;
                        }
                        trace(k, "stitch_finished");
                        n_alns = na;
                        if (dev_out) {
                            // records and BGZF members on the device; the host gets record sizes, read ids and the deflated members
                            dsize.resize((size_t)na); drid.resize((size_t)na);
                            int64_t total = 0;
                            int erc = thj_span_bam_encode(ctx, dev, tid_of_ref.data(), (int32_t)tid_of_ref.size(), dsize.data(), drid.data(), &total);
                            if (erc == THJ_OK) {
                                trace(k, "bam_encoded");
                                BamWriter::plan_cuts_closed(dsize, dcuts);
                                std::vector<int64_t> ends(dcuts.begin(), dcuts.end());
                                dclen.resize(dcuts.size()); dcrc.resize(dcuts.size());
                                dcomp = (uint8_t*)thj_pinned_alloc(dcuts.size() * (size_t)65536 + 64);
                                if (!dcomp) die("Error: out of memory\n");
                                erc = thj_bgzf_deflate(ctx, (int64_t)dcuts.size(), ends.data(), dcomp, (int64_t)(dcuts.size() * (size_t)65536), dclen.data(), dcrc.data(), &dcomp_bytes);
                                trace(k, "bam_deflated");
                            }
                            if (erc == THJ_OK) on_device = true;
                            else if (erc != THJ_EFALLBACK) die("Error: %s\n", thj_last_error());
                            else {
                                static std::atomic<bool> told{false};
                                if (!told.exchange(true)) fprintf(stderr, "\tdevice-side BAM output not possible for a shard (%s); encoding on the host\n", thj_last_error());
                                thj_pinned_free(dcomp); dcomp = nullptr;
                                if (thj_span_batch_reads_host(ctx, dev, &rinfl, &rinfl_bytes, &rloc)) die("Error: %s\n", thj_last_error());
                                rows_from_raw();
                            }
                        }
                        if (on_device) {
                            if (thj_span_batch_free(ctx, dev)) die("Error: %s\n", thj_last_error());
                            g_work.add(2, td);
                            trace(k, "stitch_end");
                        } else
                        if (parts == 1) { alns_pinned = (thj_aln*)thj_pinned_alloc((size_t)(na ? na : 1) * sizeof(thj_aln)); if (!alns_pinned) die("Error: out of memory\n"); }
                        else alns.resize((size_t)na);
                        if (!on_device) {
                        trace(k, "stitch_resized");
                        if (na && thj_span_download(ctx, parts == 1 ? alns_pinned : alns.data())) die("Error: %s\n", thj_last_error());
                        trace(k, "stitch_downloaded");
                        if (thj_span_batch_free(ctx, dev)) die("Error: %s\n", thj_last_error());
                        g_work.add(2, td);
                        trace(k, "stitch_end");
                        }
                    }
                    if (on_device) {
                        // outside the GPU's lock: the members into their BGZF envelopes
                        ++dev_out_shards;
                        BamWriter::Prepared dp;
                        dp.device = true;
                        dp.size = std::move(dsize);
                        dp.rid.assign(drid.begin(), drid.end());
                        dp.cuts = std::move(dcuts);
                        dp.members.resize(dp.cuts.size());
                        size_t at = 0;
                        for (size_t m = 0; m < dp.cuts.size(); ++m) {
                            const size_t ulen = dp.cuts[m] - (m ? dp.cuts[m - 1] : 0);
                            BamWriter::wrap_member(dcomp + at, dclen[m], dcrc[m], (uint32_t)ulen, dp.members[m]);
                            at += dclen[m];
                        }
                        if ((int64_t)at != dcomp_bytes) die("Error: the device deflater's member sizes do not add up\n");
                        thj_pinned_free(dcomp);
                        trace(k, "wrapped");
                        on_device_shard(k, std::move(dp));
                        return;
                    }
                    ++host_out_shards;
                    if (parts > 1) {
                        BamWriter::Encoded e;
                        const long long te = WorkClock::now();
                        encode_batch(enc_bw, rt, alns.data(), alns.size(), batch_rd, enc_threads, e);
                        g_work.add(3, te);
                        thj_pinned_free(rinfl);
                        bws[k]->write_encoded(e);
                    } else {
                        // the CPU part of the shard goes to the pool; this feeder moves on to the next shard's device work
                        auto job = std::make_shared<std::vector<Read>>(std::move(batch_rd));
                        pool.submit([&, k, job, rinfl, alns_pinned, n_alns] {
                            BamWriter::Encoded e;
                            const long long te = WorkClock::now();
                            encode_batch(*bws[0], rt, alns_pinned, (size_t)n_alns, *job, 1, e);
                            g_work.add(3, te);
                            trace(k, "encoded");
                            thj_pinned_free(rinfl); thj_pinned_free(alns_pinned);
                            *job = std::vector<Read>();
                            on_encoded(k, std::move(e));
                        });
                    }
                    return;
                }
                if (parts == 1) on_encoded(k, BamWriter::Encoded());
                return;
            }
            if (rc != THJ_EFALLBACK) die("Error: %s\n", thj_last_error());
            static std::atomic<bool> told{false};
            g_host_ingest_shards.fetch_add(1);
            if (!told.exchange(true)) fprintf(stderr, "\tdevice-side ingest not possible (%s); reading on the host\n", thj_last_error());
        }
        std::vector<HitStream> st((size_t)nseg);
        for (int s = 0; s < nseg; ++s)
            if (!st[(size_t)s].open(segs[(size_t)s], rt, o.p, false, sh.seg_off[(size_t)s], sh.begin_id, sh.end_id))
                die("Error opening SAM file %s\n", segs[(size_t)s].c_str());
        // junction-db ("spliced") segment maps: SplicedBAMHitFactory streams, one per segment (:3110-3123)
        std::vector<HitStream> sst(spliced_segs.size());
        for (size_t s = 0; s < spliced_segs.size(); ++s)
            if (!sst[s].open(spliced_segs[s], rt, o.p, true, sh.spliced_off[s], sh.begin_id, sh.end_id)) die("Error opening SAM file %s\n", spliced_segs[s].c_str());
        ReadStream reads;
        if (!reads.open(pos[1], o.zpacker, sh.read_off)) die("Error: cannot open %s for reading\n", pos[1].c_str());
        std::vector<Read> batch_rd;
        std::vector<uint32_t> seg_off; std::vector<thj_span_hit> hits; std::vector<int64_t> read_off; std::string bases, quals;
        size_t max_len = 0;
        BamWriter::Encoded shard_e;
        auto reset = [&]() { seg_off.assign(1, 0); hits.clear(); read_off.assign(1, 0); bases.clear(); quals.clear(); max_len = 0; batch_rd.clear(); };
        auto flush = [&]() {
            int64_t n = (int64_t)read_off.size() - 1;
            if (n == 0) return;
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
            std::vector<thj_aln> alns;
            {
                const long long tw = WorkClock::now();
                std::lock_guard<std::mutex> lk(gpu.mu);
                g_work.add(1, tw);
                const long long td = WorkClock::now();
                thj_ctx* ctx = device_ready(gpu);
                thj_span_batch* dev = nullptr;
                if (thj_span_batch_upload(ctx, &hb, (int64_t)hits.size(), &dev)) die("Error: %s\n", thj_last_error());
                int64_t na = 0;
                for (int attempt = 0;; ++attempt) {           // THJ_ERETRY: a device pool was enlarged, the pass runs again
                    if (thj_span_reset_async(ctx)) die("Error: %s\n", thj_last_error());
                    if (thj_span_run_async(ctx, &o.p, dev)) die("Error: %s\n", thj_last_error());
                    const int frc = thj_span_finish(ctx, &na);
                    if (frc == THJ_ERETRY && attempt < 4) continue;
                    if (frc) die("Error: %s\n", thj_last_error());
                    break;
                }
                alns.resize((size_t)na);
                if (na && thj_span_download(ctx, alns.data())) die("Error: %s\n", thj_last_error());
                if (thj_span_batch_free(ctx, dev)) die("Error: %s\n", thj_last_error());
                g_work.add(2, td);
            }
            BamWriter::Encoded e;
            const long long te = WorkClock::now();
            encode_batch(enc_bw, rt, alns.data(), alns.size(), batch_rd, enc_threads, e);
            g_work.add(3, te);
            if (parts > 1) bws[k]->write_encoded(e);         // this part's own file
            else {                                           // one output file: the shard's batches gather, the planner takes whole shards
                shard_e.bytes.insert(shard_e.bytes.end(), e.bytes.begin(), e.bytes.end());
                shard_e.size.insert(shard_e.size.end(), e.size.begin(), e.size.end());
                shard_e.rid.insert(shard_e.rid.end(), e.rid.begin(), e.rid.end());
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
            batch_rd.push_back(std::move(rd));
            if (read_off.size() - 1 >= batch_reads) flush();
        }
        flush();
        if (parts == 1) on_encoded(k, std::move(shard_e));
    };

    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= S) { if (!getenv("THJ_NO_DRAIN")) thj_pinned_drain(); return; }     // no shard left to start: page-locked buffers go back as they come free, beside the shards still running
            if (parts == 1) { std::unique_lock<std::mutex> lk(win_mu); win_cv.wait(lk, [&] { return k < writer_pos + LOOKAHEAD; }); }
            run_shard(k);
        }
    };
    // thread counts: with the device-side ingest a feeder mostly waits for the GPU (a few per context keep it fed) and the pool has the
    // CPUs; with the host readers the feeders parse, so they are the old workers and the pool gets what is left
    const bool split = dev_ingest && parts == 1;
    int feeders = split ? 2 * n_gpus + 2 : workers;
    if (getenv("THJ_FEEDERS") && atoi(getenv("THJ_FEEDERS")) >= 1) feeders = atoi(getenv("THJ_FEEDERS"));
    int pool_threads = split ? std::max(2, hw - 2) : std::max(2, hw - workers);
    if (getenv("THJ_POOL") && atoi(getenv("THJ_POOL")) >= 1) pool_threads = atoi(getenv("THJ_POOL"));
    if (parts == 1) pool.start(pool_threads);
    const int nthr = (int)std::min<size_t>((size_t)feeders, S);
    std::vector<std::thread> th;
    for (int t = 0; t < nthr; ++t) th.emplace_back(work);
    if (parts == 1) {
        // the writer: shard after shard, batch after batch -- the members arrive deflated, this thread appends them
        for (size_t k = 0; k < S; ++k) {
            OutShard& oq = *outq[k];
            for (;;) {
                BamWriter::Prepared e;
                {
                    std::unique_lock<std::mutex> lk(oq.mu);
                    oq.cv.wait(lk, [&] { return !oq.q.empty() || oq.done; });
                    if (oq.q.empty()) break;
                    e = std::move(oq.q.front());
                    oq.q.pop_front();
                    oq.cv.notify_all();
                }
                bws[0]->commit(e);
                trace(k, "written");
            }
            { std::lock_guard<std::mutex> lk(win_mu); writer_pos = k + 1; }
            win_cv.notify_all();
        }
    }
    for (auto& t : th) t.join();
    pool.finish();
    g_timer.lap("ingest + stitch + encode + write (all shards)");
    fprintf(stderr, "\tshards read on the host because the device-side ingest declined them: %lld\n", g_host_ingest_shards.load());
    if (dev_out) fprintf(stderr, "\tBAM records and BGZF members made on the device for %lld shard%s, on the host for %lld\n", dev_out_shards.load(), dev_out_shards.load() == 1 ? "" : "s", host_out_shards.load());
    for (auto& bw : bws) bw->close();
    g_timer.lap("BAM close");
    g_timer.report();
    thj_ingest_timing_report();
    { static const char* const nm[4] = {"shards (ingest + merge + device + encode)", "  waiting for the GPU's lock", "  device calls (upload, stitch, download)", "  record encoding"}; g_work.report(nm); }
    // Everything is written and closed.  Leave without running the exit handlers or freeing the contexts: tearing the HIP
    // runtime down after a context has been used takes ~0.2 s that nobody is waiting for.
    rt.finish_cache();                 // (the packed-genome cache's writer, when this process was the one to pack the reference)
    finish_outputs_complete(0);
}

int main(int argc, char** argv) { return run_with_handoff(argc, argv, real_main); }
