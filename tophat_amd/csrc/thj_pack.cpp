// thj_pack.cpp -- host-side packers of the C ABI (include/thj.h): genome ->
// 64-base bit-plane blocks, reads -> bit-planes, parameter defaults, error text.
// Replaces the data-preparation half of get_seqs (segment_juncs.cpp:64-88) and
// ReadStream (reads.cpp:528-630) -- the parsing half lives in the drop-in
// binaries.  Pure host C++; no device code.
#include "../../include/thj.h"
#include "thj_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";

void thj_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" const char* thj_last_error(void) { return g_err; }
extern "C" const char* thj_version(void) { return "thj-hip 0.2 (gfx950)"; }
extern "C" int thj_abi_version(void) { return THJ_ABI_VERSION; }

extern "C" void thj_params_default(thj_params* p) {
    // common.cpp:79-180
    memset(p, 0, sizeof *p);
    p->segment_length = 25; p->segment_mismatches = 2;
    p->min_segment_intron = 50; p->max_segment_intron = 500000;
    p->max_insertion_length = 3; p->max_deletion_length = 3;
    p->max_seg_multihits = 40;
    p->inner_dist_mean = 200; p->inner_dist_std_dev = 20;
    p->library_type = 0; p->bowtie2 = 1; p->read_side = 1;
    p->min_report_intron = 50; p->max_report_intron = 500000; p->min_anchor_len = 8;
    p->read_mismatches = 2; p->read_gap_length = 2; p->read_edit_dist = 2;
    p->bowtie2_max_penalty = 6; p->bowtie2_min_penalty = 2; p->bowtie2_penalty_for_N = 1;
    p->bowtie2_read_gap_open = 5; p->bowtie2_read_gap_cont = 3;
    p->bowtie2_ref_gap_open = 5; p->bowtie2_ref_gap_cont = 3;
    p->fusion_anchor_length = 20; p->fusion_min_dist = 10000000; p->fusion_search = 0;
}

extern "C" int thj_genome_layout(int32_t n_contigs, const int64_t* lens, uint32_t* contig_blk, int64_t* n_blocks) {
    if (n_contigs < 0 || !lens || !contig_blk || !n_blocks) { thj_set_error("thj_genome_layout: null argument"); return THJ_EINVAL; }
    int64_t b = 0;
    for (int32_t i = 0; i < n_contigs; ++i) {
        if (lens[i] < 0 || lens[i] > 0x7fffffff) { thj_set_error("contig %d length %lld unsupported", i, (long long)lens[i]); return THJ_EINVAL; }
        contig_blk[i] = (uint32_t)b;
        b += (lens[i] + 63) / 64 + 1;       // one zero guard block after every contig
        // the packed event keys keep the global base coordinate (+1) in 34 bits (junc_key / ins_key, thj_core.h)
        if (b * 64 + 1 >= (1ll << 34)) { thj_set_error("genome too large: %lld bases with guard blocks, the packed event keys hold 2^34", (long long)(b * 64)); return THJ_EINVAL; }
    }
    contig_blk[n_contigs] = (uint32_t)b;
    *n_blocks = b + 1;                      // and one at the very end for the funnel's second load
    return THJ_OK;
}

static inline int base_code(char c) {
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return 4;
    }
}

// Host-side packing is embarrassingly parallel; THJ_HOST_THREADS bounds the workers (default min(64, hardware threads)).
static int pack_threads() {
    int n = getenv("THJ_HOST_THREADS") ? atoi(getenv("THJ_HOST_THREADS")) : 0;
    if (n <= 0) { n = (int)std::thread::hardware_concurrency(); if (n > 64) n = 64; }
    return n < 1 ? 1 : n;
}
template <class F>
static void parallel_ranges(int64_t n, int64_t grain, F f) {       // f(begin, end) over a partition of [0, n)
    int T = pack_threads();
    if ((int64_t)T > n / grain + 1) T = (int)(n / grain + 1);
    if (T <= 1) { f((int64_t)0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] { f(n * t / T, n * (t + 1) / T); });
    for (auto& x : th) x.join();
}

extern "C" int thj_genome_pack(int32_t n_contigs, const char* const* seqs, const int64_t* lens,
                               const uint32_t* contig_blk, uint64_t* blocks, int64_t n_blocks) {
    if (!blocks || !contig_blk || !lens) { thj_set_error("thj_genome_pack: null argument"); return THJ_EINVAL; }
    // every block is written exactly once: a contig's blocks by the loop below, the guard block behind it and the one at the very end
    // here (a memset of the whole array first was a third of the time a 3 Gb genome takes)
    for (int32_t c = 0; c < n_contigs; ++c) {
        if (lens[c] != 0 && (!seqs || !seqs[c])) { thj_set_error("contig %d has a length but no sequence", c); return THJ_EINVAL; }
        for (int64_t b = (int64_t)contig_blk[c] + (lens[c] + 63) / 64; b < (int64_t)contig_blk[c + 1]; ++b) memset(blocks + b * 4, 0, 32);
    }
    for (int64_t b = n_contigs > 0 ? (int64_t)contig_blk[n_contigs] : 0; b < n_blocks; ++b) memset(blocks + b * 4, 0, 32);
    static const struct Codes { uint8_t t[256]; Codes() { for (int c = 0; c < 256; ++c) t[c] = (uint8_t)base_code((char)c); } } codes;
    // one partition of the whole genome's blocks over the workers (a contig at a time left the threads of a 25-contig genome starting
    // and stopping 25 times, the last contigs too small to share)
    std::vector<int64_t> first((size_t)n_contigs + 1, 0);           // blocks of sequence before contig c
    for (int32_t c = 0; c < n_contigs; ++c) first[(size_t)c + 1] = first[(size_t)c] + (lens[c] + 63) / 64;
    const int64_t total = first[(size_t)n_contigs];
    parallel_ranges(total, 1 << 14, [=, &first](int64_t k0, int64_t k1) {
        int32_t c = 0;
        while (c + 1 < n_contigs && first[(size_t)c + 1] <= k0) ++c;
        for (int64_t k = k0; k < k1; ++k) {
            while (first[(size_t)c + 1] <= k) ++c;
            const int64_t b0 = (k - first[(size_t)c]) * 64, n = lens[c];
            const unsigned char* s = (const unsigned char*)seqs[c] + b0;
            const int lim = n - b0 < 64 ? (int)(n - b0) : 64;
            uint64_t lo = 0, hi = 0, nm = 0;
            for (int j = 0; j < lim; ++j) {
                const uint64_t code = codes.t[s[j]];          // 0..3, 4 = N (lo = hi = 0 there)
                lo |= (code & 1ull) << j; hi |= ((code >> 1) & 1ull) << j; nm |= (code >> 2) << j;
            }
            uint64_t* blk = blocks + ((int64_t)contig_blk[c] + (k - first[(size_t)c])) * 4;
            blk[0] = lo; blk[1] = hi; blk[2] = nm; blk[3] = 0;
        }
    });
    return THJ_OK;
}

extern "C" int thj_reads_pack(int64_t n_reads, const int64_t* read_off, const char* bases,
                              int32_t W, uint64_t* planes, uint16_t* lens) {
    if (!read_off || !bases || !planes || !lens || W < 1) { thj_set_error("thj_reads_pack: bad argument"); return THJ_EINVAL; }
    for (int64_t r = 0; r < n_reads; ++r) {
        int64_t n = read_off[r + 1] - read_off[r];
        if (n < 0 || n > (int64_t)W * 64) { thj_set_error("read %lld length %lld exceeds %d bases", (long long)r, (long long)n, W * 64); return THJ_EINVAL; }
    }
    parallel_ranges(n_reads, 1 << 14, [=](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            int64_t n = read_off[r + 1] - read_off[r];
            lens[r] = (uint16_t)n;
            uint64_t* rp = planes + r * 3 * W;
            memset(rp, 0, (size_t)(3 * W) * 8);
            const char* s = bases + read_off[r];
            for (int64_t k = 0; k < n; ++k) {
                int code = base_code(s[k]);
                // reads keep their case in the reference; prep_reads emits upper-case ACGTN
                int w = (int)(k >> 6), b = (int)(k & 63);
                if (code == 4) rp[2 * W + w] |= 1ull << b;
                else { rp[w] |= (uint64_t)(code & 1) << b; rp[W + w] |= (uint64_t)(code >> 1) << b; }
            }
        }
    });
    return THJ_OK;
}

// MD:Z of one alignment, the way bowtie_sam_extra builds it (bwt_map.cpp:2467-2648) -- the host's part for the few records
// whose MD string does not fit the 40 characters a device record holds (thj_aln.md_len == THJ_MD_ON_HOST).
extern "C" int thj_md_string2(const char* ref, int64_t ref_len, const char* ref2, int64_t ref2_len, const char* seq, int32_t seq_len, int32_t left,
                              const uint32_t* cigar, int32_t n_cigar, char* out, int32_t out_cap) {
    if (!ref || !ref2 || !seq || !cigar || !out || out_cap < 2 || n_cigar < 0) { thj_set_error("thj_md_string: bad argument"); return THJ_EINVAL; }
    auto fold = [](char c) { switch (c) { case 'A': case 'a': return 'A'; case 'C': case 'c': return 'C'; case 'G': case 'g': return 'G'; case 'T': case 't': return 'T'; default: return 'N'; } };
    auto comp = [](char c) { switch (c) { case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; default: return 'N'; } };
    int n = 0;
    auto put = [&](char c) { if (n + 1 < out_cap) out[n] = c; ++n; };
    auto put_int = [&](int v) { char b[16]; int k = snprintf(b, sizeof b, "%d", v); for (int i = 0; i < k; ++i) put(b[i]); };
    const bool plain = ref2 == ref;
    // past the contig: the device genome's zero guard block for the plain tiers, N (seqan's infix) for the fusion tier
    auto at = [&](const char* r, int64_t rl, int64_t pos) { return pos >= 0 && pos < rl ? fold(r[pos]) : (plain ? 'A' : 'N'); };
    int64_t pos_ref = left;
    int pos_seq = 0, pos_mm = 0;
    bool saw = false;
    for (int i = 0; i < n_cigar; ++i) {
        const uint32_t op = cigar[i] >> 28; const int len = (int)(cigar[i] & 0x0FFFFFFFu);
        if (op == THJ_CIG_MATCH || op == THJ_CIG_mATCH) {
            for (int k = 0; k < len && pos_seq + k < seq_len; ++k) {
                const char r = op == THJ_CIG_MATCH ? at(ref, ref_len, pos_ref + k) : comp(at(ref, ref_len, pos_ref - k));
                const char s = fold(seq[pos_seq + k]);
                if (r != s) { put_int(pos_mm); put(r); pos_mm = 0; } else ++pos_mm;
            }
            pos_seq += len; pos_ref += op == THJ_CIG_MATCH ? len : -len;
        } else if (op == THJ_CIG_INS || op == THJ_CIG_iNS) pos_seq += len;
        else if (op == THJ_CIG_DEL || op == THJ_CIG_dEL) {
            put_int(pos_mm); put('^');
            for (int k = 0; k < len && k < 64; ++k) put(op == THJ_CIG_DEL ? at(ref, ref_len, pos_ref + k) : comp(at(ref, ref_len, pos_ref - k)));
            pos_ref += op == THJ_CIG_DEL ? len : -len; pos_mm = 0;
        } else if (op == THJ_CIG_REF_SKIP) pos_ref += len;
        else if (op == THJ_CIG_rEF_SKIP) pos_ref -= len;
        else if (op >= THJ_CIG_FUSION_FF && op <= THJ_CIG_FUSION_RR) {
            if (saw) { n = 0; break; }
            ref = ref2; ref_len = ref2_len; pos_ref = len; saw = true;
        }
    }
    put_int(pos_mm);
    if (n + 1 > out_cap) { thj_set_error("thj_md_string: %d characters do not fit the buffer", n); return THJ_EINVAL; }
    out[n] = 0;
    return n;
}
extern "C" int thj_md_string(const char* ref, int64_t ref_len, const char* seq, int32_t seq_len, int32_t left, const uint32_t* cigar, int32_t n_cigar,
                             char* out, int32_t out_cap) {
    return thj_md_string2(ref, ref_len, ref, ref_len, seq, seq_len, left, cigar, n_cigar, out, out_cap);
}
