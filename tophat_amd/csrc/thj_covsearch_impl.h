// thj_covsearch_impl.h -- coverage search of segment_juncs on the device.
// Included at the end of thj_segjuncs.hip (it inserts into the same junction table).
//
// Reference: segment_juncs.cpp  build_coverage_map :4140-4176, capture_island_ends :4268-4543, the POINT_DIR_LEFT /
// POINT_DIR_RIGHT windows of juncs_from_ref_segs<RecordExtendableJuncs> :2052-2377, IntronMotifs::unique/attach_mers
// :700-833, RecordExtendableJuncs::record :1568-1626, extendable_junction :1464-1566, the extension table of the
// initially unmapped reads :146-180, :240-571.
//
// Everything positional is a bitmap with the genome's own block layout: one 64-bit word per 64-base block, word index
// contig_blk[ref] + pos / 64 (a contig owns ceil(len / 64) + 1 words, so position `len` -- the largest a hit's right()
// can reach -- still has a bit).  The reference's scans over vector<bool> become word-parallel shift/and/or kernels:
//   coverage  --(runs of >= min_cov_length - 1 bases, plus the base after them)-->  long_enough
//   long_enough run starts / ends  --(dilation by [-45, +5) / [-5, +45))-->  look-left / look-right flags
//   flags & dinucleotides (N reads as A)  -->  four site bitmaps (GT, CT in look-right; AG, AC in look-left)
// and the pairing of sites within [min, max) coverage intron with the 10-mer extension test runs one thread per site.
// Quirks of the reference that are kept: position 0 of a contig never counts as covered ground for an island start;
// an island whose window would start before position 0 gets no window; a flag run starting at position 0 or reaching
// the last two bases of the contig yields no window at all.

#include "thj_cov_core.h"

namespace cov_k {
using namespace thj::cov;

__global__ void k_add_hits(Layout L, const Hit* hits, const uint32_t* n_hits_ptr, u64* covbits, int32_t* cov_size) {
    const int64_t n = (int64_t)*n_hits_ptr;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        add_hit(L, hits[i], [&](int64_t w, u64 m) { if ((covbits[w] & m) != m) atomicOr((unsigned long long*)&covbits[w], (unsigned long long)m); },
                [&](int k, int32_t sz) { if (cov_size[k] < sz) atomicMax(&cov_size[k], sz); });
}
__global__ void k_long_enough(Layout L, const u64* covbits, u64* le, int m) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w < L.n_words) long_enough_word(L, covbits, le, m, w);
}
__global__ void k_look(Layout L, const u64* le, const int32_t* cov_size, u64* ll, u64* lr) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w < L.n_words) look_word(L, le, cov_size, ll, lr, w);
}
__global__ void k_drop_windows(Layout L, const int32_t* cov_size, u64* ll, u64* lr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * L.n_contigs) drop_windows(L, cov_size, ll, lr, i);
}
__global__ void k_sites(Genome g, Layout L, const u64* ll, const u64* lr, u64* fd, u64* ra, u64* fa, u64* rd) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w < L.n_words) site_word(g, L, ll, lr, fd, ra, fa, rd, w);
}
// one record per unmapped read (thj_cov_core.h: read_record); the table is made from the records when a pass needs it:
// k_ext_count (entries per seed) -> exclusive sum = d_ext_off -> k_ext_scatter (every entry to a place in its seed's range)
__global__ void k_ium_records(const u64* planes, const uint16_t* lens, int64_t n_reads, int W, uint32_t* rec_len, u64* rec_seq, int64_t base) {
    const int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (r < n_reads) read_record(planes, lens, W, rec_len, rec_seq, base, r);
}
__global__ void k_ext_count(const uint32_t* rec_len, const u64* rec_seq, int64_t n, uint32_t* counts) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        record_entries(rec_len[i], rec_seq[i], [&](uint32_t key, u64) { atomicAdd(&counts[key], 1u); });
}
__global__ void k_ext_scatter(const uint32_t* rec_len, const u64* rec_seq, int64_t n, uint32_t* cursor, u64* vals, u64* filter, u64 filter_mask) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        record_entries(rec_len[i], rec_seq[i], [&](uint32_t key, u64 v) {
            vals[atomicAdd(&cursor[key], 1u)] = v;
            // the Bloom filter over the entries (see extendable())
            entry_filter_bits(key, v, filter_mask, [&](u64 b) { if (!((filter[b >> 6] >> (b & 63)) & 1ull)) atomicOr((unsigned long long*)&filter[b >> 6], 1ull << (b & 63)); });
        });
}
// left sites of both orientations compacted into one list: entry = contig position | contig << 32 | antisense << 63
__global__ void k_list_sites(Layout L, const u64* fd, const u64* ra, u64* list, unsigned int* n_list, unsigned int cap) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w >= L.n_words) return;
    const u64 a = fd[w], b = ra[w];
    const int n = __builtin_popcountll(a) + __builtin_popcountll(b);
    if (!n) return;
    const int k = contig_of(L, w);
    const int64_t pos0 = (w - (int64_t)L.contig_blk[k]) * 64;
    unsigned int at = atomicAdd(n_list, (unsigned int)n);
    for (int o = 0; o < 2; ++o) {
        u64 bits = o ? b : a;
        while (bits) {
            const int bit = __builtin_ctzll(bits);
            bits &= bits - 1;
            if (at < cap) list[at] = (u64)(pos0 + bit) | ((u64)k << 32) | ((u64)o << 63);
            ++at;
        }
    }
}
// one wave per listed left site: its 64 lanes share out the words of the acceptor bitmap within reach, so the candidate
// acceptors of a donor are tested side by side.  Junctions go to a list (packed key, skip count): the max_cov_juncs cut
// needs all of them before anything enters the junction set (thj_covsearch_finish).
struct WaveScan {
    __device__ int operator()(int v, int& total) const {
        int x = v;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        total = __shfl(x, 63, 64);
        return x - v;
    }
};
struct ListSink {
    const Genome& g; u64* keys; uint32_t* skips; unsigned long long* count; unsigned long long cap;
    __device__ void cov_junction(uint32_t ref, uint32_t l, uint32_t r, bool a, uint32_t skip) {
        const unsigned long long at = atomicAdd(count, 1ull);
        if (at < cap) { keys[at] = junc_key(g, ref, l, r, a); skips[at] = skip; }
    }
};
__global__ __launch_bounds__(256) void k_pair(Genome g, Layout L, ExtTable et, const u64* list, const unsigned int* n_list, unsigned int cap,
                                              const u64* fa, const u64* rd, int min_intron, int max_intron,
                                              u64* jkeys, uint32_t* jskips, unsigned long long* n_found, unsigned long long jcap) {
    const unsigned int n = *n_list < cap ? *n_list : cap;
    const int lane = threadIdx.x & 63;
    ListSink ev{g, jkeys, jskips, n_found, jcap};
    for (unsigned int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n; i += gridDim.x * 4) {
        const u64 e = list[i];
        const int anti = (int)(e >> 63), k = (int)((e >> 32) & 0x7FFFFFFFull);
        pair_site(g, L, et, anti ? rd : fa, anti, min_intron, max_intron, k, (int64_t)(e & 0xFFFFFFFFull), ev, lane, 64, WaveScan());
    }
}


// ---- butterfly search (thj_cov_core.h: bf_*)
__global__ void k_bf_drop_tail(Layout L, u64* V) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < L.n_contigs) bf_drop_tail(L, V, k);
}
__global__ void k_bf_eligible(Layout L, const u64* V, u64* E) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w < L.n_words) bf_eligible_word(L, V, E, w);
}
__global__ void k_bf_sites(Genome g, Layout L, const u64* E, u64* fd, u64* ra, u64* fa, u64* rd) {
    const int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (w < L.n_words) bf_site_word(g, L, E, fd, ra, fa, rd, w);
}
// one thread per listed site: its keys counted, their places taken with one atomic, then written (nothing is written past `cap`;
// the counter still says how many there are)
__global__ void k_bf_keys(Genome g, Layout L, ExtTable et, const u64* list, unsigned int n, int right_side, u64* out, unsigned long long* count, unsigned long long cap) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const u64 e = list[i];
        unsigned int m = 0;
        bf_site_keys(g, L, et, e, right_side != 0, [&](u64) { ++m; });
        if (!m) continue;
        unsigned long long at = atomicAdd(count, (unsigned long long)m);
        if (at + m > cap) continue;
        bf_site_keys(g, L, et, e, right_side != 0, [&](u64 key) { out[at++] = key; });
    }
}
__global__ void k_bf_join_count(Layout L, const u64* lk, int64_t nl, const u64* rk, int64_t nr, int min_intron, int max_intron, unsigned long long* total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nl; i += (int64_t)gridDim.x * blockDim.x) {
        const BfRange m = bf_match_range(L, rk, nr, lk[i], min_intron, max_intron);
        if (m.hi > m.lo) atomicAdd(total, (unsigned long long)(m.hi - m.lo));
    }
}
struct ReservedSink {
    const Genome& g; u64* keys; uint32_t* skips; unsigned long long at, end;
    __device__ void cov_junction(uint32_t ref, uint32_t l, uint32_t r, bool a, uint32_t skip) {
        if (at < end) { keys[at] = junc_key(g, ref, l, r, a); skips[at] = skip; }
        ++at;
    }
};
__global__ void k_bf_join_emit(Genome g, Layout L, const u64* lk, int64_t nl, const u64* rk, int64_t nr, int min_intron, int max_intron,
                               u64* jkeys, uint32_t* jskips, unsigned long long* count, unsigned long long cap) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nl; i += (int64_t)gridDim.x * blockDim.x) {
        const BfRange m = bf_match_range(L, rk, nr, lk[i], min_intron, max_intron);
        if (m.hi <= m.lo) continue;
        const unsigned long long at = atomicAdd(count, (unsigned long long)(m.hi - m.lo));
        ReservedSink ev{g, jkeys, jskips, at, cap};
        bf_emit_pairs(m, rk, lk[i], ev);
    }
}
// ---- microexon search (thj_cov_core.h): candidates of a batch's reads, table entries of the windows' strings, one wave per window
__global__ __launch_bounds__(256) void k_mx_cands(Genome g, const Hit* hits, const uint32_t* seg_off, const u64* planes, const uint16_t* read_len, int n_reads, int nseg, int W,
                                                  uint32_t ordinal_base, int seg_len, int min_anchor, int side, MxCand* out, unsigned long long* count, unsigned long long cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    mx_read_candidates(g, hits, seg_off + (size_t)r * nseg, nseg, planes + (size_t)r * 3 * W, W, (int)read_len[r], seg_len, min_anchor,
                       [&](int rank, uint32_t ref, int lb, int rb, u64 str, int n) {
                           const unsigned long long at = atomicAdd(count, 1ull);
                           if (out && at < cap) out[at] = MxCand{ordinal_base + (uint32_t)r, (uint16_t)rank, (uint8_t)side, (uint8_t)n, ref, lb, rb, 0u, str};
                       });
}
__global__ __launch_bounds__(256) void k_mx_entries(const u64* strs, const uint8_t* str_len, const uint32_t* str_window, const uint32_t* ent_off, int64_t n_strs, u64* keys, u64* vals) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_strs) return;
    uint32_t at = ent_off[i];
    mx_string_entries(strs[i], (int)str_len[i], (u64)str_window[i], [&](u64 k, u64 v) { keys[at] = k; vals[at] = v; ++at; });
}
__global__ void k_cut_heads(const u64* keys, const uint32_t* skips, int64_t n, uint32_t* flags) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || keys[i] != keys[i - 1] || skips[i] != skips[i - 1]) ? 1u : 0u;
}
__global__ void k_cut_compact(const u64* keys, const uint32_t* flags, const uint32_t* pos, int64_t n, u64* out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n && flags[i]) out[pos[i]] = keys[i];
}
struct MxWin { uint32_t ref_id; int32_t left, right; int32_t side; };
// item = one word of one window's site bitmaps; the window of an item: the last w with win_off[w] <= item
__device__ __forceinline__ int64_t mx_item_window(const uint32_t* win_off, int64_t n_wins, uint32_t item) {
    int64_t lo = 0, hi = n_wins;
    while (hi - lo > 1) { const int64_t mid = (lo + hi) >> 1; if (win_off[mid] <= item) lo = mid; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_mx_sites(Genome g, const MxWin* wins, const uint32_t* win_off, int64_t n_wins, int64_t n_items, int library_type, u64* fd, u64* ra, u64* fa, u64* rd) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const int64_t wi = mx_item_window(win_off, n_wins, (uint32_t)i);
    const MxWin w = wins[wi];
    MxSites s{0, 0, 0, 0};
    if (!(w.left < 0 || w.right >= g_len(g, w.ref_id) - 1)) s = mx_site_word(g, w.ref_id, w.left, w.right, library_type, w.side, (int)(i - win_off[wi]));      // :2154
    fd[i] = s.fd; ra[i] = s.ra; fa[i] = s.fa; rd[i] = s.rd;
}
// one wave per bitmap word: the left sites in it (a handful), each against the right sites of its window within reach
__global__ __launch_bounds__(256) void k_mx_pair(Genome g, MxTable t, const MxWin* wins, const uint32_t* win_off, int64_t n_wins, int64_t n_items, int min_intron,
                                                 const u64* fd, const u64* ra, const u64* fa, const u64* rd,
                                                 u64* jkeys, uint32_t* jskips, unsigned long long* n_found, unsigned long long jcap) {
    const int lane = threadIdx.x & 63;
    ListSink ev{g, jkeys, jskips, n_found, jcap};
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < n_items; i += (int64_t)gridDim.x * 4) {
        const u64 a = fd[i], b = ra[i];
        if (!(a | b)) continue;
        const int64_t wi = mx_item_window(win_off, n_wins, (uint32_t)i);
        const MxWin w = wins[wi];
        const int64_t len = g_len(g, w.ref_id), w0 = w.left >> 6;
        const int n_words = (int)(win_off[wi + 1] - win_off[wi]);
        const int j = (int)(i - win_off[wi]);
        for (int o = 0; o < 2; ++o) {                                      // record(fwd_donors, fwd_acceptors, false), then (rev_acceptors, rev_donors, true)
            u64 bits = o ? b : a;
            while (bits) {
                const int bb = __builtin_ctzll(bits);
                bits &= bits - 1;
                mx_pair_site(g, t, (u64)wi, w.ref_id, len, (o ? rd : fa) + win_off[wi], w0, n_words, o, min_intron, (w0 + j) * 64 + bb, ev, lane, 64, WaveScan());
            }
        }
    }
}

}  // namespace cov_k

// ------------------------------------------------------------------------------------------------ C ABI
static int cov_ensure(thj_ctx* c) {
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    if (!c->d_cov) {
        HIPCHK(hipMalloc(&c->d_cov, (size_t)c->n_blocks * 8 * 8));             // coverage, long_enough, 2 flag and 4 site bitmaps
        HIPCHK(hipMalloc(&c->d_cov_size, (size_t)(c->n_contigs + 1) * 4));
        HIPCHK(hipMalloc(&c->d_ext_off, ((size_t)thj::cov::N_KEYS + 2) * 4 * 2));       // per-seed offsets, and the scatter's cursors behind them
        HIPCHK(hipMalloc(&c->d_cov_found, 16));              // junctions found | left sites listed
    }
    return THJ_OK;
}

static int cov_reserve_ext(thj_ctx* c, int64_t need) {          // room for `need` read records, keeping what is there
    if (need * 23 >= (1ll << 32)) { thj_set_error("more than 2^32 extension-table entries (unmapped reads x 23)"); return THJ_EINVAL; }
    if (need <= c->ext_cap) return THJ_OK;
    const int64_t ncap = need + need / 2 + 1024;
    uint32_t* nk = nullptr; u64* nv = nullptr;
    HIPCHK(hipMalloc(&nk, (size_t)ncap * 4)); HIPCHK(hipMalloc(&nv, (size_t)ncap * 8));
    if (c->n_ext) {
        HIPCHK(hipMemcpyAsync(nk, c->d_ext_key, (size_t)c->n_ext * 4, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(nv, c->d_ext_val, (size_t)c->n_ext * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(c->d_ext_key); hipFree(c->d_ext_val);
    c->d_ext_key = nk; c->d_ext_val = nv; c->ext_cap = ncap;        // (d_ext_key: the records' lengths, d_ext_val: their 2-bit strings; the table itself: cov_build_table)
    return THJ_OK;
}

extern "C" int thj_covsearch_reserve_reads(thj_ctx* c, int64_t n_reads) {
    if (!c || n_reads < 0) { thj_set_error("thj_covsearch_reserve_reads: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    return cov_reserve_ext(c, c->n_ext + n_reads);
}

extern "C" int thj_covsearch_reset_async(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(c->d_cov, 0, (size_t)c->n_blocks * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_cov_size, 0, (size_t)(c->n_contigs + 1) * 4, c->stream));
    HIPCHK(hipMemsetAsync(c->d_cov_found, 0, 16, c->stream));
    c->n_ext = 0;
    return THJ_OK;
}

extern "C" int thj_covsearch_add_hits_async(thj_ctx* c, const thj_seg_batch* db) {
    if (!c || !db) { thj_set_error("thj_covsearch_add_hits_async: null argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if (db->n_reads == 0) return THJ_OK;
    thj::cov::Layout L{c->d_contig_blk, c->d_contig_len, c->n_contigs, c->n_blocks};
    // the batch's hit count is the last CSR offset, which lives on the device
    const uint32_t* n_hits = db->seg_off + (size_t)db->n_reads * db->nseg;
    hipLaunchKernelGGL(cov_k::k_add_hits, dim3(2048), dim3(256), 0, c->stream, L, (const Hit*)db->hits, n_hits, c->d_cov, c->d_cov_size);
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

extern "C" int thj_covsearch_add_reads(thj_ctx* c, int64_t n_reads, int32_t words_per_plane, const uint64_t* planes, const uint16_t* lens,
                                       int32_t on_device) {
    if (!c || (n_reads > 0 && (!planes || !lens))) { thj_set_error("thj_covsearch_add_reads: null argument"); return THJ_EINVAL; }
    if (words_per_plane < 1 || words_per_plane > 4) { thj_set_error("words_per_plane %d unsupported (1..4)", words_per_plane); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if (n_reads == 0) return THJ_OK;
    const int64_t need = c->n_ext + n_reads;
    if ((rc = cov_reserve_ext(c, need))) return rc;
    const u64* d_planes = (const u64*)planes; const uint16_t* d_lens = lens;
    void *tp = nullptr, *tl = nullptr;
    if (!on_device) {                     // host buffers (the executables): staged through a temporary device copy
        const size_t pb = (size_t)n_reads * 3 * words_per_plane * 8, lb = (size_t)n_reads * 2;
        HIPCHK(hipMalloc(&tp, pb)); HIPCHK(hipMalloc(&tl, lb));
        HIPCHK(hipMemcpyAsync(tp, planes, pb, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(tl, lens, lb, hipMemcpyHostToDevice, c->stream));
        d_planes = (const u64*)tp; d_lens = (const uint16_t*)tl;
    }
    hipLaunchKernelGGL(cov_k::k_ium_records, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, c->stream,
                       d_planes, d_lens, n_reads, (int)words_per_plane, c->d_ext_key, c->d_ext_val, c->n_ext);
    HIPCHK(hipGetLastError());
    if (!on_device) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(tp); hipFree(tl); }
    c->n_ext = need;
    return THJ_OK;
}

// the pairing pass over the listed left sites into the (key, skip) list; rerun by thj_covsearch_finish when the list was
// too small (it inserts nothing, so a rerun is harmless)
static int cov_launch_pair(thj_ctx* c) {
    if (!c->d_cov_jkey) {
        c->cov_jcap = 1 << 20;
        HIPCHK(hipMalloc(&c->d_cov_jkey, (size_t)c->cov_jcap * 8)); HIPCHK(hipMalloc(&c->d_cov_jskip, (size_t)c->cov_jcap * 4));
        HIPCHK(hipMalloc(&c->d_cov_jkey2, (size_t)c->cov_jcap * 8)); HIPCHK(hipMalloc(&c->d_cov_jskip2, (size_t)c->cov_jcap * 4));
    }
    const int64_t nw = c->n_blocks;
    thj::cov::Layout L{c->d_contig_blk, c->d_contig_len, c->n_contigs, nw};
    u64 *le = c->d_cov + nw, *fa = c->d_cov + 6 * nw, *rd = c->d_cov + 7 * nw;
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    thj::cov::ExtTable et{c->d_ext_off, c->d_ext_val_sorted, c->d_cov_filter, c->cov_filter_mask};
    const unsigned int list_cap = (unsigned int)(nw < 0xFFFFFFFFll ? nw : 0xFFFFFFFFll);
    HIPCHK(hipMemsetAsync(c->d_cov_found, 0, 8, c->stream));
    hipLaunchKernelGGL(cov_k::k_pair, dim3(2048), dim3(256), 0, c->stream, g, L, et, le, (const unsigned int*)(c->d_cov_found + 1), list_cap, fa, rd,
                       (int)c->cov_min_intron, (int)c->cov_max_intron, c->d_cov_jkey, c->d_cov_jskip, c->d_cov_found, (unsigned long long)c->cov_jcap);
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

// ---- multi-GPU (reads sharded over ranks): the coverage map is the OR of the ranks' maps, the extension table the
// concatenation of their entries.  A rank exposes its state, the caller moves it (RCCL all-gather), and every rank
// folds the others' in before thj_covsearch_run_async; the pairing is then the same on every rank.
__global__ void k_merge_cov(u64* bits, const u64* other_bits, int64_t n_words, int32_t* size, const int32_t* other_size, int32_t n_contigs) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n_words) { const u64 o = other_bits[i]; if (o) bits[i] |= o; }
    if (i < n_contigs && other_size[i] > size[i]) size[i] = other_size[i];
}
extern "C" int thj_covsearch_device_state(thj_ctx* c, const uint64_t** d_cov_bits, int64_t* n_words, const int32_t** d_cov_size,
                                          const uint32_t** d_ext_keys, const uint64_t** d_ext_vals, int64_t* n_ext) {
    if (!c || !d_cov_bits || !n_words || !d_cov_size || !d_ext_keys || !d_ext_vals || !n_ext) { thj_set_error("thj_covsearch_device_state: null argument"); return THJ_EINVAL; }
    int rc = cov_ensure(c);
    if (rc) return rc;
    *d_cov_bits = (const uint64_t*)c->d_cov; *n_words = c->n_blocks; *d_cov_size = c->d_cov_size;
    *d_ext_keys = c->d_ext_key; *d_ext_vals = (const uint64_t*)c->d_ext_val; *n_ext = c->n_ext;
    return THJ_OK;
}
extern "C" int thj_covsearch_merge_async(thj_ctx* c, const uint64_t* d_other_bits, const int32_t* d_other_size,
                                         const uint32_t* d_other_keys, const uint64_t* d_other_vals, int64_t n_other_ext) {
    if (!c || !d_other_bits || !d_other_size || (n_other_ext > 0 && (!d_other_keys || !d_other_vals))) { thj_set_error("thj_covsearch_merge_async: null argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    hipLaunchKernelGGL(k_merge_cov, dim3((unsigned)((c->n_blocks + 255) / 256)), dim3(256), 0, c->stream, c->d_cov, (const u64*)d_other_bits, c->n_blocks,
                       c->d_cov_size, d_other_size, c->n_contigs);
    HIPCHK(hipGetLastError());
    if (n_other_ext > 0) {
        if ((rc = cov_reserve_ext(c, c->n_ext + n_other_ext))) return rc;
        HIPCHK(hipMemcpyAsync(c->d_ext_key + c->n_ext, d_other_keys, (size_t)n_other_ext * 4, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_ext_val + c->n_ext, d_other_vals, (size_t)n_other_ext * 8, hipMemcpyDeviceToDevice, c->stream));
        c->n_ext += n_other_ext;
    }
    return THJ_OK;
}

// the extension table from the read records: entries per seed counted, offsets = their exclusive sum (d_ext_off), every entry scattered into
// its seed's range of d_ext_val_sorted (the order inside a range is whatever the atomics make it: a seed's entries are a set), the Bloom
// filter of extendable() set on the way
static int cov_build_table(thj_ctx* c) {
    using thj::cov::N_KEYS;
    uint32_t* off = c->d_ext_off; uint32_t* cursor = c->d_ext_off + (N_KEYS + 2);
    HIPCHK(hipMemsetAsync(off, 0, ((size_t)N_KEYS + 2) * 4, c->stream));
    if (c->n_ext) hipLaunchKernelGGL(cov_k::k_ext_count, dim3(4096), dim3(256), 0, c->stream, (const uint32_t*)c->d_ext_key, (const u64*)c->d_ext_val, c->n_ext, off);
    size_t need = 0;
    hipcub::DeviceScan::ExclusiveSum(nullptr, need, off, off, (int)(N_KEYS + 1), c->stream);
    if (need > c->sort_tmp_bytes) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; c->sort_tmp_bytes = 0; HIPCHK(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
    size_t bytes = c->sort_tmp_bytes;
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(c->d_sort_tmp, bytes, off, off, (int)(N_KEYS + 1), c->stream));
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, off + N_KEYS, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->ext_sorted_cap < (int64_t)total || !c->d_ext_val_sorted) {
        hipFree(c->d_ext_val_sorted); c->d_ext_val_sorted = nullptr; c->ext_sorted_cap = 0;
        const int64_t cap = (int64_t)total + (int64_t)total / 8 + 1024;
        HIPCHK(hipMalloc(&c->d_ext_val_sorted, (size_t)cap * 8));
        c->ext_sorted_cap = cap;
    }
    // Bloom filter over the entries: 64 bits per entry, a power of two between 2^16 and 2^34 bits
    u64 fbits = 1ull << 16;
    while (fbits < (u64)total * 64 && fbits < (1ull << 34)) fbits <<= 1;
    if ((int64_t)(fbits / 8) > c->cov_filter_bytes) {
        hipFree(c->d_cov_filter); c->d_cov_filter = nullptr;
        HIPCHK(hipMalloc(&c->d_cov_filter, (size_t)(fbits / 8)));
        c->cov_filter_bytes = (int64_t)(fbits / 8);
    }
    HIPCHK(hipMemsetAsync(c->d_cov_filter, 0, (size_t)(fbits / 8), c->stream));
    c->cov_filter_mask = fbits - 1;
    HIPCHK(hipMemcpyAsync(cursor, off, ((size_t)N_KEYS + 1) * 4, hipMemcpyDeviceToDevice, c->stream));
    if (c->n_ext) hipLaunchKernelGGL(cov_k::k_ext_scatter, dim3(4096), dim3(256), 0, c->stream, (const uint32_t*)c->d_ext_key, (const u64*)c->d_ext_val, c->n_ext, cursor,
                                      c->d_ext_val_sorted, c->d_cov_filter, fbits - 1);
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

extern "C" int thj_covsearch_run_async(thj_ctx* c, int32_t min_cov_length, int32_t min_intron, int32_t max_intron) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    if (min_cov_length < 2 || min_cov_length > 64) { thj_set_error("min_cov_length %d unsupported (2..64)", min_cov_length); return THJ_EINVAL; }
    if (min_intron < 1 || max_intron < min_intron) { thj_set_error("coverage intron bounds [%d, %d) unsupported", min_intron, max_intron); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if ((rc = maybe_grow_tables(c))) return rc;
    const int64_t nw = c->n_blocks;
    thj::cov::Layout L{c->d_contig_blk, c->d_contig_len, c->n_contigs, nw};
    u64 *covb = c->d_cov, *le = covb + nw, *ll = le + nw, *lr = ll + nw, *fd = lr + nw, *ra = fd + nw, *fa = ra + nw, *rd = fa + nw;
    if ((rc = cov_build_table(c))) return rc;
    const unsigned gw = (unsigned)((nw + 255) / 256);
    hipLaunchKernelGGL(cov_k::k_long_enough, dim3(gw), dim3(256), 0, c->stream, L, covb, le, (int)min_cov_length - 1);
    hipLaunchKernelGGL(cov_k::k_look, dim3(gw), dim3(256), 0, c->stream, L, le, c->d_cov_size, ll, lr);
    hipLaunchKernelGGL(cov_k::k_drop_windows, dim3((unsigned)((2 * c->n_contigs + 63) / 64)), dim3(64), 0, c->stream, L, c->d_cov_size, ll, lr);
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    hipLaunchKernelGGL(cov_k::k_sites, dim3(gw), dim3(256), 0, c->stream, g, L, ll, lr, fd, ra, fa, rd);
    // left sites -> list (its room: the long_enough bitmap, which nothing reads any more) -> one wave per site
    unsigned int* n_list = (unsigned int*)(c->d_cov_found + 1);
    HIPCHK(hipMemsetAsync(n_list, 0, 4, c->stream));
    const unsigned int list_cap = (unsigned int)(nw < 0xFFFFFFFFll ? nw : 0xFFFFFFFFll);
    hipLaunchKernelGGL(cov_k::k_list_sites, dim3(gw), dim3(256), 0, c->stream, L, fd, ra, le, n_list, list_cap);
    c->cov_min_intron = min_intron; c->cov_max_intron = max_intron; c->cov_pending = true;
    if ((rc = cov_launch_pair(c))) return rc;
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

// the (junction key, skip count) list of a pairing pass -> the pass's junction set; more than max_juncs: the set ordered by skip count
// keeps its smallest elements (segment_juncs.cpp:1611-1621): sort by (skip count, junction), take the first max_juncs
static int cov_cut_and_merge(thj_ctx* c, int64_t n, int64_t max_juncs, int64_t* n_found, bool distinct = false) {
    // distinct: the list may hold a junction twice (the butterfly search finds a pair under more than one key): count and cut over distinct elements
    const u64* keys = c->d_cov_jkey;
    int64_t take = n;
    if (take > max_juncs || (distinct && take > 0)) {
        if (n >= (1ll << 31)) { thj_set_error("more than 2^31 junction candidates"); return THJ_EOVERFLOW; }
        // stable LSD order: by junction key, then by skip count
        size_t b1 = 0, b2 = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, b1, c->d_cov_jkey, c->d_cov_jkey2, c->d_cov_jskip, c->d_cov_jskip2, (int)n, 0, 64, c->stream);
        hipcub::DeviceRadixSort::SortPairs(nullptr, b2, c->d_cov_jskip2, c->d_cov_jskip, c->d_cov_jkey2, c->d_cov_jkey, (int)n, 0, 32, c->stream);
        const size_t need = b1 > b2 ? b1 : b2;
        if (need > c->sort_tmp_bytes) { hipFree(c->d_sort_tmp); HIPCHK(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
        size_t bytes = c->sort_tmp_bytes;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, bytes, c->d_cov_jkey, c->d_cov_jkey2, c->d_cov_jskip, c->d_cov_jskip2, (int)n, 0, 64, c->stream));
        bytes = c->sort_tmp_bytes;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, bytes, c->d_cov_jskip2, c->d_cov_jskip, c->d_cov_jkey2, c->d_cov_jkey, (int)n, 0, 32, c->stream));
        // the set holds DISTINCT (skip count, junction) elements: overlapping microexon windows find the same pair more than once (the
        // coverage search never does).  Heads of runs -> positions -> compacted keys.
        uint32_t* flags = c->d_cov_jskip2;                      // free again after the second sort
        uint32_t* posn = (uint32_t*)c->d_cov_jkey2;             // n * 8 bytes: room for n positions
        hipLaunchKernelGGL(cov_k::k_cut_heads, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const u64*)c->d_cov_jkey, (const uint32_t*)c->d_cov_jskip, n, flags);
        size_t b3 = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, b3, flags, posn, (int)n, c->stream);
        if (b3 > c->sort_tmp_bytes) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; c->sort_tmp_bytes = 0; HIPCHK(hipMalloc(&c->d_sort_tmp, b3)); c->sort_tmp_bytes = b3; }
        bytes = c->sort_tmp_bytes;
        HIPCHK(hipcub::DeviceScan::ExclusiveSum(c->d_sort_tmp, bytes, flags, posn, (int)n, c->stream));
        uint32_t last_pos = 0, last_flag = 0;
        HIPCHK(hipMemcpyAsync(&last_pos, posn + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&last_flag, flags + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        const int64_t n_unique = (int64_t)last_pos + last_flag;
        // compact in place is not safe (a thread may overwrite what another still reads): through the skip buffer's sibling, 8 bytes per key
        u64* packed = nullptr;
        HIPCHK(hipMalloc(&packed, (size_t)(n_unique + 1) * 8));
        hipLaunchKernelGGL(cov_k::k_cut_compact, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const u64*)c->d_cov_jkey, (const uint32_t*)flags, (const uint32_t*)posn, n, packed);
        take = n_unique < max_juncs ? n_unique : max_juncs;
        HIPCHK(hipMemcpyAsync(c->d_cov_jkey, packed, (size_t)take * 8, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(packed);
    }
    if (n_found) *n_found = take;
    if (take > 0) return thj_segjuncs_merge_keys_async(c, 0, (const uint64_t*)keys, take);
    return THJ_OK;
}

extern "C" int thj_covsearch_finish(thj_ctx* c, int64_t max_cov_juncs, int64_t* n_found) {
    // The junctions of the pairing pass enter the pass's junction set here.  When there are more than max_cov_juncs
    // (segment_juncs.cpp:56, :1611-1621) the set ordered by skip count keeps its smallest elements: sort by (skip count,
    // junction) and take the first max_cov_juncs.
    if (!c || max_cov_juncs < 0) { thj_set_error("thj_covsearch_finish: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if (n_found) *n_found = 0;
    if (!c->cov_pending) return THJ_OK;
    unsigned long long n = 0;
    for (;;) {
        HIPCHK(hipMemcpyAsync(&n, c->d_cov_found, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if ((int64_t)n <= c->cov_jcap) break;
        hipFree(c->d_cov_jkey); hipFree(c->d_cov_jskip); hipFree(c->d_cov_jkey2); hipFree(c->d_cov_jskip2);
        c->cov_jcap = (int64_t)n + (int64_t)n / 8 + 1024;
        HIPCHK(hipMalloc(&c->d_cov_jkey, (size_t)c->cov_jcap * 8)); HIPCHK(hipMalloc(&c->d_cov_jskip, (size_t)c->cov_jcap * 4));
        HIPCHK(hipMalloc(&c->d_cov_jkey2, (size_t)c->cov_jcap * 8)); HIPCHK(hipMalloc(&c->d_cov_jskip2, (size_t)c->cov_jcap * 4));
        if ((rc = cov_launch_pair(c))) return rc;
    }
    c->cov_pending = false;
    return cov_cut_and_merge(c, (int64_t)n, max_cov_juncs, n_found);
}


// ------------------------------------------------------------------------------------------------ butterfly search
// device temporaries of a call: freed on every way out (HIPCHK returns from the middle of a function)
struct DevTemps {
    std::vector<void**> slots;
    void own(void** p) { slots.push_back(p); }
    ~DevTemps() { for (void** p : slots) if (*p) { (void)hipFree(*p); *p = nullptr; } }
};
static int bf_sorted_distinct(thj_ctx* c, u64* in, u64* tmp, int64_t n, int64_t* n_out) {          // result in `in`
    if (n == 0) { *n_out = 0; return THJ_OK; }
    if (n >= (1ll << 31)) { thj_set_error("butterfly search: more than 2^31 (site, extension) keys"); return THJ_EOVERFLOW; }
    size_t b1 = 0, b2 = 0;
    int* d_num = nullptr;
    DevTemps temps; temps.own((void**)&d_num);
    hipcub::DeviceRadixSort::SortKeys(nullptr, b1, in, tmp, (int)n, 0, 59, c->stream);
    hipcub::DeviceSelect::Unique(nullptr, b2, tmp, in, d_num, (int)n, c->stream);
    const size_t need = (b1 > b2 ? b1 : b2);
    if (need > c->sort_tmp_bytes) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; c->sort_tmp_bytes = 0; HIPCHK(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
    HIPCHK(hipMalloc(&d_num, sizeof(int)));
    size_t bytes = c->sort_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, bytes, in, tmp, (int)n, 0, 59, c->stream));
    bytes = c->sort_tmp_bytes;
    HIPCHK(hipcub::DeviceSelect::Unique(c->d_sort_tmp, bytes, tmp, in, d_num, (int)n, c->stream));
    int h = 0;
    HIPCHK(hipMemcpyAsync(&h, d_num, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *n_out = h;
    return THJ_OK;
}

extern "C" int thj_butterfly_run(thj_ctx* c, int32_t min_intron, int32_t max_intron, int64_t max_juncs, int64_t* n_found) {
    if (!c || max_juncs < 0) { thj_set_error("thj_butterfly_run: bad argument"); return THJ_EINVAL; }
    if (max_intron < 1 || max_intron > (1 << 29)) { thj_set_error("coverage intron bounds [%d, %d) unsupported", min_intron, max_intron); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if (c->cov_pending) { thj_set_error("thj_butterfly_run: thj_covsearch_finish first (the passes share buffers)"); return THJ_ESTATE; }
    if ((rc = maybe_grow_tables(c))) return rc;
    if (n_found) *n_found = 0;
    const int64_t nw = c->n_blocks;
    thj::cov::Layout L{c->d_contig_blk, c->d_contig_len, c->n_contigs, nw};
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    u64 *covb = c->d_cov, *V = covb + nw, *E = V + nw, *fd = covb + 4 * nw, *ra = fd + nw, *fa = ra + nw, *rd = fa + nw;
    if ((rc = cov_build_table(c))) return rc;
    thj::cov::ExtTable et{c->d_ext_off, c->d_ext_val_sorted, nullptr, 0};
    const unsigned gw = (unsigned)((nw + 255) / 256);
    HIPCHK(hipMemcpyAsync(V, covb, (size_t)nw * 8, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(cov_k::k_bf_drop_tail, dim3((unsigned)((c->n_contigs + 63) / 64)), dim3(64), 0, c->stream, L, V);
    hipLaunchKernelGGL(cov_k::k_bf_eligible, dim3(gw), dim3(256), 0, c->stream, L, (const u64*)V, E);
    hipLaunchKernelGGL(cov_k::k_bf_sites, dim3(gw), dim3(256), 0, c->stream, g, L, (const u64*)E, fd, ra, fa, rd);
    // the sites of each side as a list, their keys, sorted and distinct
    unsigned long long* d_cnt = nullptr;            // [0] sites listed / keys written, [1] pairs
    u64* side_keys[2] = {nullptr, nullptr}; int64_t side_n[2] = {0, 0};
    DevTemps temps; temps.own((void**)&d_cnt); temps.own((void**)&side_keys[0]); temps.own((void**)&side_keys[1]);
    HIPCHK(hipMalloc(&d_cnt, 16));
    auto cleanup = [&]() { };                       // (the temporaries go with `temps`, on every return)
    for (int side = 0; side < 2; ++side) {
        const u64 *b0 = side ? fa : fd, *b1 = side ? rd : ra;
        unsigned int* n_list = (unsigned int*)d_cnt;
        HIPCHK(hipMemsetAsync(d_cnt, 0, 16, c->stream));
        hipLaunchKernelGGL(cov_k::k_list_sites, dim3(gw), dim3(256), 0, c->stream, L, b0, b1, (u64*)nullptr, n_list, 0u);        // counts only
        unsigned int n_sites = 0;
        HIPCHK(hipMemcpyAsync(&n_sites, n_list, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (!n_sites) continue;
        u64 *list = nullptr, *tmp = nullptr;
        DevTemps side_temps; side_temps.own((void**)&list); side_temps.own((void**)&tmp);
        HIPCHK(hipMalloc(&list, (size_t)n_sites * 8));
        HIPCHK(hipMemsetAsync(d_cnt, 0, 16, c->stream));
        hipLaunchKernelGGL(cov_k::k_list_sites, dim3(gw), dim3(256), 0, c->stream, L, b0, b1, list, n_list, n_sites);
        HIPCHK(hipMemsetAsync(d_cnt, 0, 16, c->stream));
        hipLaunchKernelGGL(cov_k::k_bf_keys, dim3(2048), dim3(256), 0, c->stream, g, L, et, (const u64*)list, n_sites, side, (u64*)nullptr, d_cnt, 0ull);      // counts only
        unsigned long long n_keys = 0;
        HIPCHK(hipMemcpyAsync(&n_keys, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (n_keys) {
            if (hipMalloc(&side_keys[side], (size_t)n_keys * 8) != hipSuccess || hipMalloc(&tmp, (size_t)n_keys * 8) != hipSuccess) {
                thj_set_error("butterfly search: no device memory for %llu (site, extension) keys", n_keys); return THJ_ENOMEM;
            }
            HIPCHK(hipMemsetAsync(d_cnt, 0, 16, c->stream));
            hipLaunchKernelGGL(cov_k::k_bf_keys, dim3(2048), dim3(256), 0, c->stream, g, L, et, (const u64*)list, n_sites, side, side_keys[side], d_cnt, n_keys);
            rc = bf_sorted_distinct(c, side_keys[side], tmp, (int64_t)n_keys, &side_n[side]);
            if (rc) return rc;
        }
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    if (!side_n[0] || !side_n[1]) { cleanup(); return THJ_OK; }
    // the join: pairs counted, room made in the (junction key, skip count) list, pairs written
    HIPCHK(hipMemsetAsync(d_cnt, 0, 16, c->stream));
    hipLaunchKernelGGL(cov_k::k_bf_join_count, dim3(2048), dim3(256), 0, c->stream, L, (const u64*)side_keys[0], side_n[0], (const u64*)side_keys[1], side_n[1],
                       (int)min_intron, (int)max_intron, d_cnt + 1);
    unsigned long long n_pairs = 0;
    HIPCHK(hipMemcpyAsync(&n_pairs, d_cnt + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n_pairs >= (1ull << 31)) { cleanup(); thj_set_error("butterfly search: %llu candidate pairs (more than 2^31)", n_pairs); return THJ_EOVERFLOW; }
    if (!n_pairs) { cleanup(); return THJ_OK; }
    if (!c->d_cov_jkey || (int64_t)n_pairs > c->cov_jcap) {
        hipFree(c->d_cov_jkey); hipFree(c->d_cov_jskip); hipFree(c->d_cov_jkey2); hipFree(c->d_cov_jskip2);
        c->d_cov_jkey = c->d_cov_jkey2 = nullptr; c->d_cov_jskip = c->d_cov_jskip2 = nullptr;
        c->cov_jcap = 0;                             // (set when all four lists are there)
        const int64_t jcap = (int64_t)n_pairs + (int64_t)n_pairs / 8 + 1024;
        HIPCHK(hipMalloc(&c->d_cov_jkey, (size_t)jcap * 8)); HIPCHK(hipMalloc(&c->d_cov_jskip, (size_t)jcap * 4));
        HIPCHK(hipMalloc(&c->d_cov_jkey2, (size_t)jcap * 8)); HIPCHK(hipMalloc(&c->d_cov_jskip2, (size_t)jcap * 4));
        c->cov_jcap = jcap;
    }
    HIPCHK(hipMemsetAsync(d_cnt, 0, 16, c->stream));
    hipLaunchKernelGGL(cov_k::k_bf_join_emit, dim3(2048), dim3(256), 0, c->stream, g, L, (const u64*)side_keys[0], side_n[0], (const u64*)side_keys[1], side_n[1],
                       (int)min_intron, (int)max_intron, c->d_cov_jkey, c->d_cov_jskip, d_cnt, (unsigned long long)c->cov_jcap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    cleanup();
    return cov_cut_and_merge(c, (int64_t)n_pairs, max_juncs, n_found, true);
}


// ------------------------------------------------------------------------------------------------ microexon search
static_assert(sizeof(thj_mx_cand) == sizeof(thj::cov::MxCand) && sizeof(thj_mx_window) == sizeof(cov_k::MxWin), "microexon record layouts");

extern "C" int thj_microexon_reset_async(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    c->n_mx_cand = 0;
    return THJ_OK;
}
extern "C" int thj_microexon_collect(thj_ctx* c, const thj_params* p, const thj_seg_batch* db, int32_t read_side) {
    if (!c || !p || !db) { thj_set_error("thj_microexon_collect: null argument"); return THJ_EINVAL; }
    if (p->segment_length < 10 || p->segment_length > 32) { thj_set_error("microexon search: segment_length %d unsupported (10..32: the first segment is kept as one 64-bit string)", p->segment_length); return THJ_EINVAL; }
    if (db->words_per_plane < 1) { thj_set_error("bad batch"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if (db->n_reads == 0 || db->nseg < 2) return THJ_OK;
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    unsigned long long* cnt = c->d_cov_found + 1;
    const unsigned grid = (unsigned)((db->n_reads + 255) / 256);
    // count, make room, fill (an opt-in mode the reference runs on one thread: a round trip per batch is in the noise)
    HIPCHK(hipMemsetAsync(cnt, 0, 8, c->stream));
    hipLaunchKernelGGL(cov_k::k_mx_cands, dim3(grid), dim3(256), 0, c->stream, g, (const Hit*)db->hits, db->seg_off, (const u64*)db->read_planes, db->read_len, db->n_reads, db->nseg,
                       db->words_per_plane, db->ordinal_base, p->segment_length, p->min_anchor_len, read_side, (thj::cov::MxCand*)nullptr, cnt, 0ull);
    unsigned long long n = 0;
    HIPCHK(hipMemcpyAsync(&n, cnt, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (n == 0) return THJ_OK;
    if (c->n_mx_cand + (int64_t)n > c->mx_cand_cap) {
        const int64_t ncap = (c->n_mx_cand + (int64_t)n) * 2 + 1024;
        void* nb = nullptr;
        HIPCHK(hipMalloc(&nb, (size_t)ncap * sizeof(thj_mx_cand)));
        if (c->n_mx_cand) HIPCHK(hipMemcpy(nb, c->d_mx_cand, (size_t)c->n_mx_cand * sizeof(thj_mx_cand), hipMemcpyDeviceToDevice));
        hipFree(c->d_mx_cand); c->d_mx_cand = nb; c->mx_cand_cap = ncap;
    }
    HIPCHK(hipMemsetAsync(cnt, 0, 8, c->stream));
    hipLaunchKernelGGL(cov_k::k_mx_cands, dim3(grid), dim3(256), 0, c->stream, g, (const Hit*)db->hits, db->seg_off, (const u64*)db->read_planes, db->read_len, db->n_reads, db->nseg,
                       db->words_per_plane, db->ordinal_base, p->segment_length, p->min_anchor_len, read_side, (thj::cov::MxCand*)c->d_mx_cand + c->n_mx_cand, cnt, n);
    HIPCHK(hipGetLastError());
    c->n_mx_cand += (int64_t)n;
    return THJ_OK;
}
extern "C" int thj_microexon_candidates(thj_ctx* c, thj_mx_cand** out, int64_t* n_out) {
    if (!c || !out || !n_out) { thj_set_error("thj_microexon_candidates: null argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    *out = nullptr; *n_out = c->n_mx_cand;
    if (!c->n_mx_cand) return THJ_OK;
    thj_mx_cand* h = (thj_mx_cand*)malloc((size_t)c->n_mx_cand * sizeof(thj_mx_cand));
    if (!h) { thj_set_error("out of memory"); return THJ_ENOMEM; }
    HIPCHK(hipMemcpyAsync(h, c->d_mx_cand, (size_t)c->n_mx_cand * sizeof(thj_mx_cand), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = h;
    return THJ_OK;
}
extern "C" int thj_microexon_run(thj_ctx* c, const thj_mx_window* windows, int64_t n_windows, const uint64_t* strs, const uint8_t* str_len, const uint32_t* str_window,
                                 int64_t n_strs, int32_t min_intron, int32_t library_type, int64_t max_juncs, int64_t* n_found) {
    if (!c || n_windows < 0 || n_strs < 0 || (n_windows > 0 && !windows) || (n_strs > 0 && (!strs || !str_len || !str_window)) || max_juncs < 0 || min_intron < 1) {
        thj_set_error("thj_microexon_run: bad argument"); return THJ_EINVAL; }
    if (n_windows >= (1ll << 31) || n_strs >= (1ll << 28)) { thj_set_error("thj_microexon_run: too many windows / strings"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if (n_found) *n_found = 0;
    if (n_windows == 0) return THJ_OK;
    if ((rc = maybe_grow_tables(c))) return rc;
    // entries per string, their offsets (host: a prefix sum over a byte array)
    std::vector<uint32_t> off((size_t)n_strs + 1, 0);
    for (int64_t i = 0; i < n_strs; ++i) {
        if (str_len[i] > 32 || str_window[i] >= (uint32_t)n_windows) { thj_set_error("thj_microexon_run: string %lld out of range", (long long)i); return THJ_EINVAL; }
        off[(size_t)i + 1] = off[(size_t)i] + (str_len[i] >= 10 ? (uint32_t)str_len[i] - 9u : 0u);
    }
    const int64_t n_ent = off[(size_t)n_strs];
    // the windows' site bitmaps: words [woff[w], woff[w + 1]) of four arrays
    std::vector<uint32_t> woff((size_t)n_windows + 1, 0);
    for (int64_t w = 0; w < n_windows; ++w) {
        const int64_t nw = windows[w].right > windows[w].left ? thj::cov::mx_window_words(windows[w].left, windows[w].right) : 0;
        if (nw < 0 || (int64_t)woff[(size_t)w] + nw >= (1ll << 32)) { thj_set_error("thj_microexon_run: windows too wide"); return THJ_EINVAL; }
        woff[(size_t)w + 1] = woff[(size_t)w] + (uint32_t)nw;
    }
    const int64_t n_items = woff[(size_t)n_windows];
    void *d_win = nullptr, *d_str = nullptr, *d_len = nullptr, *d_sw = nullptr, *d_off = nullptr, *d_k = nullptr, *d_v = nullptr, *d_k2 = nullptr, *d_v2 = nullptr, *d_woff = nullptr, *d_bm = nullptr;
    auto cleanup = [&]() { hipFree(d_win); hipFree(d_str); hipFree(d_len); hipFree(d_sw); hipFree(d_off); hipFree(d_k); hipFree(d_v); hipFree(d_k2); hipFree(d_v2); hipFree(d_woff); hipFree(d_bm); };
#define MX_HIP(e) do { hipError_t e__ = (e); if (e__ != hipSuccess) { thj_set_error("%s: %s", #e, hipGetErrorString(e__)); cleanup(); return THJ_EHIP; } } while (0)
    MX_HIP(hipMalloc(&d_win, (size_t)n_windows * sizeof(thj_mx_window)));
    MX_HIP(hipMalloc(&d_str, (size_t)(n_strs + 1) * 8)); MX_HIP(hipMalloc(&d_len, (size_t)n_strs + 1)); MX_HIP(hipMalloc(&d_sw, (size_t)(n_strs + 1) * 4)); MX_HIP(hipMalloc(&d_off, (size_t)(n_strs + 1) * 4));
    MX_HIP(hipMalloc(&d_k, (size_t)(n_ent + 1) * 8)); MX_HIP(hipMalloc(&d_v, (size_t)(n_ent + 1) * 8)); MX_HIP(hipMalloc(&d_k2, (size_t)(n_ent + 1) * 8)); MX_HIP(hipMalloc(&d_v2, (size_t)(n_ent + 1) * 8));
    MX_HIP(hipMalloc(&d_woff, (size_t)(n_windows + 1) * 4)); MX_HIP(hipMalloc(&d_bm, (size_t)(n_items + 1) * 8 * 4));
    MX_HIP(hipMemcpyAsync(d_woff, woff.data(), (size_t)(n_windows + 1) * 4, hipMemcpyHostToDevice, c->stream));
    MX_HIP(hipMemcpyAsync(d_win, windows, (size_t)n_windows * sizeof(thj_mx_window), hipMemcpyHostToDevice, c->stream));
    if (n_strs) {
        MX_HIP(hipMemcpyAsync(d_str, strs, (size_t)n_strs * 8, hipMemcpyHostToDevice, c->stream));
        MX_HIP(hipMemcpyAsync(d_len, str_len, (size_t)n_strs, hipMemcpyHostToDevice, c->stream));
        MX_HIP(hipMemcpyAsync(d_sw, str_window, (size_t)n_strs * 4, hipMemcpyHostToDevice, c->stream));
        MX_HIP(hipMemcpyAsync(d_off, off.data(), (size_t)(n_strs + 1) * 4, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(cov_k::k_mx_entries, dim3((unsigned)((n_strs + 255) / 256)), dim3(256), 0, c->stream, (const u64*)d_str, (const uint8_t*)d_len, (const uint32_t*)d_sw, (const uint32_t*)d_off, n_strs,
                           (u64*)d_k, (u64*)d_v);
    }
    const u64* keys = (const u64*)d_k; const u64* vals = (const u64*)d_v;
    if (n_ent > 1) {
        int bits = 21; while (bits < 52 && (1ll << (bits - 20)) < n_windows) ++bits;
        size_t need = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, need, (const u64*)d_k, (u64*)d_k2, (const u64*)d_v, (u64*)d_v2, (int)n_ent, 0, bits, c->stream);
        if (need > c->sort_tmp_bytes) { MX_HIP(hipStreamSynchronize(c->stream)); hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; c->sort_tmp_bytes = 0; MX_HIP(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
        size_t bytes = c->sort_tmp_bytes;
        MX_HIP(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, bytes, (const u64*)d_k, (u64*)d_k2, (const u64*)d_v, (u64*)d_v2, (int)n_ent, 0, bits, c->stream));
        keys = (const u64*)d_k2; vals = (const u64*)d_v2;
    }
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    thj::cov::MxTable t{keys, vals, n_ent};
    u64 *fd = (u64*)d_bm, *ra = fd + n_items, *fa = ra + n_items, *rd = fa + n_items;
    if (n_items) hipLaunchKernelGGL(cov_k::k_mx_sites, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, c->stream, g, (const cov_k::MxWin*)d_win, (const uint32_t*)d_woff, n_windows, n_items,
                                    (int)library_type, fd, ra, fa, rd);
    unsigned long long n = 0;
    for (;;) {
        MX_HIP(hipMemsetAsync(c->d_cov_found, 0, 8, c->stream));
        const int64_t blocks_wanted = (n_items + 3) / 4;
        const unsigned grid = (unsigned)(blocks_wanted < 1 ? 1 : blocks_wanted < 65536 ? blocks_wanted : 65536);
        hipLaunchKernelGGL(cov_k::k_mx_pair, dim3(grid), dim3(256), 0, c->stream, g, t, (const cov_k::MxWin*)d_win, (const uint32_t*)d_woff, n_windows, n_items, (int)min_intron,
                           (const u64*)fd, (const u64*)ra, (const u64*)fa, (const u64*)rd, c->d_cov_jkey, c->d_cov_jskip, c->d_cov_found, (unsigned long long)c->cov_jcap);
        MX_HIP(hipMemcpyAsync(&n, c->d_cov_found, 8, hipMemcpyDeviceToHost, c->stream));
        MX_HIP(hipStreamSynchronize(c->stream));
        if ((int64_t)n <= c->cov_jcap) break;
        hipFree(c->d_cov_jkey); hipFree(c->d_cov_jskip); hipFree(c->d_cov_jkey2); hipFree(c->d_cov_jskip2);
        c->d_cov_jkey = c->d_cov_jkey2 = nullptr; c->d_cov_jskip = c->d_cov_jskip2 = nullptr;
        c->cov_jcap = (int64_t)n + (int64_t)n / 8 + 1024;
        MX_HIP(hipMalloc(&c->d_cov_jkey, (size_t)c->cov_jcap * 8)); MX_HIP(hipMalloc(&c->d_cov_jskip, (size_t)c->cov_jcap * 4));
        MX_HIP(hipMalloc(&c->d_cov_jkey2, (size_t)c->cov_jcap * 8)); MX_HIP(hipMalloc(&c->d_cov_jskip2, (size_t)c->cov_jcap * 4));
    }
    rc = cov_cut_and_merge(c, (int64_t)n, max_juncs, n_found);
    hipStreamSynchronize(c->stream);
    cleanup();
#undef MX_HIP
    return rc;
}

static void cov_free(thj_ctx* c) {
    hipFree(c->d_cov); hipFree(c->d_cov_size); hipFree(c->d_ext_off); hipFree(c->d_cov_found); hipFree(c->d_cov_filter);
    hipFree(c->d_cov_jkey); hipFree(c->d_cov_jskip); hipFree(c->d_cov_jkey2); hipFree(c->d_cov_jskip2);
    hipFree(c->d_ext_key); hipFree(c->d_ext_val); hipFree(c->d_ext_key_sorted); hipFree(c->d_ext_val_sorted);
    hipFree(c->d_mx_cand);
}
