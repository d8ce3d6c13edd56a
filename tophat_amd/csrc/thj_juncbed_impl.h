// thj_juncbed_impl.h -- junction consensus of tophat_reports on the device: the alignments
// long_spanning_reads left resident in HBM (or any records handed in) reduced to the JunctionSet that junctions.bed prints.
//
//   junctions_from_spliced_hit / junctions_from_alignment / JunctionStats::merge_with   junctions.cpp:19-142, junctions.h:87-101
//   accept_if_valid, knockout_shadow_junctions (filter_junctions)                        junctions.cpp:192-330
//   exclude_hits_on_filtered_junctions + update_junctions of the second pass             tophat_reports.cpp:1182-1230
//   the final extent filter                                                              tophat_reports.cpp:2974-2984
//
// A segmented reduce in hash-table form.  Pass 1: every REF_SKIP of every record is one "occurrence"; its junction key goes
// into an open-addressing table (support += 1, extents = max) and the occurrence is kept -- 16 bytes: table slot, extents,
// position inside its record -- because the second pass needs it again after the filter.  Filter: accept_if_valid per
// distinct junction; the distinct keys are radix-sorted and every accepted junction looks at its opposite-strand neighbours
// within min_anchor_len.  Pass 2: a record all of whose junctions survived adds its occurrences to the final statistics.
// All integer work; the order in which records arrive does not matter (sums and maxima).
//
// Included at the end of thj_span.hip.  Not part of the timed hot path of bench.py unless asked for.
#pragma once

struct JbOcc { uint32_t slot; uint16_t le, re; uint8_t nj, idx; uint16_t pad; uint32_t pad2; };      // 16 bytes
static_assert(sizeof(JbOcc) == 16, "occurrence layout");

struct JbTable {
    u64* key; u64 mask;
    uint32_t *cnt1, *le1, *re1, *cnt2, *le2, *re2, *left, *acc;
    u64* list;                         // distinct keys in arrival order
    unsigned long long* counters;      // [0] distinct, [1] occurrences counted, [2] occurrences written, [3] overflow flag
};

__device__ __forceinline__ u64 jb_mix(u64 x) { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return x; }

__device__ __forceinline__ uint32_t jb_insert(const JbTable& t, u64 k, uint32_t left) {
    u64 h = jb_mix(k) & t.mask;
    for (u64 probe = 0; probe <= t.mask; ++probe) {
        u64 cur = __hip_atomic_load(&t.key[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == ~0ull) {
            const u64 old = atomicCAS((unsigned long long*)&t.key[h], ~0ull, k);
            if (old == ~0ull) { const unsigned long long pos = atomicAdd(&t.counters[0], 1ull); if (pos <= t.mask) t.list[pos] = k; t.left[h] = left; cur = k; }
            else cur = old;
        }
        if (cur == k) return (uint32_t)h;
        h = (h + 1) & t.mask;
    }
    atomicExch(&t.counters[3], 1ull);
    return 0xFFFFFFFFu;
}

// the junctions of one record (junctions_from_spliced_hit, junctions.cpp:19-92): calls f(ref_id, left, right, left_extent, right_extent)
// per REF_SKIP / rEF_SKIP.  Pieces that run down the genome (lower-case ops 2, 6, 12) walk backwards and swap the extents; a fusion op
// (FF 7, FR 8, RF 9) jumps to its length = the position on the second contig (cigar[15] of the record) and the junctions behind it belong
// to that contig; FUSION_RR (10) has no case in the reference and none here.
// slot: the record is in the stitch kernels' slot layout (RecSink, thj_span.hip: cigar ops 4.. live in the tail line)
template <class F>
__device__ __forceinline__ int jb_rec_juncs(const OutAln& a, bool slot, F f) {
    int n = 0;
    int64_t j = a.left;
    const uint32_t* w = (const uint32_t*)&a;
    auto cg = [&](int c) { return w[slot && c >= 4 ? 12 + c : 6 + c]; };
    uint32_t ref = a.ref_id;
    for (int c = 0; c < a.n_cigar && c < SPAN_MAXC; ++c) {
        const uint32_t op = cg(c) >> 28, len = cg(c) & 0x0FFFFFFFu;
        if (op == 11 || op == 12) {
            const uint32_t prev = c > 0 ? (cg(c - 1) & 0x0FFFFFFFu) : 0u, next = c + 1 < a.n_cigar ? (cg(c + 1) & 0x0FFFFFFFu) : 0u;
            if (op == 11) { f(ref, (uint32_t)(j - 1), (uint32_t)(j + len), prev, next); j += len; }
            else { f(ref, (uint32_t)(j - len), (uint32_t)(j + 1), next, prev); j -= len; }
            ++n;
        } else if (op == 1 || op == 5) j += len;
        else if (op == 2 || op == 6) j -= len;
        else if (op == 7 || op == 8 || op == 9) { j = len; ref = cg(SPAN_MAXC - 1); }
    }
    return n;
}

// record i of a pass: slots (first record of every read that has one) then the extra pool; or a plain array
struct JbRecs { const OutAln* slots; const uint8_t* nrec; int64_t n_slots; const OutAln* extra; int64_t n_extra; bool slot_layout; };
__device__ __forceinline__ const OutAln* jb_rec(const JbRecs& r, int64_t i) {
    if (i < r.n_slots) return (!r.nrec || r.nrec[i]) ? &r.slots[i] : nullptr;
    return &r.extra[i - r.n_slots];
}

__global__ __launch_bounds__(256) void thj_k_jb_count(JbRecs r, unsigned long long* counters) {
    __shared__ unsigned int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    unsigned int mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < r.n_slots + r.n_extra; i += (int64_t)gridDim.x * blockDim.x) {
        const OutAln* a = jb_rec(r, i);
        if (a) mine += (unsigned)jb_rec_juncs(*a, r.slot_layout, [](uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) {});
    }
    if (mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&counters[1], (unsigned long long)s_n);
}

__global__ __launch_bounds__(256) void thj_k_jb_add(Genome g, JbRecs r, JbTable t, JbOcc* occ, unsigned long long occ_cap) {
    const int lane = threadIdx.x & 63;
    const int64_t total = r.n_slots + r.n_extra;
    // whole waves walk together so that the wave-wide reservation below sees every lane
    const int64_t n_iter = (total + (int64_t)gridDim.x * blockDim.x - 1) / ((int64_t)gridDim.x * blockDim.x);
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t i = it * (int64_t)gridDim.x * blockDim.x + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        const OutAln* a = i < total ? jb_rec(r, i) : nullptr;
        unsigned int nj = a ? (unsigned)jb_rec_juncs(*a, r.slot_layout, [](uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) {}) : 0u;
        // one reservation per wave: inclusive scan of nj over the lanes, the last lane adds the total
        unsigned int incl = nj;
        for (int d = 1; d < 64; d <<= 1) { const unsigned int up = __shfl_up(incl, d); if (lane >= d) incl += up; }
        const unsigned int wave_total = __shfl(incl, 63);
        unsigned long long base = 0;
        if (lane == 63 && wave_total) base = atomicAdd(&t.counters[2], (unsigned long long)wave_total);
        base = __shfl(base, 63);
        if (!nj) continue;
        unsigned long long at = base + incl - nj;
        const bool anti = (a->flags & 4u) != 0;             // THJ_HIT_ANTISENSE_SPLICE
        uint8_t idx = 0;
        const uint8_t n8 = (uint8_t)nj;
        jb_rec_juncs(*a, r.slot_layout, [&](uint32_t ref, uint32_t left, uint32_t right, uint32_t le, uint32_t re) {
            const uint32_t slot = jb_insert(t, junc_key(g, ref, left, right, anti), left);
            if (slot != 0xFFFFFFFFu) {
                atomicAdd(&t.cnt1[slot], 1u);
                atomicMax(&t.le1[slot], le);
                atomicMax(&t.re1[slot], re);
            }
            if (at < occ_cap) occ[at] = JbOcc{slot, (uint16_t)(le > 65535u ? 65535u : le), (uint16_t)(re > 65535u ? 65535u : re), n8, idx, 0, 0};
            ++at; ++idx;
        });
    }
}

// accept_if_valid (junctions.cpp:192-242) per distinct junction
__global__ __launch_bounds__(256) void thj_k_jb_accept(JbTable t, int64_t n, int min_anchor) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= (int64_t)t.mask; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 k = t.key[i];
        if (k == ~0ull) continue;
        const uint32_t le = t.le1[i], re = t.re1[i], mn = le < re ? le : re, len = (uint32_t)((k >> 1) & ((1ull << 29) - 1));
        uint32_t ok;
        if ((int)mn < min_anchor) ok = 0;
        else if (len > 50000u) ok = t.cnt1[i] >= 2u && mn > 12u;
        else ok = 1;
        t.acc[i] = ok;
    }
    (void)n;
}

__device__ __forceinline__ uint32_t jb_find(const JbTable& t, u64 k) {
    u64 h = jb_mix(k) & t.mask;
    for (u64 probe = 0; probe <= t.mask; ++probe) {
        const u64 cur = t.key[h];
        if (cur == k) return (uint32_t)h;
        if (cur == ~0ull) return 0xFFFFFFFFu;
        h = (h + 1) & t.mask;
    }
    return 0xFFFFFFFFu;
}

// knockout_shadow_junctions (junctions.cpp:244-315) over the sorted distinct keys: an accepted junction loses to a junction of
// the other strand that starts within min_anchor_len before it (or at it, ending within min_anchor_len after it) when that
// one has more support.  Writes acc2 (the junction's own flag only, as the reference does).
__global__ __launch_bounds__(256) void thj_k_jb_knockout(JbTable t, const u64* sorted, int64_t n, int min_anchor, uint32_t* acc2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 k = sorted[i];
        const uint32_t si = jb_find(t, k);
        uint32_t ok = t.acc[si];
        if (ok && t.left[si] >= (uint32_t)min_anchor) {        // left < anchor: the reference's unsigned left wraps, the range is empty
            const u64 anti = k & 1ull, len = (k >> 1) & ((1ull << 29) - 1), gp = k >> 30;
            // fuzzy_left = (left - anchor, right, !strand), fuzzy_right = (left, right + anchor, !strand): in (left, right - left)
            // key space both have the length field len + anchor
            const u64 len2 = (len + (u64)min_anchor) & ((1ull << 29) - 1);
            const u64 lo = ((gp - (u64)min_anchor) << 30) | (len2 << 1) | (anti ^ 1ull);
            const u64 hi = (gp << 30) | (len2 << 1) | (anti ^ 1ull);
            int64_t a = 0, b = n;                              // lower_bound(lo)
            while (a < b) { const int64_t m = (a + b) >> 1; if (sorted[m] < lo) a = m + 1; else b = m; }
            const uint32_t my_support = t.cnt1[si];
            for (int64_t q = a; q < n && sorted[q] <= hi; ++q) {
                const u64 k2 = sorted[q];
                if (q == i || (k2 & 1ull) == anti) continue;
                const int64_t left_diff = (int64_t)gp - (int64_t)(k2 >> 30);
                const int64_t right_diff = ((int64_t)gp + (int64_t)len) - ((int64_t)(k2 >> 30) + (int64_t)((k2 >> 1) & ((1ull << 29) - 1)));
                if (left_diff < min_anchor || right_diff < min_anchor) {
                    const uint32_t s2 = jb_find(t, k2);
                    if (my_support < t.cnt1[s2]) ok = 0;
                }
            }
        }
        acc2[si] = ok;
    }
}

// second pass: the first occurrence of a record speaks for the record
__global__ __launch_bounds__(256) void thj_k_jb_second(JbTable t, const JbOcc* occ, int64_t n_occ, const uint32_t* acc2) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_occ; i += (int64_t)gridDim.x * blockDim.x) {
        const JbOcc o = occ[i];
        if (o.idx != 0) continue;
        bool ok = true;
        for (int k = 0; k < o.nj; ++k) { const uint32_t s = occ[i + k].slot; if (s == 0xFFFFFFFFu || !acc2[s]) ok = false; }
        if (!ok) continue;
        for (int k = 0; k < o.nj; ++k) {
            const JbOcc q = occ[i + k];
            atomicAdd(&t.cnt2[q.slot], 1u);
            atomicMax(&t.le2[q.slot], (uint32_t)q.le);
            atomicMax(&t.re2[q.slot], (uint32_t)q.re);
        }
    }
}

struct JbOut { u64 key; uint32_t support, le, re, left; };
__global__ __launch_bounds__(256) void thj_k_jb_gather(JbTable t, const u64* sorted, int64_t n, JbOut* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t s = jb_find(t, sorted[i]);
        out[i] = JbOut{sorted[i], t.cnt2[s], t.le2[s], t.re2[s], t.left[s]};
    }
}

// ------------------------------------------------------------------------------------------------ host side

static void jb_free(thj_ctx* c) {
    hipFree(c->d_jb_key); hipFree(c->d_jb_u32); hipFree(c->d_jb_list); hipFree(c->d_jb_cnt); hipFree(c->d_jb_occ); hipFree(c->d_jb_sorted);
    c->d_jb_key = nullptr; c->d_jb_u32 = nullptr; c->d_jb_list = nullptr; c->d_jb_cnt = nullptr; c->d_jb_occ = nullptr; c->d_jb_sorted = nullptr;
    c->jb_cap = 0; c->jb_occ_cap = 0;
}

static JbTable jb_table(thj_ctx* c) {
    uint32_t* u = c->d_jb_u32; const int64_t n = c->jb_cap;
    return JbTable{c->d_jb_key, (u64)n - 1, u, u + n, u + 2 * n, u + 3 * n, u + 4 * n, u + 5 * n, u + 6 * n, u + 7 * n, c->d_jb_list, c->d_jb_cnt};
}

static int jb_alloc(thj_ctx* c, int64_t cap) {
    int64_t p = 1 << 16;
    while (p < cap) p <<= 1;
    if (p == c->jb_cap) return THJ_OK;
    hipFree(c->d_jb_key); hipFree(c->d_jb_u32); hipFree(c->d_jb_list); hipFree(c->d_jb_sorted);
    c->d_jb_key = nullptr; c->d_jb_u32 = nullptr; c->d_jb_list = nullptr; c->d_jb_sorted = nullptr; c->jb_cap = 0;
    HIPCHK(hipMalloc(&c->d_jb_key, (size_t)p * 8));
    HIPCHK(hipMalloc(&c->d_jb_u32, (size_t)p * 4 * 9));        // cnt1 le1 re1 cnt2 le2 re2 left acc acc2
    HIPCHK(hipMalloc(&c->d_jb_list, (size_t)p * 8));
    HIPCHK(hipMalloc(&c->d_jb_sorted, (size_t)p * 8));
    if (!c->d_jb_cnt) HIPCHK(hipMalloc(&c->d_jb_cnt, 4 * sizeof(unsigned long long)));
    c->jb_cap = p;
    return THJ_OK;
}

extern "C" int thj_juncbed_configure(thj_ctx* c, int64_t junction_capacity) {
    if (!c || junction_capacity < 1) { thj_set_error("thj_juncbed_configure: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->jb_want = junction_capacity;
    return THJ_OK;
}

extern "C" int thj_juncbed_reset_async(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    HIPCHK(hipSetDevice(c->device));
    // twice the candidate set long_spanning_reads was given (its records cannot hold other junctions than those and the ones
    // already in spliced segment hits), at least 2^20 slots, or what thj_juncbed_configure asked for
    int64_t want = c->jb_want > 0 ? c->jb_want : (c->n_span_junc * 4 > (1 << 20) ? c->n_span_junc * 4 : (1 << 20));
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = jb_alloc(c, want);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(c->d_jb_key, 0xFF, (size_t)c->jb_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_jb_u32, 0, (size_t)c->jb_cap * 4 * 9, c->stream));
    HIPCHK(hipMemsetAsync(c->d_jb_cnt, 0, 4 * sizeof(unsigned long long), c->stream));
    c->jb_occ_used = 0;
    c->h_jb.clear();
    return THJ_OK;
}

static int jb_add(thj_ctx* c, const JbRecs& r) {
    if (!c->d_jb_key) { int rc = thj_juncbed_reset_async(c); if (rc) return rc; }
    const int64_t total = r.n_slots + r.n_extra;
    if (total == 0) return THJ_OK;
    int64_t blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    // occurrences of these records: counted first, so that the occurrence buffer is exactly large enough
    unsigned long long before = 0, after = 0;
    HIPCHK(hipMemcpyAsync(&before, &c->d_jb_cnt[1], 8, hipMemcpyDeviceToHost, c->stream));
    hipLaunchKernelGGL(thj_k_jb_count, dim3((unsigned)blocks), dim3(256), 0, c->stream, r, c->d_jb_cnt);
    HIPCHK(hipMemcpyAsync(&after, &c->d_jb_cnt[1], 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if ((int64_t)after > c->jb_occ_cap) {
        const int64_t ncap = (int64_t)after + (int64_t)after / 4 + 4096;
        JbOcc* n = nullptr;
        HIPCHK(hipMalloc(&n, (size_t)ncap * sizeof(JbOcc)));
        if (c->d_jb_occ && before) HIPCHK(hipMemcpy(n, c->d_jb_occ, (size_t)before * sizeof(JbOcc), hipMemcpyDeviceToDevice));
        hipFree(c->d_jb_occ);
        c->d_jb_occ = n; c->jb_occ_cap = ncap;
    }
    if (after == before) return THJ_OK;
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    hipLaunchKernelGGL(thj_k_jb_add, dim3((unsigned)blocks), dim3(256), 0, c->stream, g, r, jb_table(c), (JbOcc*)c->d_jb_occ, (unsigned long long)c->jb_occ_cap);
    HIPCHK(hipGetLastError());
    c->jb_occ_used = (int64_t)after;
    return THJ_OK;
}

extern "C" int thj_juncbed_add_span_async(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    JbRecs r{(const OutAln*)c->d_aln_pool, c->d_nrec, c->span_reads, (const OutAln*)c->d_aln_sorted, c->n_ovf, true};
    return jb_add(c, r);
}

extern "C" int thj_juncbed_add_records(thj_ctx* c, const thj_aln* recs, int64_t n, int32_t on_device) {
    if (!c || n < 0 || (n > 0 && !recs)) { thj_set_error("thj_juncbed_add_records: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return THJ_OK;
    for (int64_t i = 0; !on_device && i < n; ++i)
        if (recs[i].ref_id < 1 || (int32_t)recs[i].ref_id > c->n_contigs || recs[i].n_cigar > 16) { thj_set_error("record %lld: contig or cigar out of range", (long long)i); return THJ_EINVAL; }
    const thj_aln* d = recs;
    void* tmp = nullptr;
    if (!on_device) {
        HIPCHK(hipMalloc(&tmp, (size_t)n * sizeof(thj_aln)));
        HIPCHK(hipMemcpyAsync(tmp, recs, (size_t)n * sizeof(thj_aln), hipMemcpyHostToDevice, c->stream));
        d = (const thj_aln*)tmp;
    }
    JbRecs r{(const OutAln*)d, nullptr, n, nullptr, 0, false};
    int rc = jb_add(c, r);
    if (tmp) { hipStreamSynchronize(c->stream); hipFree(tmp); }
    return rc;
}

extern "C" int thj_juncbed_finish(thj_ctx* c, int32_t min_anchor_len, int64_t* n_juncs) {
    if (!c || min_anchor_len < 0 || min_anchor_len > 60) { thj_set_error("thj_juncbed_finish: bad argument (min_anchor_len 0..60)"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    c->h_jb.clear();
    if (n_juncs) *n_juncs = 0;
    if (!c->d_jb_key) return THJ_OK;
    unsigned long long h[4];
    HIPCHK(hipMemcpyAsync(h, c->d_jb_cnt, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (h[3] || (int64_t)h[0] > c->jb_cap - c->jb_cap / 4) {
        thj_set_error("junction table full (%llu distinct junctions, capacity %lld): call thj_juncbed_configure with a larger capacity and add the records again",
                      h[0], (long long)c->jb_cap);
        return THJ_EOVERFLOW;
    }
    const int64_t n = (int64_t)h[0];
    if (n == 0) return THJ_OK;
    JbTable t = jb_table(c);
    uint32_t* acc2 = c->d_jb_u32 + 8 * c->jb_cap;
    int64_t blocks = (c->jb_cap + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(thj_k_jb_accept, dim3((unsigned)blocks), dim3(256), 0, c->stream, t, n, (int)min_anchor_len);
    size_t need = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, (const u64*)c->d_jb_list, c->d_jb_sorted, n, 0, 64, c->stream));
    if (need > c->sort_tmp_bytes) { hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; HIPCHK(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
    size_t bytes = c->sort_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, bytes, (const u64*)c->d_jb_list, c->d_jb_sorted, n, 0, 64, c->stream));
    int64_t b2 = (n + 255) / 256; if (b2 > 4096) b2 = 4096;
    hipLaunchKernelGGL(thj_k_jb_knockout, dim3((unsigned)b2), dim3(256), 0, c->stream, t, (const u64*)c->d_jb_sorted, n, (int)min_anchor_len, acc2);
    // second pass (cnt2 / le2 / re2 start from zero: a finish can be repeated)
    HIPCHK(hipMemsetAsync(t.cnt2, 0, (size_t)c->jb_cap * 4 * 3, c->stream));
    if (c->jb_occ_used) {
        int64_t b3 = (c->jb_occ_used + 255) / 256; if (b3 > 4096) b3 = 4096;
        hipLaunchKernelGGL(thj_k_jb_second, dim3((unsigned)b3), dim3(256), 0, c->stream, t, (const JbOcc*)c->d_jb_occ, c->jb_occ_used, (const uint32_t*)acc2);
    }
    JbOut* d_out = nullptr;
    HIPCHK(hipMalloc(&d_out, (size_t)n * sizeof(JbOut)));
    hipLaunchKernelGGL(thj_k_jb_gather, dim3((unsigned)b2), dim3(256), 0, c->stream, t, (const u64*)c->d_jb_sorted, n, d_out);
    HIPCHK(hipGetLastError());
    std::vector<JbOut> out((size_t)n);
    HIPCHK(hipMemcpyAsync(out.data(), d_out, (size_t)n * sizeof(JbOut), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(d_out);
    for (auto& o : out) {
        if (o.support == 0 || o.le < 8 || o.re < 8) continue;                          // tophat_reports.cpp:2974-2984
        thj_juncstat s;
        const int64_t gp = (int64_t)(o.key >> 30) - 1;                                   // global coordinate of `left`
        int lo = 0, hi = c->n_contigs;
        const int64_t start = gp - (int64_t)o.left;                                      // = contig start (left is contig-relative)
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int64_t)c->h_contig_blk[(size_t)mid] * 64 <= start) lo = mid; else hi = mid; }
        s.ref_id = (uint32_t)lo + 1; s.left = o.left; s.right = o.left + (uint32_t)((o.key >> 1) & ((1ull << 29) - 1)); s.antisense = (uint32_t)(o.key & 1ull);
        s.left_extent = o.le; s.right_extent = o.re; s.support = o.support; s.reserved = 0;
        c->h_jb.push_back(s);
    }
    if (n_juncs) *n_juncs = (int64_t)c->h_jb.size();
    return THJ_OK;
}

extern "C" int thj_juncbed_download(thj_ctx* c, thj_juncstat* out) {
    if (!c || (!c->h_jb.empty() && !out)) { thj_set_error("thj_juncbed_download: bad argument"); return THJ_EINVAL; }
    if (!c->h_jb.empty()) memcpy(out, c->h_jb.data(), c->h_jb.size() * sizeof(thj_juncstat));
    return THJ_OK;
}
