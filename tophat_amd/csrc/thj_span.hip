// thj_span.hip -- gfx950 kernel + C ABI for the long_spanning_reads hot path.
//
//   thj_k_stitch_contig / thj_k_stitch (1 thread / read), thj_k_stitch_pack (1 lane / chain): three tiers of the same
//   algorithm -- DFS over one hit per segment (dfs_seg_hits), closure of every adjacent pair through the
//   sorted junction / insertion key arrays (merge_chain), edit-distance consistency, sort/unique/filter and
//   the AS/XM/XO/XG/MD pass (bowtie_sam_extra).  128-byte records land in one slot per read, so the device
//   layout is already the order of the reference's BAM (read order, BowtieHit::operator< inside a read).
// Integer / bit-plane work; no MFMA.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstddef>
#include <chrono>
#include <vector>

#include "../../include/thj.h"
#include "thj_span_core.h"
#include "thj_span_fusion.h"
#include "thj_ctx.h"
#include "thj_scan.h"

using namespace thj;


static_assert(sizeof(thj_span_hit) == 32 && sizeof(SpanHit) == 32, "span hit layout");
static_assert(sizeof(thj_aln) == 128 && sizeof(OutAln) == 128 && sizeof(thj_aln_slot) == 128 && offsetof(thj_aln_slot, cigar_hi) == 64, "aln layout");

struct DevSpanBatch {
    int32_t n_reads, nseg, W, qual_stride;
    const uint32_t* seg_off;
    const SpanHit* hits;
    const u64* planes;
    const uint16_t* read_len;
    const uint8_t* quals;
    // optional: the first 16 bytes of every hit record, densely (thj_span_hit_heads_async).  Tier 0 streams it instead of the
    // 32-byte records (-14 % on that kernel); tiers 1 / 2 gather single reads, where head and tail sharing one cache line
    // is worth more (measured +3 % with the dense array), so they keep reading the records.
    const SpanHitHead* heads;
};
static_assert(sizeof(DevSpanBatch) == sizeof(thj_span_batch), "span batch layout");

// Output layout: one 128-byte slot per read of the pass (batches in run order, reads in batch order) holding the
// read's first record, a per-read record count, and a small overflow pool for the 2nd.. records of multihit reads.
// Walking the slots in order IS the order of the reference's BAM (read order, BowtieHit::operator< inside a read),
// so no sort or gather of the records is needed.
//
// A slot is two 64-byte lines (thj_aln_slot, include/thj.h).  The lead line holds everything an ordinary alignment has:
// the 24-byte header, cigar ops 0..3, MD characters 0..23.  The tail line (cigar ops 4..15, MD characters 24..39) is
// written -- and THJ_SLOT_TAIL set in the lead line's flags -- only by a record that needs it: more than four cigar
// ops, an MD string of more than 24 characters, or a fusion alignment (second contig in the last cigar slot).  A plain
// `100M` / `60M2000N40M` record therefore costs one 64-byte line of HBM write traffic, not two; the tail line of such a
// slot is never touched and holds whatever an earlier pass left there.  thj_span_download hands out API-layout thj_aln.
static constexpr uint32_t SLOT_TAIL = 0x80u;     // == THJ_SLOT_TAIL, in the flags byte (word 3, bits 0..7)
__device__ __forceinline__ bool slot_needs_tail(const uint32_t* w) {     // w: the 32 words of an API-layout record
    const uint32_t n_cigar = w[3] >> 24, md_len = (w[5] >> 8) & 0xFFu;
    return n_cigar > 4u || (md_len > 24u && md_len != 255u) || w[21] != 0u;
}
// slot word k of an API-layout record (compile-time k after unrolling)
__device__ __forceinline__ uint32_t slot_word(const uint32_t* w, int k, bool tail) {
    if (k == 3) return w[3] | (tail ? SLOT_TAIL : 0u);
    if (k < 10) return w[k];                 // header, cigar 0..3
    if (k < 16) return w[k + 12];            // md bytes 0..23  (API words 22..27)
    if (k < 28) return w[k - 6];             // cigar 4..15     (API words 10..21)
    return w[k];                             // md bytes 24..39 (API words 28..31)
}
struct RecSink {
    OutAln* slots; uint8_t* nrec; uint32_t base;
    OutAln* ovf; u64* ovf_key; unsigned long long* ovf_count; unsigned long long ovf_cap;
    unsigned long long* total; unsigned int* status;
    int emitted;       // per-thread: records emitted for the current read
    int acc;           // per-thread: records emitted so far (summed per block at the end of the kernel)
    __device__ __forceinline__ void emit_words(const uint32_t* w) {
        ++emitted;
        const uint32_t order = w[5] >> 16;
        uint4* dst;
        if (order == 0) dst = (uint4*)(slots + (size_t)base + w[0]);
        else {
            // (the lanes of a wave that come here together -- the packed tier's chains of a read, a lane each -- take their places in the pool
            // with one add: a launch of the mix puts half a million records there, on one address)
            unsigned long long pos;
            if (THJ_EXPF(8192)) pos = (unsigned long long)(base + w[0]);
            else {
                const unsigned long long act = __ballot(1);
                const int lane = (int)(threadIdx.x & 63u);
                unsigned long long b0 = 0;
                if (lane == __ffsll((long long)act) - 1) b0 = atomicAdd(ovf_count, (unsigned long long)__popcll(act));
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b0), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b0 >> 32));
                pos = (((unsigned long long)hi << 32) | lo) + (unsigned long long)__popcll(act & (lane ? (~0ull >> (64 - lane)) : 0ull));
            }
            if (pos >= ovf_cap) { atomicExch(&status[3], 1u); return; }
            ovf_key[pos] = ((u64)(base + w[0]) << 16) | (u64)order;
            dst = (uint4*)(ovf + pos);
        }
        if (THJ_EXPF(32768)) { if (w[7] == 0x12345u) dst[0] = make_uint4(w[0], w[9], w[18], w[31]); return; }
        const bool tail = slot_needs_tail(w);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_uint4(slot_word(w, 4 * k, tail), slot_word(w, 4 * k + 1, tail), slot_word(w, 4 * k + 2, tail), slot_word(w, 4 * k + 3, tail));
        if (tail) {
#pragma unroll
            for (int k = 4; k < 8; ++k) dst[k] = make_uint4(slot_word(w, 4 * k, true), slot_word(w, 4 * k + 1, true), slot_word(w, 4 * k + 2, true), slot_word(w, 4 * k + 3, true));
        }
    }
    // the packed tier: the records of read r were written by several lanes, one of them reports their number
    __device__ __forceinline__ void set_count(uint32_t r, int n) { nrec[(size_t)base + r] = (uint8_t)(n > 255 ? 255 : n); acc += n; }
    __device__ __forceinline__ void done(uint32_t r) { nrec[(size_t)base + r] = (uint8_t)(emitted > 255 ? 255 : emitted); acc += emitted; emitted = 0; }   // the count saturates: consumers test it for zero
};

// The three tiers hand reads down through worklists.  A global append counter would serialise every wave of the
// launch on one L2 address (measured: 1.5 ms of a 2 ms kernel), so the lists are segmented instead: block b owns
// reads [b*chunk, (b+1)*chunk) in tier 0 and the slice [b*chunk, ..) of both lists in every tier; positions come
// from an LDS counter, per-block totals go to blk_lean / blk_multi, and global counters see one add per block.
struct Tiers {
    uint32_t* wl_lean; uint32_t* wl_multi; uint32_t* wl_gen;
    unsigned int* blk_lean; unsigned int* blk_multi; unsigned int* blk_gen;
    unsigned int* counters;          // reads handed to [0] tier 1, [1] tier 2, [2] tier 3 (this run)
    int chunk;
    // reads with more joined alignments than a thread's own array holds (a read in a large repeat family): listed here when the
    // context owns the big workspace (thj_k_stitch_huge does them again with room), else reported (SPAN_TOO_MANY_JOINED)
    uint32_t* huge_list; unsigned int* huge_cnt; int huge_list_cap;
    // chain entries (tier 0 -> thj_k_join, see thj_span_core.h): class-major block-owned slices like wl_lean, 32 bytes an entry;
    // null: every one-hit-per-segment read takes wl_lean.  The joined hits, densely: ja / jb / jc[i] for entry i of the concatenation
    ChainEntry* ent; unsigned int* blk_chain;
    Q16* ja; Q16* jb; Q16* jc;
    unsigned int* status;            // the pass's status words (RecSink::status)
};
__device__ __forceinline__ bool defer_huge(const Tiers& t, uint32_t r) {
    if (!t.huge_list) return false;
    const unsigned int k = atomicAdd(t.huge_cnt, 1u);
    if (k >= (unsigned int)t.huge_list_cap) return false;
    t.huge_list[k] = r;
    return true;
}
// Tier 1's list is class-major on top of that (SPAN_LEAN_CLASSES x G slices: the reads whose gap lean_join meets in the same
// loop iteration sit together, so a wave runs the closure code once, not once per gap position); batches of more than
// four segments per read keep one class -- their kernel has no LDS to spare for the longer offset table.
__host__ __device__ constexpr int lean_classes(int MS) { return MS <= 4 ? SPAN_LEAN_CLASSES : 1; }

// Tier 0: every read.  Reads made of abutting single plain-match hits (unspliced reads cut into segments) are
// finished here with a handful of registers.  A block works on 256 consecutive reads at a time: their records are
// assembled in LDS (chunk-rotated so the 16-byte writes of a wave spread over all banks) and leave as full
// 128-byte lines, eight lanes per record.
struct StageSink {
    uint4* stage; int rec; int emitted;      // emitted: 0 none, 1 lead line only, 2 lead + tail
    __device__ __forceinline__ void emit_words(const uint32_t* w) {
        const bool tail = slot_needs_tail(w);
        emitted = tail ? 2 : 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) stage[rec * 8 + ((k + rec) & 7)] = make_uint4(slot_word(w, 4 * k, tail), slot_word(w, 4 * k + 1, tail), slot_word(w, 4 * k + 2, tail), slot_word(w, 4 * k + 3, tail));
        if (tail) {
#pragma unroll
            for (int k = 4; k < 8; ++k) stage[rec * 8 + ((k + rec) & 7)] = make_uint4(slot_word(w, 4 * k, true), slot_word(w, 4 * k + 1, true), slot_word(w, 4 * k + 2, true), slot_word(w, 4 * k + 3, true));
        }
    }
};

// MS: upper bound of the batch's segments per read (4 covers reads up to 4 x segment_length, e.g. 100 bp at 25)
template <int MS>
__global__ __launch_bounds__(256, 4) void thj_k_stitch_contig(Genome g, Params p, DevSpanBatch b, RecSink sink, Tiers t) {
    __shared__ uint4 stage[256 * 8];
    __shared__ uint8_t has_rec[256];
    __shared__ unsigned int s_cnt[3];          // (unused), multihit, records
    __shared__ unsigned int s_lean[SPAN_LEAN_CLASSES], s_chain[SPAN_LEAN_CLASSES];
    constexpr int NC = lean_classes(MS);
    const int tid = threadIdx.x;
    if (tid < 3) s_cnt[tid] = 0;
    if (tid < SPAN_LEAN_CLASSES) { s_lean[tid] = 0; s_chain[tid] = 0; }
    __syncthreads();
    // reads are numbered in 32 bits (n_reads < 2^31 is checked on the host); offsets are one 32x32->64 multiply each
    const uint32_t c0 = blockIdx.x * (uint32_t)t.chunk;
    const uint32_t c1 = c0 + (uint32_t)t.chunk < (uint32_t)b.n_reads ? c0 + (uint32_t)t.chunk : (uint32_t)b.n_reads;
    unsigned int my_rec = 0;
    // Segment offsets and read length are fetched one read ahead: one dependent round trip less per read.  (Going one
    // step further -- hit heads one read ahead, offsets two -- was measured slower: the registers it takes cost more
    // in occupancy and spills than the round trip it saves.)
    uint32_t sv_next[MS + 1]; int rl_next = 0;
#pragma unroll
    for (int s = 0; s <= MS; ++s) sv_next[s] = 0;
    if (c0 + (uint32_t)tid < c1) { contig_offsets<MS>(b.seg_off + (u64)(c0 + tid) * (uint32_t)b.nseg, b.nseg, sv_next); rl_next = (int)b.read_len[c0 + tid]; }
    for (uint32_t r0 = c0; r0 < c1; r0 += 256) {
        const uint32_t r = r0 + (uint32_t)tid;
        StageSink ss{stage, tid, 0};
        uint32_t sv[MS + 1];
#pragma unroll
        for (int s = 0; s <= MS; ++s) sv[s] = sv_next[s];
        const int rl = rl_next;
        if (r + 256 < c1) { contig_offsets<MS>(b.seg_off + (u64)(r + 256) * (uint32_t)b.nseg, b.nseg, sv_next); rl_next = (int)b.read_len[r + 256]; }
        if (r < c1) {
            ChainEntry ent;
            int st = span_read_contig_pre<MS>(g, p, b.hits, sv, b.nseg, b.planes + (u64)r * (uint32_t)(3 * b.W), b.W,
                                          rl, b.quals + (u64)r * (uint32_t)b.qual_stride, r, ss, b.heads, MS <= CHAIN_MAXSEG ? &ent : nullptr, t.ent != nullptr);
            if (MS <= CHAIN_MAXSEG && (st & 0xFF) == SPAN_NEED_CHAIN) {
                const unsigned int cls = (unsigned int)(st >> 8);
                Q16* dst = (Q16*)(t.ent + (u64)(cls * gridDim.x + blockIdx.x) * (uint32_t)t.chunk + atomicAdd(&s_chain[cls], 1u));
                dst[0] = Q16{ent.read, ent.meta, ent.spare0, ent.spare1};
                dst[1] = Q16{ent.hit[0], ent.hit[1], ent.hit[2], ent.hit[3]};
            }
            else if ((st & 0xFF) == SPAN_NEED_LEAN) {
                const unsigned int cls = NC > 1 ? (unsigned int)(st >> 8) : 0u;
                t.wl_lean[(u64)(cls * gridDim.x + blockIdx.x) * (uint32_t)t.chunk + atomicAdd(&s_lean[cls], 1u)] = r;
            }
            else if (st == SPAN_NEED_GENERIC) t.wl_multi[c0 + atomicAdd(&s_cnt[1], 1u)] = r;
            else {
                my_rec += ss.emitted != 0;
                if (st) atomicAdd(&sink.status[st], 1u);
            }
            // every read's record count starts here (a read handed on: 0 until the kernel that takes it says otherwise) -- a byte a lane, a
            // wave's 64 side by side; the 10 MB memset in front of the kernel that did this before took 20-500 us of the side's stream
            sink.nrec[(size_t)sink.base + r] = (uint8_t)(ss.emitted != 0);
        }
        has_rec[tid] = (uint8_t)ss.emitted;
        __syncthreads();
        uint4* out = (uint4*)(sink.slots + (size_t)sink.base + r0);
        // lead lines: four lanes per record, 64 contiguous bytes each
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int j = tid + 256 * m, i = j >> 2, ch = j & 3;
            if (has_rec[i] && !THJ_EXPF(1)) out[i * 8 + ch] = stage[i * 8 + ((ch + i) & 7)];
        }
        // tail lines of the few records that have one
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int j = tid + 256 * m, i = j >> 2, ch = 4 + (j & 3);
            if (has_rec[i] == 2 && !THJ_EXPF(1)) out[i * 8 + ch] = stage[i * 8 + ((ch + i) & 7)];
        }
        __syncthreads();
    }
    if (my_rec) atomicAdd(&s_cnt[2], my_rec);
    __syncthreads();
    if (tid == 0) {
        unsigned int n_lean = 0, n_chain = 0;
        for (int k = 0; k < NC; ++k) { t.blk_lean[k * gridDim.x + blockIdx.x] = s_lean[k]; n_lean += s_lean[k]; }
        if (t.ent) for (int k = 0; k < SPAN_LEAN_CLASSES; ++k) { t.blk_chain[k * gridDim.x + blockIdx.x] = s_chain[k]; n_chain += s_chain[k]; }
        t.blk_multi[blockIdx.x] = s_cnt[1];
        if (n_lean + n_chain) atomicAdd(&t.counters[0], n_lean + n_chain);
        if (n_chain) atomicAdd(&t.counters[6], n_chain);
        if (s_cnt[1]) atomicAdd(&t.counters[1], s_cnt[1]);
        if (s_cnt[2]) atomicAdd(sink.total, (unsigned long long)s_cnt[2]);
    }
}

// Tiers 1 and 2 walk the concatenation of the per-block slices so that every lane has work whatever the spread of
// spliced / multihit reads over the batch: each block scans the (at most 2048) slice lengths into LDS, and entry i
// of the concatenation is found by a binary search there.
static constexpr int MAX_SLICES = 1024;
template <int TPB, int NS = MAX_SLICES>
__device__ unsigned int slice_offsets(const unsigned int* blk_cnt, int G, unsigned int* s_off /* [NS + 1] */) {
    constexpr int IPT = NS / TPB;
    typedef hipcub::BlockScan<unsigned int, TPB> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const int tid = threadIdx.x;
    unsigned int v[IPT], sum = 0, excl, total;
#pragma unroll
    for (int k = 0; k < IPT; ++k) { const int j = tid * IPT + k; v[k] = j < G ? blk_cnt[j] : 0u; sum += v[k]; }
    Scan(tmp).ExclusiveSum(sum, excl, total);
#pragma unroll
    for (int k = 0; k < IPT; ++k) { s_off[tid * IPT + k] = excl; excl += v[k]; }
    if (tid == 0) s_off[NS] = total;
    __syncthreads();
    return total;
}
__device__ __forceinline__ int slice_of(const unsigned int* s_off, int G, unsigned int i) {   // last b with s_off[b] <= i
    int lo = 0, hi = G;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}

// Tier 1: single-hit-per-segment reads that need closures (spliced / indel reads): streamed merge_chain on registers.
template <int MS, int WPE = 4>
__global__ __launch_bounds__(256, WPE) void thj_k_stitch(Genome g, Params p, SpanSets S, DevSpanBatch b, RecSink sink, Tiers t, int G) {
    extern __shared__ uint4 lds_stage[];          // nseg hit heads per thread
    constexpr int NC = lean_classes(MS);
    __shared__ unsigned int s_off[NC * MAX_SLICES + 1];
    __shared__ unsigned int s_rec, s_fwd;
    if (threadIdx.x == 0) { s_rec = 0; s_fwd = 0; }
    unsigned int n_fwd = 0;
    SpanHitHead* stage = (SpanHitHead*)lds_stage + (size_t)threadIdx.x * b.nseg;
    const unsigned int total = slice_offsets<256, NC * MAX_SLICES>(t.blk_lean, NC * G, s_off);
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int sl = slice_of(s_off, NC * G, i);
        const int r = (int)t.wl_lean[(int64_t)sl * t.chunk + (i - s_off[sl])];
        int st = span_read_lean<MS>(g, p, S, b.hits, b.seg_off + (size_t)r * b.nseg, b.nseg, b.planes + (size_t)r * 3 * b.W, b.W,
                                (int)b.read_len[r], b.quals + (size_t)r * b.qual_stride, (uint32_t)r, stage, sink);
        // fusion search: a one-hit-per-segment read that joins the plain way joins the same way with fusion search on (its only
        // chain never takes a fusion direction); one that does not may be a fusion read -- thj_k_stitch_fusion decides
        // -- its only chain takes a fusion direction exactly when two neighbours are not compatible the plain way (SPAN_INCOMPAT);
        // a compatible chain that does not join fails the same way there, so only the former go on (a third of what went on when
        // every read without a record did)
        if (st == SPAN_INCOMPAT) st = p.fusion_search ? SPAN_NEED_GENERIC : SPAN_OK;
        const bool fwd = st == SPAN_NEED_GENERIC;       // rare without fusion search: more cigar ops than the registers hold
        if (!fwd) { sink.done((uint32_t)r); if (st) atomicAdd(&sink.status[st], 1u); }
        sl %= G;                                // the slice of the block of tier 0 that owns the read
        // with fusion search on every unjoined read comes this way (one read in eight of this tier on configs[3]'s shape): the lanes
        // of a wave that forward to the same slice -- neighbours in the list almost always do -- take their places with one atomic
        for (unsigned long long todo = __ballot(fwd); todo;) {
            const int leader = __ffsll((long long)todo) - 1;
            const int sl_l = __shfl(sl, leader);
            const unsigned long long same = __ballot(fwd && sl == sl_l);
            const int lane = (int)(threadIdx.x & 63u);
            unsigned int base = 0;
            if (lane == leader) base = atomicAdd(&t.blk_multi[sl_l], (unsigned int)__popcll(same));
            base = (unsigned int)__shfl((int)base, leader);
            if (fwd && sl == sl_l) {
                t.wl_multi[(int64_t)sl * t.chunk + base + (unsigned int)__popcll(same & (lane ? (~0ull >> (64 - lane)) : 0ull))] = (uint32_t)r;
                ++n_fwd;
            }
            todo &= ~same;
        }
    }
    if (sink.acc) atomicAdd(&s_rec, (unsigned int)sink.acc);
    if (n_fwd) atomicAdd(&s_fwd, n_fwd);
    __syncthreads();
    if (threadIdx.x == 0 && s_rec) atomicAdd(sink.total, (unsigned long long)s_rec);
    if (threadIdx.x == 0 && s_fwd) atomicAdd(&t.counters[1], s_fwd);
}

// The join of the chain entries (merge_chain, long_spanning_reads.cpp:805-2038, on register cigars): a thread per entry of the
// concatenated class-major slices.  The entry names the chain's hit records; they are fetched whole and all together (eight
// 16-byte loads in flight), parked in the thread's own column of LDS (lean_join picks hits by a running index), and the join
// touches global memory again only for the junction keys its closures look at and -- when a boundary moves -- a few genome and
// read words.  Entry i's joined hit is ja / jb / jc[i]; an entry that does not join leaves JOINED_NONE; one that needs more cigar
// ops than the registers hold goes to the general tier's list.
struct LdsChainHits {       // word pair (2 s, 2 s + 1) of column `col` = the record of the chain's segment s
    const Q16* col;
    __device__ __forceinline__ SpanHit operator[](int s) const {
        const Q16 a = col[(2 * s) * 256], b = col[(2 * s + 1) * 256];
        SpanHit h;
        h.ref_id = a.x; h.left = (int32_t)a.y; h.meta = a.z; h.cigar[0] = a.w;
        h.cigar[1] = b.x; h.cigar[2] = b.y; h.cigar[3] = b.z; h.cigar[4] = b.w;
        return h;
    }
};
// thj_k_join's own: six words a hit (records of at most three cigar ops: everything but a segment hit with two splices, which the
// closure kernel takes) -- 24 KB a workgroup instead of 32, six workgroups a CU instead of four
struct LdsChainHits6 {
    const Q16* col; const uint2* col2;
    __device__ __forceinline__ SpanHit operator[](int s) const {
        const Q16 a = col[s * 256]; const uint2 b = col2[s * 256];
        SpanHit h;
        h.ref_id = a.x; h.left = (int32_t)a.y; h.meta = a.z; h.cigar[0] = a.w;
        h.cigar[1] = b.x; h.cigar[2] = b.y; h.cigar[3] = 0; h.cigar[4] = 0;
        return h;
    }
};
// (Work distribution.  Region A, tier 0's entries: workgroup b takes what tier 0's workgroup b wrote -- its four class slices one
// after the other -- and writes the joined hits to J[b * chunk ..): no table of slice offsets, no search per entry.  Region B,
// the chains of multihit reads (thj_k_chains): one dense list of *n2 entries, workgroups G .. G + G2 stride over it, joined hits
// at J[G * chunk ..).  thj_k_join_closure and thj_k_finish walk the same way.)
struct ChainLists {
    const ChainEntry* ent; const unsigned int* blk_cnt; int G, chunk;
    const ChainEntry* ent2; const unsigned int* n2; int G2; unsigned int cap2, slice2;
    Q16* ja; Q16* jb; Q16* jc;
};
__device__ __forceinline__ unsigned int chain_block_counts(const ChainLists& L, int blk, unsigned int (&c)[SPAN_LEAN_CLASSES]) {
    unsigned int total = 0;
#pragma unroll
    for (int k = 0; k < SPAN_LEAN_CLASSES; ++k) { c[k] = L.blk_cnt[k * L.G + blk]; total += c[k]; }
    return total;
}
__device__ __forceinline__ unsigned int chain_dense_count(const ChainLists& L) { const unsigned int n = *L.n2; return n < L.cap2 ? n : L.cap2; }
// Two kernels.  Most chains of a sample mapped against a junction database abut everywhere -- the spliced read's junction sits
// INSIDE a segment hit (aM gN bM) and merge_chain only concatenates -- and need nothing but their records; a chain with a gap
// between two hits (a junction at a segment boundary, an indel) needs the closure search: dependent loads of junction keys,
// genome and read words that a wave pays for as a whole even when one lane takes them (one lane in seven does here: every wave
// would).  thj_k_join joins the abutting chains (lean_join<ABUT>: the closure code is not in it) and lists the others, per
// workgroup; thj_k_join_closure runs the full join over those lists with its lanes dense.  (As two passes of one kernel the
// second pass ran cold: 230 us per workgroup for some four hundred entries, most of it instruction fetch.)
struct DeferList { uint32_t* idx; unsigned int* cnt; };      // region A: idx[b * chunk ..), cnt[b]; region B: idx[G * chunk + j * slice2 ..), cnt[G + j]
template <bool ABUT>
__device__ __forceinline__ int join_entry(const Genome& g, const Params& p, const SpanSets& S, const SpanHit* hits, const u64* planes, int W,
                                          const ChainLists& L, const Tiers& t, Q16* s_rec, const ChainEntry* entry, u64 at) {
    const Q16* src = (const Q16*)entry;
    const Q16 e0 = src[0], e1 = src[1];
    const uint32_t r = e0.x, meta = e0.y;
    if (r == JOINED_PAD) { if (ABUT) L.ja[at] = Q16{JOINED_PAD, 0u, 0u, 0u}; return LJ_NONE; }       // padding of a group (thj_k_chains)
    RAln res;
    int jr;
    {
        Q16 rec[2 * CHAIN_MAXSEG];
        const uint32_t hi[CHAIN_MAXSEG] = {e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int k = 0; k < CHAIN_MAXSEG; ++k) { const Q16* hp = (const Q16*)(hits + hi[k]); rec[2 * k] = hp[0]; rec[2 * k + 1] = hp[1]; }
        if (ABUT) {
            bool wide = false;
#pragma unroll
            for (int k = 0; k < CHAIN_MAXSEG; ++k) wide = wide || (rec[2 * k].z >> 24) > 3u;
            if (wide) return LJ_DEFER;
            uint2* s2 = (uint2*)(s_rec + CHAIN_MAXSEG * 256);
#pragma unroll
            for (int k = 0; k < CHAIN_MAXSEG; ++k) { s_rec[k * 256 + threadIdx.x] = rec[2 * k]; s2[k * 256 + threadIdx.x] = make_uint2(rec[2 * k + 1].x, rec[2 * k + 1].y); }
            const LdsChainHits6 ch{s_rec + threadIdx.x, s2 + threadIdx.x};
            jr = chain_join<ABUT>(g, p, S, ch, meta, planes + (u64)r * (uint32_t)(3 * W), W, res);
        } else {
#pragma unroll
            for (int k = 0; k < 2 * CHAIN_MAXSEG; ++k) s_rec[k * 256 + threadIdx.x] = rec[k];
            const LdsChainHits ch{s_rec + threadIdx.x};
            jr = chain_join<ABUT>(g, p, S, ch, meta, planes + (u64)r * (uint32_t)(3 * W), W, res);
        }
    }
    if (ABUT && jr == LJ_DEFER) return jr;
    Q16 ja, jb, jc;
    joined_pack(res, r, chain_nsegs(meta) == 1, chain_q(meta), chain_k(meta), ja, jb, jc);
    if (jr != LJ_OK) ja.w = joined_none_meta(chain_q(meta), chain_k(meta));
    L.ja[at] = ja;
    if (jr == LJ_OK) { L.jb[at] = jb; if (res.n > 4) L.jc[at] = jc; }
    if (jr == LJ_PUNT) {
        // rare: more cigar ops than the registers hold.  A read on its own goes to the general tier (its slice of that list is its
        // tier-0 block's); one chain of several cannot take its siblings' records back: the pass fails loudly (status[5])
        if (chain_k(meta) > 1) atomicExch(&t.status[5], 1u);
        else {
            const uint32_t gb = r / (uint32_t)t.chunk;
            t.wl_gen[(u64)gb * (uint32_t)t.chunk + atomicAdd(&t.blk_gen[gb], 1u)] = r;
            atomicAdd(&t.counters[2], 1u);
        }
    }
    return jr;
}
__device__ __forceinline__ const ChainEntry* chain_locate(const unsigned int (&c)[SPAN_LEAN_CLASSES], const ChainLists& L, int blk, unsigned int i) {
    unsigned int cls = 0, local = i;
#pragma unroll
    for (int k = 0; k < SPAN_LEAN_CLASSES - 1; ++k) { const bool next = cls == (unsigned int)k && local >= c[k]; local -= next ? c[k] : 0u; cls += next ? 1u : 0u; }
    return L.ent + (u64)(cls * (unsigned int)L.G + (unsigned int)blk) * (unsigned int)L.chunk + local;
}
template <int WPE>
__global__ __launch_bounds__(256, WPE) void thj_k_join(Genome g, Params p, SpanSets S, const SpanHit* hits, const u64* planes, int W, ChainLists L, Tiers t, DeferList D) {
    __shared__ Q16 s_rec[(CHAIN_MAXSEG + CHAIN_MAXSEG / 2) * 256];        // four hits' first four words, then their next two (LdsChainHits6)
    __shared__ unsigned int s_qn;
    if (threadIdx.x == 0) s_qn = 0;
    __syncthreads();
    if ((int)blockIdx.x < L.G) {
        const int blk = (int)blockIdx.x;
        unsigned int c[SPAN_LEAN_CLASSES];
        const unsigned int total = chain_block_counts(L, blk, c);
        for (unsigned int i = threadIdx.x; i < total; i += 256) {
            const int jr = join_entry<true>(g, p, S, hits, planes, W, L, t, s_rec, chain_locate(c, L, blk, i), (u64)blk * (uint32_t)L.chunk + i);
            if (jr == LJ_DEFER) D.idx[(u64)blk * (uint32_t)L.chunk + atomicAdd(&s_qn, 1u)] = i;
        }
    } else {
        const unsigned int j = blockIdx.x - (unsigned int)L.G, n2 = chain_dense_count(L);
        const u64 jbase = (u64)L.G * (uint32_t)L.chunk;
        for (unsigned int i = j * 256u + threadIdx.x; i < n2; i += (unsigned int)L.G2 * 256u) {
            const int jr = join_entry<true>(g, p, S, hits, planes, W, L, t, s_rec, L.ent2 + i, jbase + i);
            if (jr == LJ_DEFER) D.idx[jbase + (u64)j * L.slice2 + atomicAdd(&s_qn, 1u)] = i;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) D.cnt[blockIdx.x] = s_qn;
}
template <int WPE>
__global__ __launch_bounds__(256, WPE) void thj_k_join_closure(Genome g, Params p, SpanSets S, const SpanHit* hits, const u64* planes, int W, ChainLists L, Tiers t, DeferList D) {
    __shared__ Q16 s_rec[2 * CHAIN_MAXSEG * 256];
    const unsigned int n = D.cnt[blockIdx.x];
    if ((int)blockIdx.x < L.G) {
        const int blk = (int)blockIdx.x;
        unsigned int c[SPAN_LEAN_CLASSES];
        chain_block_counts(L, blk, c);
        for (unsigned int k = threadIdx.x; k < n; k += 256) {
            const unsigned int i = D.idx[(u64)blk * (uint32_t)L.chunk + k];
            join_entry<false>(g, p, S, hits, planes, W, L, t, s_rec, chain_locate(c, L, blk, i), (u64)blk * (uint32_t)L.chunk + i);
        }
    } else {
        const unsigned int j = blockIdx.x - (unsigned int)L.G;
        const u64 jbase = (u64)L.G * (uint32_t)L.chunk;
        for (unsigned int k = threadIdx.x; k < n; k += 256) {
            const unsigned int i = D.idx[jbase + (u64)j * L.slice2 + k];
            join_entry<false>(g, p, S, hits, planes, W, L, t, s_rec, L.ent2 + i, jbase + i);
        }
    }
}

// The finish of the joined hits (check_editdist_consistency, bowtie_sam_extra, the record: bwt_map.cpp:2349-2648, :1888-2093): a
// thread per joined hit, the lists walked as thj_k_join walks them.  The chains of a multihit read sit in adjacent lanes, in rank
// order (groups never straddle a wave, thj_k_chains): a record's rank among the read's records is the number of lower-ranked
// siblings that are reported -- a ballot.  Second and later records go to the extra pool; a wave reserves the room for all its lanes'
// with one atomic.
struct FinishSink {
    uint4* dst;
    __device__ __forceinline__ void emit_words(const uint32_t* w) {
        const bool tail = slot_needs_tail(w);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = make_uint4(slot_word(w, 4 * k, tail), slot_word(w, 4 * k + 1, tail), slot_word(w, 4 * k + 2, tail), slot_word(w, 4 * k + 3, tail));
        if (tail) {
#pragma unroll
            for (int k = 4; k < 8; ++k) dst[k] = make_uint4(slot_word(w, 4 * k, true), slot_word(w, 4 * k + 1, true), slot_word(w, 4 * k + 2, true), slot_word(w, 4 * k + 3, true));
        }
    }
};
__device__ __forceinline__ unsigned int finish_one(const Genome& g, const Params& p, const DevSpanBatch& b, const RecSink& sink, const ChainLists& L, bool has, u64 at) {
    const int lane = (int)(threadIdx.x & 63u);
    Q16 ja{JOINED_PAD, 0u, 0u, 0u};
    if (has) ja = L.ja[at];
    const bool real = ja.x != JOINED_PAD;
    const uint32_t r = ja.x;
    RAln res; Extras e;
    bool emit = false;
    if (real && joined_n(ja.w) > 0) {
        const Q16 jb = L.jb[at];
        Q16 jc{0u, 0u, 0u, 0u};
        if (joined_n(ja.w) > 4) jc = L.jc[at];
        emit = joined_prepare(g, p, ja, jb, jc, b.planes, b.W, b.read_len, b.quals, b.qual_stride, res, e);
    }
    const int q = joined_q(ja.w), kp = chains_padded(joined_k(ja.w));
    const unsigned long long m = __ballot(emit);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const unsigned long long gmask = (~0ull >> (64 - kp)) << (lane - q);
    const int order = __popcll(m & gmask & below), cnt = __popcll(m & gmask);
    const unsigned long long xm = __ballot(emit && order > 0);
    unsigned long long pbase = 0;
    if (xm) {                                    // (wave-uniform)
        const int leader = __ffsll((long long)xm) - 1;
        if (lane == leader) pbase = atomicAdd(sink.ovf_count, (unsigned long long)__popcll(xm));
        pbase = ((unsigned long long)(uint32_t)__shfl((int)(pbase >> 32), leader) << 32) | (uint32_t)__shfl((int)(uint32_t)pbase, leader);
    }
    if (emit) {
        uint4* dst = (uint4*)(sink.slots + (size_t)sink.base + r);
        if (order > 0) {
            const unsigned long long pos = pbase + (unsigned long long)__popcll(xm & below);
            if (pos >= sink.ovf_cap) { atomicExch(&sink.status[3], 1u); dst = nullptr; }
            else { sink.ovf_key[pos] = ((u64)(sink.base + r) << 16) | (u64)order; dst = (uint4*)(sink.ovf + pos); }
        }
        if (dst) { FinishSink fs{dst}; emit_aln(fs, r, order, res, e); }
    }
    if (real && q == 0) sink.nrec[(size_t)sink.base + r] = (uint8_t)(cnt > 255 ? 255 : cnt);
    return emit ? 1u : 0u;
}
template <int WPE>
__global__ __launch_bounds__(256, WPE) void thj_k_finish(Genome g, Params p, DevSpanBatch b, RecSink sink, ChainLists L) {
    __shared__ unsigned int s_rec;
    if (threadIdx.x == 0) s_rec = 0;
    __syncthreads();
    unsigned int acc = 0;
    if ((int)blockIdx.x < L.G) {
        const int blk = (int)blockIdx.x;
        unsigned int c[SPAN_LEAN_CLASSES];
        const unsigned int total = chain_block_counts(L, blk, c);
        for (unsigned int i0 = 0; i0 < total; i0 += 256) {       // (uniform trip count: the ballots want whole waves)
            const unsigned int i = i0 + threadIdx.x;
            acc += finish_one(g, p, b, sink, L, i < total, (u64)blk * (uint32_t)L.chunk + i);
        }
    } else {
        const unsigned int j = blockIdx.x - (unsigned int)L.G, n2 = chain_dense_count(L);
        const u64 jbase = (u64)L.G * (uint32_t)L.chunk;
        for (unsigned int i0 = j * 256u; i0 < n2; i0 += (unsigned int)L.G2 * 256u) {
            const unsigned int i = i0 + threadIdx.x;
            acc += finish_one(g, p, b, sink, L, i < n2, jbase + i);
        }
    }
    if (acc) atomicAdd(&s_rec, acc);
    __syncthreads();
    if (threadIdx.x == 0 && s_rec) atomicAdd(sink.total, (unsigned long long)s_rec);
}

// The chains of the multihit reads that need no search (chains_discover, thj_span_core.h): a thread per entry of the multihit list,
// workgroup b over tier 0's workgroup b's slice, the read's hit heads (with each hit's right end in place of its first cigar op) in
// the thread's own column of LDS.  A read with k chains becomes a group of k chain entries in rank order, padded to 1 / 2 / 4 / 8 and
// laid out at a multiple of that size in the dense list (a round's groups by size, its room reserved with one atomic per workgroup
// and round); a read that is declined goes on to thj_k_stitch_pack's list.
static constexpr int CH_TPB = 128;
struct LdsHitTab {
    const Q16* col;
    __device__ __forceinline__ SpanHitHead head(int j) const { const Q16 v = col[j * CH_TPB]; return SpanHitHead{v.x, (int32_t)v.y, v.z, v.w}; }
};
__global__ __launch_bounds__(CH_TPB) void thj_k_chains(Params p, DevSpanBatch b, Tiers t, ChainEntry* ent2, unsigned int* n2, unsigned int cap2, uint32_t* wl_pack, unsigned int* blk_pack, int G) {
    __shared__ Q16 s_head[CHAINS_MAXHITS * CH_TPB];
    __shared__ unsigned int s_base;
    typedef hipcub::BlockScan<u64, CH_TPB> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const int tid = (int)threadIdx.x;
    for (int blk = blockIdx.x; blk < G; blk += gridDim.x) {
        const unsigned int n_b = t.blk_multi[blk];
        unsigned int pack_n = 0;
        for (unsigned int j0 = 0; j0 < n_b; j0 += CH_TPB) {
            const unsigned int j = j0 + (unsigned int)tid;
            const bool has = j < n_b;
            const uint32_t r = has ? t.wl_multi[(u64)blk * (uint32_t)t.chunk + j] : 0u;
            int k = 0, nsegs = 0;
            uint32_t sel[CHAINS_MAX]; int q[CHAINS_MAX];
#pragma unroll
            for (int c = 0; c < CHAINS_MAX; ++c) { sel[c] = 0; q[c] = 0; }
            uint32_t sof0 = 0;
            if (has) {
                const uint32_t* so = b.seg_off + (u64)r * (uint32_t)b.nseg;
                uint32_t sof[CHAIN_MAXSEG + 1];
#pragma unroll
                for (int s = 0; s <= CHAIN_MAXSEG; ++s) sof[s] = s <= b.nseg ? so[s <= b.nseg ? s : 0] : 0u;
                {
                    bool open = true;
#pragma unroll
                    for (int s = 0; s < CHAIN_MAXSEG; ++s) { open = open && s < b.nseg && sof[s + 1] > sof[s]; nsegs += open ? 1 : 0; }
                }
                sof0 = sof[0];
                uint32_t last_so = sof[0], end_so = sof[0];
#pragma unroll
                for (int s = 1; s <= CHAIN_MAXSEG; ++s) { last_so = (s == nsegs - 1) ? sof[s] : last_so; end_so = (s == nsegs) ? sof[s] : end_so; }
                bool ok = nsegs > 0 && (load_head(b.hits, b.heads, (u64)last_so).z & SH_END) != 0;          // :2777-2785
                if (ok && p.bowtie2) {
#pragma unroll
                    for (int s = 0; s < CHAIN_MAXSEG; ++s) ok = ok && !(s < nsegs && (int)(sof[s + 1] - sof[s]) > p.max_seg_multihits);   // :2625-2632
                }
                if (ok) {                        // (else: nothing for this read, as in every tier)
                    const uint32_t nh = end_so - sof[0];
                    k = CHAINS_DECLINE;
                    if (nh <= (uint32_t)CHAINS_MAXHITS) {
                        // the heads eight at a time, all in flight together (one after the other behind pack_hit_right's branch they were
                        // sixteen round trips)
                        for (uint32_t h0 = 0; h0 < nh; h0 += 8) {
                            Q16 hq[8];
#pragma unroll
                            for (int x = 0; x < 8; ++x) hq[x] = load_head(b.hits, b.heads, (u64)sof[0] + (h0 + x < nh ? h0 + x : nh - 1));
#pragma unroll
                            for (int x = 0; x < 8; ++x)
                                if (h0 + x < nh) s_head[(h0 + x) * CH_TPB + tid] = Q16{hq[x].x, hq[x].y, hq[x].z, (uint32_t)pack_hit_right(hq[x], b.hits, (u64)sof[0] + h0 + x)};
                        }
                        uint32_t off[CHAIN_MAXSEG + 1];
#pragma unroll
                        for (int s = 0; s <= CHAIN_MAXSEG; ++s) off[s] = sof[s] - sof[0];
                        const LdsHitTab tab{s_head + tid};
                        k = chains_discover(p, tab, off, nsegs, sel, q);
                    }
                }
            }
            const int kp = k > 0 ? chains_padded(k) : 0;
            const bool declined = has && k == CHAINS_DECLINE;
            const u64 mine = (kp == 8 ? 1ull : 0ull) | (kp == 4 ? 1ull << 8 : 0ull) | (kp == 2 ? 1ull << 16 : 0ull) | (kp == 1 ? 1ull << 24 : 0ull) | (declined ? 1ull << 32 : 0ull);
            u64 excl, tot;
            Scan(tmp).ExclusiveSum(mine, excl, tot);
            const unsigned int N8 = (unsigned int)(tot & 255u), N4 = (unsigned int)((tot >> 8) & 255u), N2 = (unsigned int)((tot >> 16) & 255u), N1 = (unsigned int)((tot >> 24) & 255u);
            const unsigned int ND = (unsigned int)((tot >> 32) & 255u);
            const unsigned int T = 8u * N8 + 4u * N4 + 2u * N2 + N1, T8 = (T + 7u) & ~7u;
            if (tid == 0) s_base = T8 ? atomicAdd(n2, T8) : 0u;
            __syncthreads();
            const unsigned int gb = s_base;
            const bool room = gb + T8 <= cap2 && gb + T8 >= gb;
            if (tid == 0 && room && T) atomicAdd(&t.counters[7], N8 + N4 + N2 + N1);
            if (room) {
                if (kp) {
                    const unsigned int e8 = (unsigned int)(excl & 255u), e4 = (unsigned int)((excl >> 8) & 255u), e2 = (unsigned int)((excl >> 16) & 255u), e1 = (unsigned int)((excl >> 24) & 255u);
                    const unsigned int at = gb + (kp == 8 ? 8u * e8 : kp == 4 ? 8u * N8 + 4u * e4 : kp == 2 ? 8u * N8 + 4u * N4 + 2u * e2 : 8u * N8 + 4u * N4 + 2u * N2 + e1);
                    const uint32_t rl = (uint32_t)b.read_len[r];
                    for (int c = 0; c < kp; ++c) {
                        Q16* dst = (Q16*)(ent2 + at + (unsigned int)c);
                        if (c >= k) { dst[0] = Q16{JOINED_PAD, 0u, 0u, 0u}; continue; }
                        uint32_t sl = 0;                   // the chain of rank c
#pragma unroll
                        for (int d = 0; d < CHAINS_MAX; ++d) sl = (d < k && q[d] == c) ? sel[d] : sl;
                        uint32_t hx[CHAIN_MAXSEG];
#pragma unroll
                        for (int s = 0; s < CHAIN_MAXSEG; ++s) hx[s] = sof0 + ((sl >> (4 * (s < nsegs ? s : 0))) & 15u);
                        dst[0] = Q16{r, chain_meta(nsegs, c, k, (int)rl), 0u, 0u};
                        dst[1] = Q16{hx[0], hx[1], hx[2], hx[3]};
                    }
                }
                if ((unsigned int)tid < T8 - T) ((Q16*)(ent2 + gb + T + (unsigned int)tid))[0] = Q16{JOINED_PAD, 0u, 0u, 0u};
                if (declined) wl_pack[(u64)blk * (uint32_t)t.chunk + pack_n + (unsigned int)((excl >> 32) & 255u)] = r;
                pack_n += ND;
            } else {
                // the dense list is full: what lies below its end is padding, and the round's reads all go on to the packed tier
                for (unsigned int x = gb + (unsigned int)tid; x < cap2 && x - gb < T8; x += CH_TPB) ((Q16*)(ent2 + x))[0] = Q16{JOINED_PAD, 0u, 0u, 0u};
                const unsigned int groups_before = (unsigned int)(excl & 255u) + (unsigned int)((excl >> 8) & 255u) + (unsigned int)((excl >> 16) & 255u) + (unsigned int)((excl >> 24) & 255u);
                if (declined || kp) wl_pack[(u64)blk * (uint32_t)t.chunk + pack_n + (unsigned int)((excl >> 32) & 255u) + groups_before] = r;
                pack_n += ND + N8 + N4 + N2 + N1;
            }
            __syncthreads();
        }
        if (tid == 0) blk_pack[blk] = pack_n;
    }
}

// Tier 2, packed: the multihit list, 64 entries per wave at a time, as chains over the lanes (span_pack_wave): a read with its
// segments in c copies of a repeat is c one-hit-per-segment chains, each joined on registers the way tier 1 joins its reads.  What it
// does not take (reads with more than 64 first-segment hits, 256 hits or 64 chains, joins of more than LEAN_C cigar ops) goes on to
// thj_k_stitch_generic.  The waves draw their batches from one counter: a draw of 40-copy reads is several rounds, a draw of
// two-copy reads one.
struct WaveOps {
    int lane;
    __device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
    __device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }
    __device__ __forceinline__ uint32_t shfl(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)v); }
    __device__ __forceinline__ uint32_t incl_scan(uint32_t v) {
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);
        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);
        return v;
    }
    __device__ __forceinline__ uint32_t wmax(uint32_t v) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t o = shfl(v, lane ^ d); v = o > v ? o : v; }
        return v;
    }
    __device__ __forceinline__ void wsync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ __forceinline__ unsigned long long clock() { return wall_clock64(); }
};
static constexpr int PACK_MAXROOTS = 64, PACK_MAXHITS = 256, PACK_CHAINLIST = 128;
// list entries a wave draws at a time: the launch ends one draw's time after its last wave found the list empty.  32 while the list
// held every multihit read (a draw of two-copy reads is one round); since thj_k_chains takes those, what is left are the reads of
// five and more copies -- a draw of 32 could be ten rounds, and the slowest wave ran 570 us against a mean of 250 (THJ_PACK_TIMING).
// THJ_PACK_DRAW: developer switch
static const int PACK_DRAW_DEFAULT = 8;
static constexpr int PACK_HEAVY_HITS = 48;      // a read with more hits than this is drawn in the first pass (twelve copies of a four-segment read and up)
// Eight waves per workgroup, two workgroups per CU: 16 waves per CU is what 128 VGPRs allow, and a wave's 9 KB of LDS with the
// workgroup's slice table fit the CU's 160 KB twice over that way (four workgroups of four waves do not: three were resident).
static constexpr int PACK_TPB = 512;
template <int MS, int PACK_TPB = 512, int PACK_WPE = 4>
__global__ __launch_bounds__(PACK_TPB, PACK_WPE) void thj_k_stitch_pack(Genome g, Params p, SpanSets S, DevSpanBatch b, RecSink sink, Tiers t, int G, unsigned long long* dbg, int draw, int heavy_draw) {
    __shared__ unsigned int s_off[MAX_SLICES + 1];
    __shared__ unsigned int s_rec;
    __shared__ PackLds<MS, PACK_MAXHITS, PACK_CHAINLIST> s_pack[PACK_TPB / 64];
    if (threadIdx.x == 0) s_rec = 0;
    const int wave = (int)(threadIdx.x >> 6);
    WaveOps x{(int)(threadIdx.x & 63)};
    const unsigned int total = slice_offsets<PACK_TPB>(t.blk_multi, G, s_off);
    unsigned long long tmk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_start = dbg ? wall_clock64() : 0ull;
    // heavy_draw > 0: the list is walked twice -- first the reads with many hits (heavy_draw entries a draw, the others passed over), then
    // the rest.  The launch ends when its last wave does, and a wave that draws a batch of 30-copy reads last is the tail (mean 335 us a
    // wave, slowest 520: THJ_PACK_TIMING); the long batches go out first.
    for (int pass = heavy_draw > 0 ? 0 : 1; pass < 2; ++pass) {
        const int dr = pass == 0 ? heavy_draw : draw;
        for (;;) {
            unsigned int i0 = 0;
            if (x.lane == 0) i0 = atomicAdd(&t.counters[pass == 0 ? 8 : 3], (unsigned int)dr);
            i0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)i0);
            if (i0 >= total) break;
            const unsigned int i = i0 + (unsigned int)x.lane;
            bool has = x.lane < dr && i < total;
            const int sl = has ? slice_of(s_off, G, i) : 0;
            const uint32_t r = has ? t.wl_multi[(int64_t)sl * t.chunk + (i - s_off[sl])] : 0u;
            if (heavy_draw > 0) {
                uint32_t nh = 0;
                if (has) { const uint32_t* so = b.seg_off + (u64)r * (uint32_t)b.nseg; nh = so[b.nseg] - so[0]; }
                has = has && (nh > (uint32_t)PACK_HEAVY_HITS) == (pass == 0);
                if (x.ballot(has) == 0ull) continue;
            }
            const bool fwd = span_pack_wave<MS, PACK_MAXROOTS, PACK_MAXHITS, PACK_CHAINLIST>(x, g, p, S, b.hits, b.heads, b.seg_off, b.nseg, b.planes, b.W, b.read_len,
                                                                                             b.quals, b.qual_stride, r, has, s_pack[wave], sink, dbg ? tmk : nullptr);
            if (fwd) {
                t.wl_gen[(int64_t)sl * t.chunk + atomicAdd(&t.blk_gen[sl], 1u)] = r;
                atomicAdd(&t.counters[2], 1u);
            }
            x.wsync();
        }
    }
    if (dbg && x.lane == 0) {             // THJ_PACK_TIMING: the phases' ticks (10 ns) summed over the waves, [8] a wave's whole time: sum, [9] max
        for (int k = 0; k < 8; ++k) atomicAdd(&dbg[k], tmk[k]);
        const unsigned long long dt = wall_clock64() - t_start;
        atomicAdd(&dbg[8], dt); atomicMax(&dbg[9], dt); atomicAdd(&dbg[10], 1ull);
    }
    if (sink.acc) atomicAdd(&s_rec, (unsigned int)sink.acc);
    __syncthreads();
    if (threadIdx.x == 0 && s_rec) atomicAdd(sink.total, (unsigned long long)s_rec);
}

static constexpr int GENERIC_MAXJOIN = 48;     // joined alignments of a read on tier 3's lean pass (40 copies of a repeat and some)
__global__ __launch_bounds__(128) void thj_k_stitch_generic(Genome g, Params p, SpanSets S, DevSpanBatch b, RecSink sink, Tiers t, int G) {
    extern __shared__ uint4 lds_stage[];          // nseg hits per thread: the chain under construction
    __shared__ unsigned int s_off[MAX_SLICES + 1];
    __shared__ unsigned int s_rec;
    if (threadIdx.x == 0) s_rec = 0;
    SpanHit* stage = (SpanHit*)lds_stage + (size_t)threadIdx.x * b.nseg;
    const unsigned int total = slice_offsets<128>(t.blk_gen, G, s_off);
    for (unsigned int i = blockIdx.x * 128 + threadIdx.x; i < total; i += gridDim.x * 128) {
        const int sl = slice_of(s_off, G, i);
        const int r = (int)t.wl_gen[(int64_t)sl * t.chunk + (i - s_off[sl])];
        const uint32_t* so = b.seg_off + (size_t)r * b.nseg;
        const u64* rp = b.planes + (size_t)r * 3 * b.W;
        const uint8_t* q = b.quals + (size_t)r * b.qual_stride;
        int st = span_read_multi<GENERIC_MAXJOIN>(g, p, S, b.hits, so, b.nseg, rp, b.W, (int)b.read_len[r], q, (uint32_t)r, stage, sink);
        if (st == SPAN_NEED_GENERIC) st = span_read(g, p, S, b.hits, so, b.nseg, rp, b.W, (int)b.read_len[r], q, (uint32_t)r, sink);
        if (st == SPAN_TOO_MANY_JOINED && defer_huge(t, (uint32_t)r)) continue;
        sink.done((uint32_t)r);
        if (st) atomicAdd(&sink.status[st], 1u);
    }
    if (sink.acc) atomicAdd(&s_rec, (unsigned int)sink.acc);
    __syncthreads();
    if (threadIdx.x == 0 && s_rec) atomicAdd(sink.total, (unsigned long long)s_rec);
}

// --fusion-search: every read tier 0 did not finish (the multihit list holds them all), through the fusion branches of
// dfs_seg_hits / merge_chain on general per-thread arrays (thj_span_fusion.h).
__global__ __launch_bounds__(64) void thj_k_stitch_fusion(Genome g, Params p, SpanSets S, FusionSet F, DevSpanBatch b, RecSink sink, Tiers t, int G) {
    __shared__ unsigned int s_off[MAX_SLICES + 1];
    __shared__ unsigned int s_rec;
    if (threadIdx.x == 0) s_rec = 0;
    const unsigned int total = slice_offsets<64>(t.blk_multi, G, s_off);
    for (unsigned int i = blockIdx.x * 64 + threadIdx.x; i < total; i += gridDim.x * 64) {
        const int sl = slice_of(s_off, G, i);
        const int r = (int)t.wl_multi[(int64_t)sl * t.chunk + (i - s_off[sl])];
        if (fusion_read_heavy(b.seg_off + (size_t)r * b.nseg, b.nseg) && defer_huge(t, (uint32_t)r)) continue;
        int st = span_read_fusion(g, p, S, F, b.hits, b.seg_off + (size_t)r * b.nseg, b.nseg, b.planes + (size_t)r * 3 * b.W, b.W,
                                  (int)b.read_len[r], b.quals + (size_t)r * b.qual_stride, (uint32_t)r, sink);
        if (st == SPAN_TOO_MANY_JOINED && defer_huge(t, (uint32_t)r)) continue;
        sink.done((uint32_t)r);
        if (st) atomicAdd(&sink.status[st], 1u);
    }
    if (sink.acc) atomicAdd(&s_rec, (unsigned int)sink.acc);
    __syncthreads();
    if (threadIdx.x == 0 && s_rec) atomicAdd(sink.total, (unsigned long long)s_rec);
}

// The reads the generic / fusion kernels listed: one at a time per workgroup (lane 0), joined alignments in the workgroup's slice
// of the context's big workspace (2 * cap records: the list and the merge sort's scratch).  Rare by construction.
static constexpr int HUGE_BLOCKS = 256, HUGE_BLOCKS_MAX = 1024, HUGE_CAP = 8192, HUGE_LIST_CAP = 1 << 18;
static constexpr int HUGE_REC_BYTES = 128;               // >= sizeof(Aln), sizeof(FHit)
static constexpr size_t HUGE_BLOCK_BYTES = (size_t)2 * HUGE_CAP * HUGE_REC_BYTES;      // a workgroup's slice (2 MB); a scratch set's workspace is c->huge_blocks of them
static_assert(sizeof(Aln) <= HUGE_REC_BYTES && sizeof(FHit) <= HUGE_REC_BYTES, "workspace record size");
struct FusWaveDev {
    int lane;
    unsigned int* tm; unsigned long long t_last;       // THJ_HUGE_TIMERS (developer): [k] += 10 ns ticks of phase k (search, reorder, sort, records), [4] = longest list, [5] = reads
    __device__ __forceinline__ void mark(int k, int nj) {
        if (!tm || lane != 0) return;
        const unsigned long long now = wall_clock64();
        if (k >= 0) atomicAdd(&tm[k], (unsigned int)(now - t_last));
        if (k == 0) { atomicMax(&tm[4], (unsigned int)nj); atomicAdd(&tm[5], 1u); }
        t_last = now;
    }
    __device__ __forceinline__ void counts(uint32_t leaves, uint32_t tests) { if (tm) { atomicAdd(&tm[6], leaves); atomicAdd(&tm[7], tests); } }
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ uint32_t atomic_add(uint32_t* q, uint32_t v) { return atomicAdd(q, v); }
    __device__ __forceinline__ unsigned long long ballot(bool q) { return __ballot(q); }
};
static_assert(fus_wave_ws_bytes(HUGE_CAP) <= (size_t)2 * HUGE_CAP * HUGE_REC_BYTES, "a workgroup's slice of the workspace holds fusion_read_wave's arrays");
__global__ __launch_bounds__(64) void thj_k_stitch_huge(Genome g, Params p, SpanSets S, FusionSet F, DevSpanBatch b, RecSink sink, Tiers t, char* ws, int cap, int by_wave) {
    const unsigned int n = *t.huge_cnt < (unsigned int)t.huge_list_cap ? *t.huge_cnt : (unsigned int)t.huge_list_cap;
    char* mine = ws + (size_t)blockIdx.x * 2 * (size_t)cap * HUGE_REC_BYTES;
    if (p.fusion_search && by_wave) {                                // the wave on a read (fusion_read_wave)
        __shared__ FusWaveShared sh;
        FusWaveDev x{(int)threadIdx.x, by_wave == 2 ? sink.status + 8 : nullptr, 0ull};
        for (unsigned int i = blockIdx.x; i < n; i += gridDim.x) {
            const int r = (int)t.huge_list[i];
            int n_rec = 0;
            const int st = fusion_read_wave(x, g, p, S, F, b.hits, b.seg_off + (size_t)r * b.nseg, b.nseg, b.planes + (size_t)r * 3 * b.W, b.W, (int)b.read_len[r],
                                            b.quals + (size_t)r * b.qual_stride, (uint32_t)r, sink, mine, cap, sh, n_rec);
            sink.emitted = 0;
            if (threadIdx.x == 0) {
                sink.set_count((uint32_t)r, n_rec);
                if (st) atomicAdd(&sink.status[st], 1u);
            }
        }
        if (sink.acc) atomicAdd(sink.total, (unsigned long long)sink.acc);
        return;
    }
    if (threadIdx.x != 0) return;
    for (unsigned int i = blockIdx.x; i < n; i += gridDim.x) {
        const int r = (int)t.huge_list[i];
        const uint32_t* so = b.seg_off + (size_t)r * b.nseg;
        const u64* rp = b.planes + (size_t)r * 3 * b.W;
        const uint8_t* q = b.quals + (size_t)r * b.qual_stride;
        int st;
        if (p.fusion_search) st = span_read_fusion(g, p, S, F, b.hits, so, b.nseg, rp, b.W, (int)b.read_len[r], q, (uint32_t)r, sink, (FHit*)mine, cap);
        else st = span_read(g, p, S, b.hits, so, b.nseg, rp, b.W, (int)b.read_len[r], q, (uint32_t)r, sink, (Aln*)mine, cap);
        sink.done((uint32_t)r);
        if (st) atomicAdd(&sink.status[st], 1u);
    }
    if (sink.acc) atomicAdd(sink.total, (unsigned long long)sink.acc);
}

__global__ __launch_bounds__(256) void thj_k_ins_split(const u64* keys, const u64* vals, int64_t n, u64* okeys, uint32_t* oseq) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        okeys[i] = keys[i];
        oseq[i] = (uint32_t)(vals[i] & ((1u << INS_SEQ_BITS) - 1u));
    }
}

// junc_bucket[b] = first key whose left position is >= b << JUNC_BUCKET_SHIFT (see junc_range)
__global__ __launch_bounds__(256) void thj_k_junc_buckets(const u64* keys, int64_t n, uint32_t* bucket, int64_t n_buckets) {
    for (int64_t bk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; bk <= n_buckets; bk += (int64_t)gridDim.x * blockDim.x)
        bucket[bk] = bk == n_buckets ? (uint32_t)n : (uint32_t)lower_bound_u64(keys, n, (u64)bk << (JUNC_BUCKET_SHIFT + 30));
}

static int build_junc_buckets(thj_ctx* c) {
    const int64_t nb = ((c->n_blocks * 64 + 2) >> JUNC_BUCKET_SHIFT) + 1;
    if (c->n_span_junc >= (1ll << 32)) { thj_set_error("more than 2^32 junctions"); return THJ_EINVAL; }
    if (nb != c->n_junc_buckets || !c->d_junc_bucket) {
        hipFree(c->d_junc_bucket); c->d_junc_bucket = nullptr;
        HIPCHK(hipMalloc(&c->d_junc_bucket, (size_t)(nb + 1) * 4));
        c->n_junc_buckets = nb;
    }
    int64_t blocks = (nb + 256) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(thj_k_junc_buckets, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const u64*)c->d_span_junc, c->n_span_junc,
                       c->d_junc_bucket, nb);
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

static void jb_free(thj_ctx* c);
void thj_span_free(thj_ctx* c) {
    jb_free(c);
    hipFree(c->d_span_junc); hipFree(c->d_span_cat); hipFree(c->d_span_ins_key); hipFree(c->d_span_ins_seq); hipFree(c->d_junc_bucket); hipFree(c->d_span_fus); hipFree(c->d_huge_ws); hipFree(c->d_huge_list);
    hipFree(c->d_aln_pool); hipFree(c->d_aln_sorted); hipFree(c->d_aln_keys); hipFree(c->d_nrec);
    hipFree(c->d_aln_count); hipFree(c->d_span_status);
    for (auto& ss : c->span_set) { hipFree(ss.d_worklist); hipFree(ss.d_ent); hipFree(ss.d_joined); hipFree(ss.d_defer); }
    if (c->span_stream_own) for (auto& st : c->span_stream) if (st) hipStreamDestroy(st);
    for (auto& e : c->span_ev) if (e) hipEventDestroy(e);
    for (auto& pr : c->span_prof_events) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
}

static int ensure_span_state(thj_ctx* c) {
    if (!c->d_aln_count) {
        HIPCHK(hipMalloc(&c->d_aln_count, 16));
        HIPCHK(hipMalloc(&c->d_span_status, 64 * sizeof(unsigned int)));      // [0..3] statuses of the pass, [16 + 16 k ..] the counters of scratch set k (span_launch)
        HIPCHK(hipMemsetAsync(c->d_aln_count, 0, 16, c->stream));
        HIPCHK(hipMemsetAsync(c->d_span_status, 0, 256, c->stream));
    }
    return THJ_OK;
}

static int ensure_sets_cap(thj_ctx* c, int64_t nj, int64_t ni) {
    if (nj + 1 > c->cap_span_junc) {
        hipFree(c->d_span_junc); hipFree(c->d_span_cat); c->d_span_junc = c->d_span_cat = nullptr;
        c->cap_span_junc = nj + nj / 4 + 1024;
        HIPCHK(hipMalloc(&c->d_span_junc, (size_t)c->cap_span_junc * 8));
        HIPCHK(hipMalloc(&c->d_span_cat, (size_t)c->cap_span_junc * 8));
    }
    if (ni + 1 > c->cap_span_ins) {
        hipFree(c->d_span_ins_key); hipFree(c->d_span_ins_seq); c->d_span_ins_key = nullptr; c->d_span_ins_seq = nullptr;
        c->cap_span_ins = ni + ni / 4 + 1024;
        HIPCHK(hipMalloc(&c->d_span_ins_key, (size_t)c->cap_span_ins * 8));
        HIPCHK(hipMalloc(&c->d_span_ins_seq, (size_t)c->cap_span_ins * 4));
    }
    return THJ_OK;
}

static u64 host_junc_key(const thj_ctx* c, uint32_t ref, uint32_t left, uint32_t right, bool anti) {
    Genome g{nullptr, c->h_contig_blk.data(), nullptr, c->n_contigs};
    return junc_key(g, ref, left, right, anti);
}

extern "C" int thj_span_sets_upload(thj_ctx* c, const thj_junction* juncs, int64_t n_juncs, const uint32_t* ins, int64_t n_ins) {
    if (!c || (n_juncs > 0 && !juncs) || (n_ins > 0 && !ins)) { thj_set_error("thj_span_sets_upload: bad argument"); return THJ_EINVAL; }
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    HIPCHK(hipSetDevice(c->device));
    std::vector<u64> jk((size_t)n_juncs), ik((size_t)n_ins);
    std::vector<uint32_t> iseq((size_t)n_ins);
    Genome g{nullptr, c->h_contig_blk.data(), nullptr, c->n_contigs};
    for (int64_t i = 0; i < n_juncs; ++i) {
        const thj_junction& j = juncs[i];
        if (j.ref_id == 0 || (int32_t)j.ref_id > c->n_contigs) { thj_set_error("junction %lld: ref_id %u out of range", (long long)i, j.ref_id); return THJ_EINVAL; }
        if ((uint32_t)(j.right - j.left) >= (1u << 29)) { thj_set_error("junction %lld longer than 2^29", (long long)i); return THJ_EINVAL; }
        jk[i] = junc_key(g, j.ref_id, j.left, j.right, j.antisense != 0);
        if (i && jk[i] <= jk[i - 1]) { thj_set_error("junctions must be sorted unique in Junction::operator< order (entry %lld)", (long long)i); return THJ_EINVAL; }
    }
    for (int64_t i = 0; i < n_ins; ++i) {
        uint32_t ref = ins[4 * i], left = ins[4 * i + 1], len = ins[4 * i + 2];
        if (ref == 0 || (int32_t)ref > c->n_contigs || len == 0 || len > 6) { thj_set_error("insertion %lld invalid (length must be 1..6)", (long long)i); return THJ_EINVAL; }
        ik[i] = ins_key(g, ref, left, (int)len);
        iseq[i] = ins[4 * i + 3];
        if (i && ik[i] <= ik[i - 1]) { thj_set_error("insertions must be sorted unique by (ref,left,len) (entry %lld)", (long long)i); return THJ_EINVAL; }
    }
    int rc = ensure_sets_cap(c, n_juncs, n_ins);
    if (rc) return rc;
    if (n_juncs) HIPCHK(hipMemcpyAsync(c->d_span_junc, jk.data(), (size_t)n_juncs * 8, hipMemcpyHostToDevice, c->stream));
    if (n_ins) {
        HIPCHK(hipMemcpyAsync(c->d_span_ins_key, ik.data(), (size_t)n_ins * 8, hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->d_span_ins_seq, iseq.data(), (size_t)n_ins * 4, hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_span_junc = n_juncs; c->n_span_ins = n_ins;
    return build_junc_buckets(c);
}

extern "C" int thj_span_sets_from_segjuncs(thj_ctx* c) {
    // junction set := sorted union of the context's junction and deletion keys (what
    // long_spanning_reads.cpp:2897-2944 builds from the .juncs and .deletions files);
    // insertion set := the sorted insertion keys + their sequences.  Device to device.
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    const int64_t nj = c->n_junc + c->n_del, ni = c->n_ins;
    int rc = ensure_sets_cap(c, nj, ni);
    if (rc) return rc;
    if (nj) {
        // the two sorted lists merged (a sort of their concatenation was eight more radix passes over what thj_segjuncs_finish had just sorted).
        // Merged, NOT made unique: a deletion and a '+' junction with the same ends give the same key twice, and
        // a repeated key is harmless to its only consumer -- closure_search skips a candidate that does not improve
        // on the best one (`diff >= best_diff`), which an identical twin never does.
        if (c->n_junc == 0 || c->n_del == 0)
            HIPCHK(hipMemcpyAsync(c->d_span_junc, c->n_junc ? c->d_junc_sorted : c->d_del_sorted, (size_t)nj * 8, hipMemcpyDeviceToDevice, c->stream));
        else {
            size_t need = 0;
            HIPCHK(rocprim::merge(nullptr, need, (const u64*)c->d_junc_sorted, (const u64*)c->d_del_sorted, c->d_span_junc, (size_t)c->n_junc, (size_t)c->n_del,
                                  rocprim::less<u64>(), c->stream));
            if (need > c->sort_tmp_bytes) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; c->sort_tmp_bytes = 0; HIPCHK(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
            size_t tmp = c->sort_tmp_bytes;
            HIPCHK(rocprim::merge(c->d_sort_tmp, tmp, (const u64*)c->d_junc_sorted, (const u64*)c->d_del_sorted, c->d_span_junc, (size_t)c->n_junc, (size_t)c->n_del,
                                  rocprim::less<u64>(), c->stream));
        }
        c->n_span_junc = nj;
    } else c->n_span_junc = 0;
    if (ni) {
        int64_t blocks = (ni + 255) / 256; if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(thj_k_ins_split, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const u64*)c->d_ins_key_sorted,
                           (const u64*)c->d_ins_val_sorted, ni, c->d_span_ins_key, c->d_span_ins_seq);
    }
    c->n_span_ins = ni;
    return build_junc_buckets(c);          // stream-ordered: thj_span_run_async on this context needs no synchronisation
}


// the dense head array of a batch: the first 16 bytes of every 32-byte hit record
__global__ void thj_k_hit_heads(const Q16* hits, int64_t n, Q16* heads) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) heads[i] = hits[2 * i];
}
extern "C" int thj_span_hit_heads_async(thj_ctx* c, const thj_span_hit* d_hits, int64_t n_hits, void* d_heads) {
    if (!c || (n_hits > 0 && (!d_hits || !d_heads))) { thj_set_error("thj_span_hit_heads_async: null argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (n_hits <= 0) return THJ_OK;
    int64_t blocks = (n_hits + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(thj_k_hit_heads, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const Q16*)d_hits, n_hits, (Q16*)d_heads);
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

extern "C" int thj_span_batch_upload(thj_ctx* c, const thj_span_batch* h, int64_t n_hits, thj_span_batch** out) {
    if (!c || !h || !out) { thj_set_error("thj_span_batch_upload: null argument"); return THJ_EINVAL; }
    if (h->n_reads < 0 || h->nseg < 1 || h->nseg > SPAN_MAXSEG || h->words_per_plane < 1 || h->words_per_plane > 8 || h->qual_stride < 0) {
        thj_set_error("thj_span_batch_upload: bad shape (nseg 1..16, words_per_plane 1..8)"); return THJ_EINVAL;
    }
    HIPCHK(hipSetDevice(c->device));
    OwnedSpanBatch* ob = new OwnedSpanBatch();
    memset(ob, 0, sizeof *ob);
    ob->desc = *h;
    const int64_t n = h->n_reads;
    const size_t sizes[5] = {(size_t)(n * h->nseg + 1) * 4, (size_t)n_hits * 32, (size_t)n * 3 * h->words_per_plane * 8,
                             (size_t)n * 2, (size_t)n * h->qual_stride};
    const void* src[5] = {h->seg_off, h->hits, h->read_planes, h->read_len, h->quals};
    for (int i = 0; i < 5; ++i) {
        { int rc_ = thj_dev_alloc(c, &ob->ptrs[i], sizes[i] ? sizes[i] : 16); if (rc_) return rc_; }
        if (sizes[i]) HIPCHK(hipMemcpyAsync(ob->ptrs[i], src[i], sizes[i], hipMemcpyHostToDevice, c->stream));
    }
    ob->desc.seg_off = (const uint32_t*)ob->ptrs[0];
    ob->desc.hits = (const thj_span_hit*)ob->ptrs[1];
    ob->desc.read_planes = (const uint64_t*)ob->ptrs[2];
    ob->desc.read_len = (const uint16_t*)ob->ptrs[3];
    ob->desc.quals = (const uint8_t*)ob->ptrs[4];
    // Every batch carries the dense head array (the first 16 bytes of each record: contig, position, flags, first cigar op --
    // all tier 0 reads of a hit), so tier 0 streams 16 B per hit.  For an uploaded batch it is derived here, once, when the
    // batch is made; the device-side ingest writes it while it scatters the records.
    { int rc_ = thj_dev_alloc(c, &ob->ptrs[5], (size_t)(n_hits ? n_hits : 1) * 16); if (rc_) return rc_; }
    if (n_hits > 0) {
        int64_t blocks = (n_hits + 255) / 256; if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(thj_k_hit_heads, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const Q16*)ob->ptrs[1], n_hits, (Q16*)ob->ptrs[5]);
    }
    ob->desc.hit_heads = ob->ptrs[5];
    HIPCHK(hipStreamSynchronize(c->stream));
    *out = &ob->desc;
    return THJ_OK;
}

extern "C" int thj_span_batch_free(thj_ctx* c, thj_span_batch* dev) {
    if (!c || !dev) return THJ_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    OwnedSpanBatch* ob = (OwnedSpanBatch*)dev;
    for (int i = 0; i < 8; ++i) thj_dev_release(c, ob->ptrs[i]);
    delete ob;
    return THJ_OK;
}

extern "C" int thj_span_reset_async(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_span_state(c);
    if (rc) return rc;
    if (c->span_t0_pending && c->span_stream[0] && c->span_stream[1] && c->span_ev[3]) {
        // a pair's tier 0 went out on the side streams (thj_span_tier0_pair_async) and nothing joined it: its kernels still add to the
        // record count and write slots -- the memsets below wait for them
        HIPCHK(hipEventRecord(c->span_ev[3], c->span_stream[0])); HIPCHK(hipStreamWaitEvent(c->stream, c->span_ev[3], 0));
        HIPCHK(hipEventRecord(c->span_ev[4], c->span_stream[1])); HIPCHK(hipStreamWaitEvent(c->stream, c->span_ev[4], 0));
    }
    HIPCHK(hipMemsetAsync(c->d_aln_count, 0, 16, c->stream));       // [0] total records, [1] overflow-pool records
    HIPCHK(hipMemsetAsync(c->d_span_status, 0, 32, c->stream));     // [0..3] statuses, [5] thj_k_join's "cannot happen"
    c->n_alns = 0;
    c->span_reads = 0;
    c->span_t0_pending = false;
    c->h_alns.clear();
    return THJ_OK;
}

// slots for `want` reads in this pass, preserving what earlier runs of the pass wrote
static int ensure_slots(thj_ctx* c, int64_t want) {
    if (want <= c->aln_cap) return THJ_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    int64_t ncap = want + want / 4 + 4096;
    void* ns = nullptr; uint8_t* nn = nullptr;
    HIPCHK(hipMalloc(&ns, (size_t)ncap * 128));
    HIPCHK(hipMalloc(&nn, (size_t)ncap));
    if (c->d_aln_pool && c->span_reads) {
        HIPCHK(hipMemcpy(ns, c->d_aln_pool, (size_t)c->span_reads * 128, hipMemcpyDeviceToDevice));
        HIPCHK(hipMemcpy(nn, c->d_nrec, (size_t)c->span_reads, hipMemcpyDeviceToDevice));
    }
    hipFree(c->d_aln_pool); hipFree(c->d_nrec);
    c->d_aln_pool = ns; c->d_nrec = nn; c->aln_cap = ncap;
    // overflow pool (2nd.. records of multihit reads): as many records again as there are slots -- never smaller than it already is
    // (thj_span_finish enlarges it on its own when a pass needed more), and nothing to keep when no pass is open
    {
        const int64_t ocap = ncap > c->ovf_cap ? ncap : c->ovf_cap;
        void* no = nullptr; u64* nk = nullptr;
        HIPCHK(hipMalloc(&no, (size_t)ocap * 128));
        HIPCHK(hipMalloc(&nk, (size_t)ocap * 8));
        if (c->d_aln_sorted && c->ovf_cap && c->span_reads) {
            HIPCHK(hipMemcpy(no, c->d_aln_sorted, (size_t)c->ovf_cap * 128, hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(nk, c->d_aln_keys, (size_t)c->ovf_cap * 8, hipMemcpyDeviceToDevice));
        }
        hipFree(c->d_aln_sorted); hipFree(c->d_aln_keys);
        c->d_aln_sorted = no; c->d_aln_keys = nk; c->ovf_cap = ocap;
    }
    return THJ_OK;
}

static int check_span_params(const thj_params* p, const thj_span_batch* b) {
    if (p->segment_length < 8 || p->segment_length > 64) { thj_set_error("segment_length %d unsupported by the stitch kernel (8..64)", p->segment_length); return THJ_EINVAL; }
    if (p->max_insertion_length < 0 || p->max_insertion_length > 6) { thj_set_error("max_insertion_length %d unsupported (0..6)", p->max_insertion_length); return THJ_EINVAL; }
    if (p->max_report_intron + 64 >= (1 << 29)) { thj_set_error("max_report_intron too large for the packed key"); return THJ_EINVAL; }
    if (b->nseg < 1 || b->nseg > SPAN_MAXSEG) { thj_set_error("nseg %d unsupported (1..16)", b->nseg); return THJ_EINVAL; }
    if (b->words_per_plane < 1 || b->words_per_plane > 8) { thj_set_error("words_per_plane %d unsupported (1..8: reads of up to 512 bases)", b->words_per_plane); return THJ_EINVAL; }
    if (p->fusion_search && (b->nseg > FUS_MAXSEG || b->words_per_plane > 4)) { thj_set_error("--fusion-search takes reads of at most eight segments and 256 bases"); return THJ_EINVAL; }
    if ((int64_t)b->n_reads >= (1ll << 31)) { thj_set_error("batch too large"); return THJ_EINVAL; }
    return THJ_OK;
}

extern "C" int thj_span_fusions_upload(thj_ctx* c, const thj_span_fusion* f, int64_t n) {
    if (!c || n < 0 || (n > 0 && !f)) { thj_set_error("thj_span_fusions_upload: bad argument"); return THJ_EINVAL; }
    static_assert(sizeof(thj_span_fusion) == sizeof(FusKey), "fusion key layout");
    for (int64_t i = 1; i < n; ++i) {
        const FusKey a{f[i - 1].ref_id1, f[i - 1].ref_id2, f[i - 1].left, f[i - 1].right, f[i - 1].dir}, b{f[i].ref_id1, f[i].ref_id2, f[i].left, f[i].right, f[i].dir};
        if (f_fus_cmp(a, b) >= 0) { thj_set_error("thj_span_fusions_upload: the list must be sorted unique in Fusion::operator< order"); return THJ_EINVAL; }
    }
    HIPCHK(hipSetDevice(c->device));
    if (c->cap_span_fus < n || !c->d_span_fus) {
        hipFree(c->d_span_fus); c->d_span_fus = nullptr; c->cap_span_fus = 0;
        const int64_t cap = n + n / 4 + 64;
        HIPCHK(hipMalloc(&c->d_span_fus, (size_t)cap * sizeof(FusKey)));
        c->cap_span_fus = cap;
    }
    if (n) HIPCHK(hipMemcpyAsync(c->d_span_fus, f, (size_t)n * sizeof(FusKey), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_span_fus = n;
    return THJ_OK;
}

// the fusion list of this context's own fusion search (thj_fusion_finish), handed over on the device the way
// thj_span_sets_from_segjuncs hands over the junctions: what the .fusions file carries between the two programs
__global__ __launch_bounds__(256) void thj_k_fus_to_keys(const thj_fusion* f, int64_t n, FusKey* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const thj_fusion e = f[i];
        out[i] = FusKey{e.ref_id1, e.ref_id2, e.left, e.right, e.dir};
    }
}
extern "C" int thj_span_fusions_from_segjuncs(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_fus_out) {
        // no device copy (reduced on the host, or merged across ranks): up from the host list
        std::vector<thj_span_fusion> f(c->h_fusions.size());
        for (size_t i = 0; i < f.size(); ++i) { const thj_fusion& e = c->h_fusions[i]; f[i] = thj_span_fusion{e.ref_id1, e.ref_id2, e.left, e.right, e.dir}; }
        return thj_span_fusions_upload(c, f.data(), (int64_t)f.size());
    }
    const int64_t n = c->n_fus_out;
    if (c->cap_span_fus < n || !c->d_span_fus) {
        HIPCHK(hipStreamSynchronize(c->stream));
        hipFree(c->d_span_fus); c->d_span_fus = nullptr; c->cap_span_fus = 0;
        const int64_t cap = n + n / 4 + 64;
        HIPCHK(hipMalloc(&c->d_span_fus, (size_t)cap * sizeof(FusKey)));
        c->cap_span_fus = cap;
    }
    if (n) {
        int64_t grid = (n + 255) / 256; if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(thj_k_fus_to_keys, dim3((unsigned)grid), dim3(256), 0, c->stream, (const thj_fusion*)c->d_fus_out, n, (FusKey*)c->d_span_fus);
        HIPCHK(hipGetLastError());
    }
    c->n_span_fus = n;
    return THJ_OK;
}


// ---- one batch through the tiers -----------------------------------------------------------------------------------------
// A batch runs on a "set" of scratch (worklists, chain entries, joined hits, counters: c->span_set[k]) and on two streams:
//   sm  thj_k_stitch_contig -> thj_k_stitch (the one-hit-per-segment reads that cannot travel as chain entries; it may add to the
//       multihit list) -> thj_k_stitch_pack -> thj_k_stitch_generic   [with --fusion-search: thj_k_stitch_fusion]
//   sa  (after tier 0)  thj_k_join -> thj_k_finish
// thj_span_run_async runs one batch on set 0 (sm = the context's stream); thj_span_run_pair_async runs two batches -- the two sides
// of a pass -- beside each other on both sets: the latency-bound kernels of one side overlap the bandwidth-bound tier 0 of the other.
// Everything is joined on the context's stream before either call returns.
static constexpr int SPAN_CNT_WORDS = 16;      // a set's counters: [0] reads to the closure kernels, [1] multihit, [2] general, [3] the packed tier's
                                               // place in its list, [4] reads for thj_k_stitch_huge, [5] entries in the dense list of the multihit reads' chains, [6] reads that
                                               // travel as chain entries from tier 0, [7] multihit reads that travel as groups of chain entries
enum { SPK_CONTIG = 0, SPK_CHAINS, SPK_JOIN, SPK_CLOSURE, SPK_FINISH, SPK_LEAN, SPK_PACK, SPK_GENERIC, SPK_N };

// the dense list of the multihit reads' chains (thj_k_chains): room for one entry per read of the batch -- three times what SURVEY 8(d)'s
// mix fills; a workgroup that finds it full hands its reads to the packed tier
static int64_t chain_cap2(int64_t n_reads) {
    if (const char* e = getenv("THJ_CHAIN_CAP")) { const int64_t v = atoll(e); if (v >= 8) return v / 8 * 8; }      // (tests: a list that fills up)
    return (n_reads + 7) / 8 * 8 + 2048;
}
static int64_t chain_g2(int64_t G) { return G >= 4 ? G / 4 : 1; }
static int64_t chain_slice2(int64_t n_reads, int64_t G) { const int64_t g2 = chain_g2(G); return (chain_cap2(n_reads) + g2 * 256 - 1) / (g2 * 256) * 256; }
static int ensure_span_set(thj_ctx* c, int set, int64_t n_reads, int64_t G, int64_t chunk, bool chains) {
    thj_ctx::SpanSet& ss = c->span_set[set];
    const int NC = SPAN_LEAN_CLASSES;
    const int64_t wl_need = (3 + NC) * G * chunk + (3 + 2 * NC) * MAX_SLICES;
    if (ss.worklist_cap < wl_need) {
        HIPCHK(hipDeviceSynchronize());
        hipFree(ss.d_worklist); ss.d_worklist = nullptr; ss.worklist_cap = 0;
        HIPCHK(hipMalloc(&ss.d_worklist, (size_t)wl_need * 4));
        ss.worklist_cap = wl_need;
    }
    if (chains) {
        const int64_t cap2 = chain_cap2(n_reads), g2 = chain_g2(G);
        const int64_t ent_need = NC * G * chunk + cap2, j_need = G * chunk + cap2, d_need = G * chunk + g2 * chain_slice2(n_reads, G) + G + g2 + 16;
        if (ss.ent_cap < ent_need) {
            HIPCHK(hipDeviceSynchronize());
            hipFree(ss.d_ent); ss.d_ent = nullptr; ss.ent_cap = 0;
            HIPCHK(hipMalloc(&ss.d_ent, (size_t)ent_need * sizeof(ChainEntry)));
            ss.ent_cap = ent_need;
        }
        if (ss.defer_cap < d_need) {
            HIPCHK(hipDeviceSynchronize());
            hipFree(ss.d_defer); ss.d_defer = nullptr; ss.defer_cap = 0;
            HIPCHK(hipMalloc(&ss.d_defer, (size_t)d_need * 4));
            ss.defer_cap = d_need;
        }
        if (ss.joined_cap < j_need) {
            HIPCHK(hipDeviceSynchronize());
            hipFree(ss.d_joined); ss.d_joined = nullptr; ss.joined_cap = 0;
            const int64_t cap = j_need + j_need / 8 + 1024;
            HIPCHK(hipMalloc(&ss.d_joined, (size_t)cap * 3 * sizeof(Q16)));
            ss.joined_cap = cap;
        }
    }
    return THJ_OK;
}
static int ensure_span_streams(thj_ctx* c, int need) {
    // the context's side streams (thj_ctx.h: shared with stage 1, which is over when this stage runs), as many as the call uses.
    // THJ_SPAN_PRIO=1: developer switch -- streams of this stage's own, the join / closure search / finish ones at the highest priority
    // (measured worse)
    static const bool prio = getenv("THJ_SPAN_PRIO") != nullptr;
    if (!c->span_ev[0]) for (int i = 0; i < 10; ++i) HIPCHK(hipEventCreateWithFlags(&c->span_ev[i], hipEventDisableTiming));
    if (prio) {
        if (c->span_stream[0]) return THJ_OK;
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        for (int i = 0; i < 3; ++i) HIPCHK(hipStreamCreateWithPriority(&c->span_stream[i], hipStreamNonBlocking, i != 1 ? hi : 0));
        c->span_stream_own = true;
        return THJ_OK;
    }
    const int rc = thj_ensure_aux_streams(c, need);
    if (rc) return rc;
    for (int i = 0; i < 3; ++i) c->span_stream[i] = c->aux_stream[i];
    return THJ_OK;
}

// the launches of one batch: `base` = its first slot in the pass; sm / sa as above (sa == sm: everything on one stream)
// st0: tier 0's stream (the others' work of this batch follows it); phases: 1 = tier 0, 2 = everything behind it, 3 = both (a pair
// call enqueues both batches' tier 0 before anything else); pe: the batch's profile events
struct SpanProf { hipEvent_t ev[32]; };
static int span_launch(thj_ctx* c, const thj_params* tp, const thj_span_batch* db, int set, uint32_t base, hipStream_t st0, hipStream_t sm, hipStream_t sa, hipStream_t sp,
                       hipEvent_t ev_t0, hipEvent_t ev_fork, hipEvent_t ev_joined, int phases, SpanProf& pe, hipEvent_t ev_sets = nullptr) {
    // ev_sets: (tier 0 went out ahead of the junction set) what reads the set waits for this event; thj_k_chains does not
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    Params p; memcpy(&p, tp, sizeof p);
    DevSpanBatch b; memcpy(&b, db, sizeof b);
    SpanSets S{c->d_span_junc, c->n_span_junc, c->d_span_ins_key, c->d_span_ins_seq, c->n_span_ins, c->d_junc_bucket, c->n_junc_buckets};
#ifdef THJ_EXP
    { int f = getenv("THJ_EXP_FLAGS") ? atoi(getenv("THJ_EXP_FLAGS")) : 0; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(thj_exp_flags), &f, sizeof f)); }
#endif
    RecSink sink{(OutAln*)c->d_aln_pool, c->d_nrec, base, (OutAln*)c->d_aln_sorted, c->d_aln_keys, c->d_aln_count + 1,
                 (unsigned long long)c->ovf_cap, c->d_aln_count, c->d_span_status, 0, 0};
    // block-owned slices of the worklists (see Tiers)
    int64_t G = ((int64_t)b.n_reads + 511) / 512;
    if (G > MAX_SLICES) G = MAX_SLICES;               // 4 resident workgroups per CU: one full wave of equal chunks
    if (G < 1) G = 1;
    int64_t chunk = ((int64_t)b.n_reads + G - 1) / G;
    chunk = (chunk + 255) / 256 * 256;
    G = ((int64_t)b.n_reads + chunk - 1) / chunk;
    const int NC = SPAN_LEAN_CLASSES;
    // THJ_NO_CHAINS: developer switch -- every one-hit-per-segment read through thj_k_stitch, as before round 5
    static const bool no_chains = getenv("THJ_NO_CHAINS") != nullptr;
    const bool chains = !no_chains && !p.fusion_search && b.nseg <= CHAIN_MAXSEG;
    if (!chains) sp = sm;                       // (the packed tier's list is then tier 0's own: no fork to wait for)
    int rc = ensure_span_set(c, set, b.n_reads, G, chunk, chains);
    if (rc) return rc;
    thj_ctx::SpanSet& ss = c->span_set[set];
    Tiers t;
    t.wl_lean = ss.d_worklist;
    t.wl_multi = ss.d_worklist + NC * G * chunk;
    t.wl_gen = t.wl_multi + G * chunk;
    t.blk_lean = t.wl_gen + 2 * G * chunk;
    t.blk_multi = t.blk_lean + NC * MAX_SLICES;
    t.blk_gen = t.blk_multi + MAX_SLICES;
    t.blk_chain = t.blk_gen + MAX_SLICES;
    uint32_t* const wl_pack = t.wl_gen + G * chunk;                          // what thj_k_chains leaves for the packed tier
    unsigned int* const blk_pack = t.blk_chain + NC * MAX_SLICES;
    t.counters = &c->d_span_status[16 + SPAN_CNT_WORDS * set];
    t.chunk = (int)chunk;
    t.huge_list = c->d_huge_list ? c->d_huge_list + (size_t)set * c->huge_list_cap : nullptr; t.huge_cnt = t.counters + 4;      // a list and a workspace per scratch set: the two batches of a pair call run beside each other
    t.huge_list_cap = c->d_huge_list ? c->huge_list_cap : 0;
    t.ent = chains ? (ChainEntry*)ss.d_ent : nullptr;
    t.ja = (Q16*)ss.d_joined; t.jb = t.ja ? t.ja + ss.joined_cap : nullptr; t.jc = t.ja ? t.ja + 2 * ss.joined_cap : nullptr;
    t.status = c->d_span_status;
    static_assert(2 * SPK_N <= 32, "SpanProf");
    hipEvent_t* const ev = pe.ev;
    const bool prof = c->span_profile;
#define SPK_BEGIN(k, st) do { if (prof) HIPCHK(hipEventRecord(ev[2 * (k)], st)); } while (0)
#define SPK_END(k, st) do { if (prof) HIPCHK(hipEventRecord(ev[2 * (k) + 1], st)); } while (0)
    if (phases & 1) {
        HIPCHK(hipMemsetAsync(t.counters, 0, SPAN_CNT_WORDS * 4, st0));
        HIPCHK(hipMemsetAsync(t.blk_gen, 0, (size_t)MAX_SLICES * 4, st0));      // tiers 0 / 1 write the others
        c->span_last_set = set;
        for (int k = 0; k < 2 * SPK_N; ++k) ev[k] = prof ? thj_get_event(c) : nullptr;
        SPK_BEGIN(SPK_CONTIG, st0);
        if (b.nseg <= 4) hipLaunchKernelGGL(thj_k_stitch_contig<4>, dim3((unsigned)G), dim3(256), 0, st0, g, p, b, sink, t);
        else if (b.nseg <= SPAN_MIDSEG) hipLaunchKernelGGL(thj_k_stitch_contig<SPAN_MIDSEG>, dim3((unsigned)G), dim3(256), 0, st0, g, p, b, sink, t);
        else hipLaunchKernelGGL(thj_k_stitch_contig<SPAN_MAXSEG>, dim3((unsigned)G), dim3(256), 0, st0, g, p, b, sink, t);
        SPK_END(SPK_CONTIG, st0);
        if (st0 != sm) { HIPCHK(hipEventRecord(ev_t0, st0)); HIPCHK(hipStreamWaitEvent(sm, ev_t0, 0)); }
        HIPCHK(hipGetLastError());
    }
    if (!(phases & 2)) return THJ_OK;
    // ---- the chains: those of the multihit reads that need no search join tier 0's entries (thj_k_chains); then join, closure
    // search, finish on the side stream, beside the kernels of the reads that are left
    Tiers tpk = t;                // the packed tier's view: the list thj_k_chains leaves
    if (chains) {
        const int64_t cap2 = chain_cap2(b.n_reads), G2 = chain_g2(G);
        unsigned int* const n2 = t.counters + 5;
        ChainEntry* const ent2 = (ChainEntry*)ss.d_ent + NC * G * chunk;
        SPK_BEGIN(SPK_CHAINS, sm);
        hipLaunchKernelGGL(thj_k_chains, dim3((unsigned)G), dim3(CH_TPB), 0, sm, p, b, t, ent2, n2, (unsigned int)cap2, wl_pack, blk_pack, (int)G);
        SPK_END(SPK_CHAINS, sm);
        tpk.wl_multi = wl_pack; tpk.blk_multi = blk_pack;
        if (sa != sm || sp != sm) HIPCHK(hipEventRecord(ev_fork, sm));
        if (sa != sm) HIPCHK(hipStreamWaitEvent(sa, ev_fork, 0));
        if (sp != sm) HIPCHK(hipStreamWaitEvent(sp, ev_fork, 0));
        const ChainLists cl{t.ent, t.blk_chain, (int)G, (int)chunk, ent2, n2, (int)G2, (unsigned int)cap2, (unsigned int)chain_slice2(b.n_reads, G), t.ja, t.jb, t.jc};
        const DeferList dl{ss.d_defer, (unsigned int*)(ss.d_defer + G * chunk + G2 * chain_slice2(b.n_reads, G))};
        // THJ_JOIN_WPE = 4 / THJ_FIN_WPE = 3: developer switches -- thj_k_join_closure with four workgroups' worth of registers per CU (128 VGPRs,
        // 44 spilled) instead of three (162, nothing spilled), thj_k_finish the other way round
        static const int join_wpe = getenv("THJ_JOIN_WPE") ? atoi(getenv("THJ_JOIN_WPE")) : 3, fin_wpe = getenv("THJ_FIN_WPE") ? atoi(getenv("THJ_FIN_WPE")) : 4;
        const dim3 grid((unsigned)(G + G2));
        if (ev_sets) HIPCHK(hipStreamWaitEvent(sa, ev_sets, 0));
        SPK_BEGIN(SPK_JOIN, sa);
        static const int abut_wpe = getenv("THJ_ABUT_WPE") ? atoi(getenv("THJ_ABUT_WPE")) : 6;       // developer switch
        if (abut_wpe == 4) hipLaunchKernelGGL(thj_k_join<4>, grid, dim3(256), 0, sa, g, p, S, b.hits, b.planes, b.W, cl, t, dl);
        else hipLaunchKernelGGL(thj_k_join<6>, grid, dim3(256), 0, sa, g, p, S, b.hits, b.planes, b.W, cl, t, dl);
        SPK_END(SPK_JOIN, sa);
        SPK_BEGIN(SPK_CLOSURE, sa);
        if (join_wpe == 3) hipLaunchKernelGGL(thj_k_join_closure<3>, grid, dim3(256), 0, sa, g, p, S, b.hits, b.planes, b.W, cl, t, dl);
        else hipLaunchKernelGGL(thj_k_join_closure<4>, grid, dim3(256), 0, sa, g, p, S, b.hits, b.planes, b.W, cl, t, dl);
        SPK_END(SPK_CLOSURE, sa);
        SPK_BEGIN(SPK_FINISH, sa);
        if (fin_wpe == 5) hipLaunchKernelGGL(thj_k_finish<5>, grid, dim3(256), 0, sa, g, p, b, sink, cl);
        else if (fin_wpe == 6) hipLaunchKernelGGL(thj_k_finish<6>, grid, dim3(256), 0, sa, g, p, b, sink, cl);
        else if (fin_wpe == 3) hipLaunchKernelGGL(thj_k_finish<3>, grid, dim3(256), 0, sa, g, p, b, sink, cl);
        else hipLaunchKernelGGL(thj_k_finish<4>, grid, dim3(256), 0, sa, g, p, b, sink, cl);
        SPK_END(SPK_FINISH, sa);
    } else {
        if (ev_sets) HIPCHK(hipStreamWaitEvent(sm, ev_sets, 0));
        SPK_BEGIN(SPK_CHAINS, sm); SPK_END(SPK_CHAINS, sm); SPK_BEGIN(SPK_JOIN, sm); SPK_END(SPK_JOIN, sm);
        SPK_BEGIN(SPK_CLOSURE, sm); SPK_END(SPK_CLOSURE, sm); SPK_BEGIN(SPK_FINISH, sm); SPK_END(SPK_FINISH, sm);
    }
    const int64_t g1 = G, g2 = G;
    // THJ_LEAN_WPE = 2 | 3: developer switch -- tier 1 with two or three workgroups' worth of registers per CU (256 / 168 VGPRs, nothing
    // spilled) instead of four (128 VGPRs, 6 spilled): 0.92 / 0.75 ms per launch against 0.70 (profiles/r04_zzz_stage2_occupancy_ab2.txt;
    // the packed tier below is the other way round)
    static const int lean_wpe = getenv("THJ_LEAN_WPE") ? atoi(getenv("THJ_LEAN_WPE")) : 4;
    const size_t lean_lds = (size_t)256 * b.nseg * sizeof(SpanHitHead);
    SPK_BEGIN(SPK_LEAN, sm);
    if (chains) { /* every one-hit-per-segment read travels as a chain entry: thj_k_stitch's list is empty */ }
    else if (b.nseg <= 4 && lean_wpe == 3) hipLaunchKernelGGL((thj_k_stitch<4, 3>), dim3((unsigned)((g1 * 3 + 3) / 4)), dim3(256), lean_lds, sm, g, p, S, b, sink, t, (int)G);
    else if (b.nseg <= 4 && lean_wpe == 2) hipLaunchKernelGGL((thj_k_stitch<4, 2>), dim3((unsigned)((g1 + 1) / 2)), dim3(256), lean_lds, sm, g, p, S, b, sink, t, (int)G);
    else if (b.nseg <= 4) hipLaunchKernelGGL(thj_k_stitch<4>, dim3((unsigned)g1), dim3(256), lean_lds, sm, g, p, S, b, sink, t, (int)G);
    else if (b.nseg <= SPAN_MIDSEG) hipLaunchKernelGGL(thj_k_stitch<SPAN_MIDSEG>, dim3((unsigned)g1), dim3(256), (size_t)256 * b.nseg * sizeof(SpanHitHead), sm, g, p, S, b, sink, t, (int)G);
    else {
        // a workgroup's staging area passes the 64 KB a launch may ask for by default: say so once
        static const hipError_t big1 = hipFuncSetAttribute((const void*)thj_k_stitch<SPAN_MAXSEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 * SPAN_MAXSEG * (int)sizeof(SpanHitHead));
        HIPCHK(big1);
        hipLaunchKernelGGL(thj_k_stitch<SPAN_MAXSEG>, dim3((unsigned)g1), dim3(256), (size_t)256 * b.nseg * sizeof(SpanHitHead), sm, g, p, S, b, sink, t, (int)G);
    }
    SPK_END(SPK_LEAN, sm);
    hipStream_t sg = sm;                         // where the general tier's kernel runs
    if (p.fusion_search) {
        // tiers 0 and 1 keep the reads that join without a fusion; multihit reads, reads with a fused segment hit and reads tier 1
        // could not join are on the multihit list and go through the fusion branches
        FusionSet F{(const FusKey*)c->d_span_fus, c->n_span_fus};
        int64_t gf = ((int64_t)b.n_reads + 63) / 64;
        static const int fusion_grid = getenv("THJ_FUSION_GRID") ? atoi(getenv("THJ_FUSION_GRID")) : 8192;      // developer switch
        if (gf > fusion_grid) gf = fusion_grid;
        SPK_BEGIN(SPK_PACK, sm); SPK_END(SPK_PACK, sm);
        SPK_BEGIN(SPK_GENERIC, sm);
        hipLaunchKernelGGL(thj_k_stitch_fusion, dim3((unsigned)gf), dim3(64), 0, sm, g, p, S, F, b, sink, t, (int)G);
        SPK_END(SPK_GENERIC, sm);
    } else {
        static const bool pack_timing = getenv("THJ_PACK_TIMING") != nullptr;         // developer switch: phase times of the packed tier on stderr
        unsigned long long* d_dbg = nullptr;
        if (pack_timing) { HIPCHK(hipMalloc((void**)&d_dbg, 128)); HIPCHK(hipMemsetAsync(d_dbg, 0, 128, sp)); }
        // reads of up to four segments: four waves a workgroup and three workgroups a CU (168 VGPRs, nothing spilled) instead of two
        // workgroups of eight waves (128 VGPRs, 43 spilled): 0.69 -> 0.635 ms per launch; THJ_PACK_WPE = 2 | 4: developer switch
        static const int pack_wpe = getenv("THJ_PACK_WPE") ? atoi(getenv("THJ_PACK_WPE")) : 3;
        static const int pack_draw_env = getenv("THJ_PACK_DRAW") ? atoi(getenv("THJ_PACK_DRAW")) : 0;
        const int pack_draw = pack_draw_env >= 1 && pack_draw_env <= 64 ? pack_draw_env : (chains ? PACK_DRAW_DEFAULT : 32);
        static const int pack_heavy_env = getenv("THJ_PACK_HEAVY_DRAW") ? atoi(getenv("THJ_PACK_HEAVY_DRAW")) : 0;          // developer switch (0, the default: one pass; measured 32: 0.42 ms, 16: 0.32, one pass: 0.345)
        const int pack_heavy = chains && pack_heavy_env >= 1 && pack_heavy_env <= 64 ? pack_heavy_env : 0;
        SPK_BEGIN(SPK_PACK, sp);
        if (b.nseg <= 4 && pack_wpe == 3) hipLaunchKernelGGL((thj_k_stitch_pack<4, 256, 3>), dim3((unsigned)((g2 * 3 + 3) / 4)), dim3(256), 0, sp, g, p, S, b, sink, tpk, (int)G, d_dbg, pack_draw, pack_heavy);
        else if (b.nseg <= 4 && pack_wpe == 2) hipLaunchKernelGGL((thj_k_stitch_pack<4, 256, 2>), dim3((unsigned)((g2 + 1) / 2)), dim3(256), 0, sp, g, p, S, b, sink, tpk, (int)G, d_dbg, pack_draw, pack_heavy);
        else if (b.nseg <= 4) hipLaunchKernelGGL(thj_k_stitch_pack<4>, dim3((unsigned)((g2 + 1) / 2)), dim3(PACK_TPB), 0, sp, g, p, S, b, sink, tpk, (int)G, d_dbg, pack_draw, pack_heavy);
        else if (b.nseg <= SPAN_MIDSEG) hipLaunchKernelGGL(thj_k_stitch_pack<SPAN_MIDSEG>, dim3((unsigned)((g2 + 1) / 2)), dim3(PACK_TPB), 0, sp, g, p, S, b, sink, tpk, (int)G, d_dbg, pack_draw, pack_heavy);
        // (reads of more than eight segments: the packed tier keeps a chain's choices in eight bytes -- the general kernel takes the multihit list as it is)
        SPK_END(SPK_PACK, sp);
        if (pack_timing) {
            unsigned long long h[16];
            HIPCHK(hipStreamSynchronize(sp));
            HIPCHK(hipMemcpy(h, d_dbg, 128, hipMemcpyDeviceToHost));
            (void)hipFree(d_dbg);
            const double nw = h[10] ? (double)h[10] : 1.0;
            fprintf(stderr, "[packed tier] %llu waves, %llu rounds, %llu sub-rounds; per wave us: all %.1f (max %.1f) = entries %.1f + staging %.1f + searches %.1f + joins %.1f + rank %.1f + tags and records %.1f\n",
                    h[10], h[6], h[7], h[8] / nw / 100.0, h[9] / 100.0, h[0] / nw / 100.0, h[1] / nw / 100.0, h[2] / nw / 100.0, h[3] / nw / 100.0, h[4] / nw / 100.0, h[5] / nw / 100.0);
        }
        Tiers tg = t;
        if (b.nseg > SPAN_MIDSEG) {
            tg.wl_gen = t.wl_multi; tg.blk_gen = t.blk_multi;
            static const hipError_t big3 = hipFuncSetAttribute((const void*)thj_k_stitch_generic, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * SPAN_MAXSEG * (int)sizeof(SpanHit));
            HIPCHK(big3);
        }
        // the general tier's list takes reads from the packed tier (its stream) and from thj_k_join / thj_k_join_closure (the side
        // stream): its kernel goes behind thj_k_finish on the side stream and waits there for the packed tier
        if (chains) sg = sa;
        if (sg != sp) { HIPCHK(hipEventRecord(ev_joined, sp)); HIPCHK(hipStreamWaitEvent(sg, ev_joined, 0)); }
        SPK_BEGIN(SPK_GENERIC, sg);
        hipLaunchKernelGGL(thj_k_stitch_generic, dim3((unsigned)g2), dim3(128), (size_t)128 * b.nseg * sizeof(SpanHit), sg, g, p, S, b, sink, tg, (int)G);
        SPK_END(SPK_GENERIC, sg);
    }
#undef SPK_BEGIN
#undef SPK_END
    if (prof) for (int k = 0; k < SPK_N; ++k) c->span_prof_events.emplace_back(ev[2 * k], ev[2 * k + 1]);
    if (c->d_huge_ws && sg != sm) { HIPCHK(hipEventRecord(ev_fork, sg)); HIPCHK(hipStreamWaitEvent(sm, ev_fork, 0)); }      // (the fork's wait is long enqueued: the event is free)
    if (c->d_huge_ws) {        // a pass that met a read with too many joined alignments runs with the workspace from then on
        FusionSet F{(const FusKey*)c->d_span_fus, c->n_span_fus};
        static const int huge_by_wave = getenv("THJ_HUGE_ONE_LANE") ? 0 : getenv("THJ_HUGE_TIMERS") ? 2 : 1;      // developer switch: the fusion reads of the list by lane 0 alone (rounds 4-5)
        hipLaunchKernelGGL(thj_k_stitch_huge, dim3((unsigned)c->huge_blocks), dim3(64), 0, sm, g, p, S, F, b, sink, t, (char*)c->d_huge_ws + (size_t)set * c->huge_blocks * HUGE_BLOCK_BYTES, HUGE_CAP, huge_by_wave);
    }
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

// blocks: the workgroups of thj_k_stitch_huge, each with a slice of the workspace.  Under --fusion-search the reads of the list are many (every
// read with a few hits a segment: fusion_read_heavy) and each is a wave's work for milliseconds, so a large batch gets up to 1 024 of them
// (bench.py's 10 M pairs in one batch: 450 000 listed reads, 0.8 -> 0.35 s a launch); a shard of the executables gets 256 and up, a slice per 1 024 reads.
// list_reads: the reads of the largest batch of the call -- under --fusion-search a good part of them may be listed (a read the list has no
// room for stays with its thread: correct, and slow when it is a family read)
static int ensure_huge_workspace(thj_ctx* c, int blocks, int64_t list_reads) {
    if (blocks < HUGE_BLOCKS) blocks = HUGE_BLOCKS;
    if (blocks > HUGE_BLOCKS_MAX) blocks = HUGE_BLOCKS_MAX;
    int list_cap = list_reads > HUGE_LIST_CAP ? (int)(list_reads < (1ll << 30) ? list_reads : (1ll << 30)) : HUGE_LIST_CAP;
    const bool ws_ok = c->d_huge_ws && c->huge_blocks >= blocks, list_ok = c->d_huge_list && c->huge_list_cap >= list_cap;
    if (ws_ok && list_ok) return THJ_OK;
    if (c->d_huge_ws || c->d_huge_list) HIPCHK(hipDeviceSynchronize());       // (a launch may still be using the smaller ones)
    if (!ws_ok) {
        hipFree(c->d_huge_ws); c->d_huge_ws = nullptr;
        HIPCHK(hipMalloc(&c->d_huge_ws, (size_t)2 * blocks * HUGE_BLOCK_BYTES));                  // (one per scratch set)
        c->huge_blocks = blocks;
    }
    if (!list_ok) {
        hipFree(c->d_huge_list); c->d_huge_list = nullptr;
        HIPCHK(hipMalloc(&c->d_huge_list, (size_t)2 * list_cap * 4));
        c->huge_list_cap = list_cap;
    }
    return THJ_OK;
}

// mode: 3 = a whole run; 1 = thj_span_tier0_pair_async (tier 0 of a pair, nothing else); a run that finds tier 0 done only does the rest
static int span_run_common(thj_ctx* c, const thj_params* tp, const thj_span_batch* db0, const thj_span_batch* db1, int mode = 3) {
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    int rc = check_span_params(tp, db0);
    if (!rc && db1) rc = check_span_params(tp, db1);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    // THJ_TRACE: what a context's first run spends before its kernels go out (seconds since the call began), on stderr
    static const bool trace_env = getenv("THJ_TRACE") != nullptr;
    const bool trace_first = trace_env && !c->span_ev[0];
    const auto tr0 = std::chrono::steady_clock::now();
    auto lapse = [&](const char* what) { if (trace_first) fprintf(stderr, "[trace] first run of a context: %-28s %.4f\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count()); };
    if ((rc = ensure_span_state(c))) return rc;
    // --fusion-search: the reads with many hits a segment go to thj_k_stitch_huge from the start (thj_k_stitch_fusion's fusion_read_heavy)
    if (tp->fusion_search && (rc = ensure_huge_workspace(c, (int)((db0->n_reads + (db1 ? db1->n_reads : 0) + 1023) / 1024),
                                                         db1 && db1->n_reads > db0->n_reads ? db1->n_reads : db0->n_reads))) return rc;
    lapse("state");
    const int64_t n0 = db0->n_reads, n1 = db1 ? db1->n_reads : 0;
    const bool t0_done = c->span_t0_pending;
    if (t0_done && (mode == 1 || !db1 || n0 != c->span_t0_n[0] || n1 != c->span_t0_n[1])) {
        thj_set_error("thj_span_tier0_pair_async is followed by thj_span_run_pair_async on the same two batches");
        return THJ_ESTATE;
    }
    if (n0 + n1 == 0) return THJ_OK;
    if ((rc = ensure_sets_cap(c, c->n_span_junc, c->n_span_ins))) return rc;
    if (c->span_reads + n0 + n1 >= (1ll << 32)) { thj_set_error("more than 2^32 reads in one pass"); return THJ_EINVAL; }
    lapse("sets' room");
    if ((rc = ensure_slots(c, c->span_reads + n0 + n1))) return rc;
    lapse("slots");
    {
        static const bool own_a1_env = getenv("THJ_SPAN_A1") != nullptr;
        if ((rc = ensure_span_streams(c, (n0 && n1) ? (own_a1_env ? 3 : 2) : 1))) return rc;
    }
    lapse("streams (with the queue probe)");
    // THJ_SPAN_SERIAL: developer switch -- every kernel on the context's stream, one after the other
    static const bool serial_env = getenv("THJ_SPAN_SERIAL") != nullptr;
    const bool serial = serial_env || c->serial_launch;
    // Three streams, whatever the call holds (two streams on one hardware queue run one after the other: thj_ensure_aux_streams).  Pair
    // call: the first batch's tier 0 on the context's stream, the second's beside it on its side stream; behind its tier 0 each batch's
    // chains / join / closure search / finish / general tier on a side stream of its own, the packed tiers one after the other on the
    // context's stream.  One batch: its tier 0, chains and packed tier on the context's stream, the join chain on a side stream.
    // (THJ_SPAN_T0_SERIAL: developer switch -- the second batch's tier 0 behind the first's on the context's stream.  Each fills the HBM on
    // its own, but behind the first the second runs beside the first batch's chains and join and is the slower for it: 5.65-5.71 against
    // 5.46-5.50 ms per step, profiles/r05_m_timeline.txt.)
    static const bool own_a1 = getenv("THJ_SPAN_A1") != nullptr;      // developer switch: the second batch's join chain on the third side stream
    static const bool t0_beside = getenv("THJ_SPAN_T0_SERIAL") == nullptr;
    const uint32_t base0 = (uint32_t)c->span_reads, base1 = (uint32_t)(c->span_reads + n0);
    SpanProf pe0, pe1;
    hipStream_t cs = c->stream;
    if (mode == 1) {
        // tier 0 needs the batches and nothing of stage 1: enqueued before the caller asks stage 1 for its counts and sorts its events, it
        // fills the third of a millisecond the GPU otherwise waits through (one host round trip, twenty small sort kernels).  Both
        // batches' tier 0 on the side streams, behind what the context's stream holds now; the context's stream stays free for the sorts.
        if (serial || !n0 || !n1 || tp->fusion_search) return THJ_OK;              // (the run does everything)
        hipStream_t x0 = c->span_stream[0], x1 = c->span_stream[1];
        HIPCHK(hipEventRecord(c->span_ev[0], cs)); HIPCHK(hipStreamWaitEvent(x0, c->span_ev[0], 0)); HIPCHK(hipStreamWaitEvent(x1, c->span_ev[0], 0));
        if ((rc = span_launch(c, tp, db0, 0, base0, x0, x0, x0, cs, c->span_ev[8], c->span_ev[1], c->span_ev[6], 1, pe0))) return rc;
        if ((rc = span_launch(c, tp, db1, 1, base1, x1, x1, x1, cs, c->span_ev[9], c->span_ev[2], c->span_ev[7], 1, pe1))) return rc;
        memcpy(c->span_t0_prof[0], pe0.ev, sizeof pe0.ev); memcpy(c->span_t0_prof[1], pe1.ev, sizeof pe1.ev);
        c->span_t0_pending = true; c->span_t0_n[0] = n0; c->span_t0_n[1] = n1;
        return THJ_OK;
    }
    if (t0_done) {
        hipStream_t x0 = c->span_stream[0], x1 = c->span_stream[1];
        memcpy(pe0.ev, c->span_t0_prof[0], sizeof pe0.ev); memcpy(pe1.ev, c->span_t0_prof[1], sizeof pe1.ev);
        c->span_t0_pending = false;
        // (the sets were built on the context's stream after tier 0 went out: the kernels that read them wait for this event)
        HIPCHK(hipEventRecord(c->span_ev[0], cs));
        if ((rc = span_launch(c, tp, db0, 0, base0, x0, x0, x0, cs, c->span_ev[8], c->span_ev[1], c->span_ev[6], 2, pe0, c->span_ev[0]))) return rc;
        if ((rc = span_launch(c, tp, db1, 1, base1, x1, x1, x1, cs, c->span_ev[9], c->span_ev[2], c->span_ev[7], 2, pe1, c->span_ev[0]))) return rc;
        HIPCHK(hipEventRecord(c->span_ev[3], x0)); HIPCHK(hipStreamWaitEvent(cs, c->span_ev[3], 0));
        HIPCHK(hipEventRecord(c->span_ev[4], x1)); HIPCHK(hipStreamWaitEvent(cs, c->span_ev[4], 0));
    } else if (serial) {
        if (n0 && (rc = span_launch(c, tp, db0, 0, base0, cs, cs, cs, cs, c->span_ev[8], c->span_ev[1], c->span_ev[6], 3, pe0))) return rc;
        if (n1 && (rc = span_launch(c, tp, db1, 1, base1, cs, cs, cs, cs, c->span_ev[9], c->span_ev[2], c->span_ev[7], 3, pe1))) return rc;
    } else if (!n1 || !n0) {
        const thj_span_batch* db = n0 ? db0 : db1;
        const int set = n0 ? 0 : 1;
        if ((rc = span_launch(c, tp, db, set, base0, cs, cs, c->span_stream[0], cs, c->span_ev[8], c->span_ev[1], c->span_ev[6], 3, pe0))) return rc;
        HIPCHK(hipEventRecord(c->span_ev[3], c->span_stream[0])); HIPCHK(hipStreamWaitEvent(cs, c->span_ev[3], 0));
    } else {
        hipStream_t x0 = c->span_stream[0], x1 = c->span_stream[1], a1 = own_a1 ? c->span_stream[2] : x1;
        hipStream_t t1 = t0_beside ? x1 : cs;
        if (t0_beside) { HIPCHK(hipEventRecord(c->span_ev[0], cs)); HIPCHK(hipStreamWaitEvent(x1, c->span_ev[0], 0)); }
        if ((rc = span_launch(c, tp, db0, 0, base0, cs, x0, x0, cs, c->span_ev[8], c->span_ev[1], c->span_ev[6], 1, pe0))) return rc;
        if ((rc = span_launch(c, tp, db1, 1, base1, t1, x1, a1, own_a1 ? x1 : cs, c->span_ev[9], c->span_ev[2], c->span_ev[7], 1, pe1))) return rc;
        if ((rc = span_launch(c, tp, db0, 0, base0, cs, x0, x0, cs, c->span_ev[8], c->span_ev[1], c->span_ev[6], 2, pe0))) return rc;
        if ((rc = span_launch(c, tp, db1, 1, base1, t1, x1, a1, own_a1 ? x1 : cs, c->span_ev[9], c->span_ev[2], c->span_ev[7], 2, pe1))) return rc;
        HIPCHK(hipEventRecord(c->span_ev[3], x0)); HIPCHK(hipStreamWaitEvent(cs, c->span_ev[3], 0));
        HIPCHK(hipEventRecord(c->span_ev[4], x1)); HIPCHK(hipStreamWaitEvent(cs, c->span_ev[4], 0));
        if (a1 != x1) { HIPCHK(hipEventRecord(c->span_ev[5], a1)); HIPCHK(hipStreamWaitEvent(cs, c->span_ev[5], 0)); }
    }
    c->span_reads += n0 + n1;
    lapse("kernels enqueued");
    if (trace_first) { (void)hipStreamSynchronize(c->stream); lapse("kernels done"); }
    return THJ_OK;
}

extern "C" int thj_span_run_async(thj_ctx* c, const thj_params* tp, const thj_span_batch* db) {
    if (!c || !tp || !db) { thj_set_error("thj_span_run_async: null argument"); return THJ_EINVAL; }
    return span_run_common(c, tp, db, nullptr);
}

extern "C" int thj_span_tier0_pair_async(thj_ctx* c, const thj_params* tp, const thj_span_batch* db0, const thj_span_batch* db1) {
    if (!c || !tp || !db0 || !db1) { thj_set_error("thj_span_tier0_pair_async: null argument"); return THJ_EINVAL; }
    return span_run_common(c, tp, db0, db1, 1);
}

extern "C" int thj_span_run_pair_async(thj_ctx* c, const thj_params* tp, const thj_span_batch* db0, const thj_span_batch* db1) {
    if (!c || !tp || !db0 || !db1) { thj_set_error("thj_span_run_pair_async: null argument"); return THJ_EINVAL; }
    return span_run_common(c, tp, db0, db1);
}

extern "C" int thj_span_finish(thj_ctx* c, int64_t* n_alns) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    if (c->span_t0_pending) { thj_set_error("thj_span_tier0_pair_async is followed by thj_span_run_pair_async on the same two batches"); return THJ_ESTATE; }
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_span_state(c);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(&c->h_pinned[24], c->d_aln_count, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&c->h_pinned[26], c->d_span_status, 32, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const unsigned int* st = (const unsigned int*)&c->h_pinned[26];
    {
        static const bool huge_timers = getenv("THJ_HUGE_TIMERS") != nullptr;
        if (huge_timers) {
            unsigned int tm[8];
            HIPCHK(hipMemcpy(tm, c->d_span_status + 8, sizeof tm, hipMemcpyDeviceToHost));
            fprintf(stderr, "[huge timers] lane-0 ms summed over reads: search %.1f  reorder %.1f  sort %.1f  records %.1f   longest list %u  reads %u  leaves %u  whole pair tests %u\n",
                    tm[0] * 1e-5, tm[1] * 1e-5, tm[2] * 1e-5, tm[3] * 1e-5, tm[4], tm[5], tm[6], tm[7]);
        }
    }
    if (st[5]) {      // thj_k_chains only lets through chains whose joined hit fits the registers' cigar ops: a chain of a group that does not is a bug, not an input
        thj_set_error("internal: a chain of a multihit read needed more cigar ops than thj_k_join holds (set THJ_NO_CHAINS=1 and report)");
        return THJ_ESTATE;
    }
    if (st[3]) {
        // more 2nd.. records of multihit reads than the pool holds.  The counter kept counting, so the need is known: make the
        // pool that large (and then some) and ask for the pass again -- the slots are rewritten by the rerun, nothing is kept
        const int64_t need = (int64_t)c->h_pinned[25];
        const int64_t ncap = need + need / 4 + 4096;
        hipFree(c->d_aln_sorted); hipFree(c->d_aln_keys);
        c->d_aln_sorted = nullptr; c->d_aln_keys = nullptr; c->ovf_cap = 0;
        HIPCHK(hipMalloc(&c->d_aln_sorted, (size_t)ncap * 128));
        HIPCHK(hipMalloc(&c->d_aln_keys, (size_t)ncap * 8));
        c->ovf_cap = ncap;
        thj_set_error("the pool for the extra records of multihit reads was too small (%lld needed); it has been enlarged: run the pass again "
                      "(thj_span_reset_async, the thj_span_run_async calls, thj_span_finish)", (long long)need);
        return THJ_ERETRY;
    }
    if (st[SPAN_TOO_MANY_JOINED]) {
        if (!c->d_huge_ws) {
            // a read has more joined alignments than a thread's own array holds: get the big workspace (HUGE_BLOCKS slices of
            // 2 * HUGE_CAP records) and ask for the pass again -- thj_k_stitch_huge then takes such reads one by one
            int rc2 = ensure_huge_workspace(c, HUGE_BLOCKS, 0);
            if (rc2) return rc2;
            thj_set_error("%u read(s) have more joined alignments than the stitch kernels keep per thread (%d, %d with fusion search); a workspace "
                          "for them has been set up: run the pass again (thj_span_reset_async, the thj_span_run_async calls, thj_span_finish)",
                          st[SPAN_TOO_MANY_JOINED], SPAN_MAXJOIN, FUS_MAXJOIN);
            return THJ_ERETRY;
        }
        thj_set_error("%u read(s) have more than %d joined alignments before sort + unique (device limit)", st[SPAN_TOO_MANY_JOINED], HUGE_CAP);
        return THJ_EOVERFLOW;
    }
    c->n_alns = (int64_t)c->h_pinned[24];
    c->n_ovf = (int64_t)c->h_pinned[25];
    c->h_alns.clear();
    if (n_alns) *n_alns = c->n_alns;
    return THJ_OK;
}

extern "C" int thj_span_device_records(thj_ctx* c, const thj_aln_slot** d_slots, const uint8_t** d_counts, int64_t* n_reads,
                                       const thj_aln_slot** d_extra, const uint64_t** d_extra_keys, int64_t* n_extra) {
    if (!c || !d_slots || !d_counts || !n_reads) { thj_set_error("thj_span_device_records: bad argument"); return THJ_EINVAL; }
    *d_slots = (const thj_aln_slot*)c->d_aln_pool; *d_counts = c->d_nrec; *n_reads = c->span_reads;
    if (d_extra) *d_extra = (const thj_aln_slot*)c->d_aln_sorted;
    if (d_extra_keys) *d_extra_keys = (const uint64_t*)c->d_aln_keys;
    if (n_extra) *n_extra = c->n_ovf;
    return THJ_OK;
}

// one slot (device layout, see RecSink) -> the API record; fields the record does not use read as zero
static inline void slot_to_aln(const thj_aln& slot, thj_aln& out) {
    uint32_t s[32], w[32];
    memcpy(s, &slot, 128);
    const bool tail = (s[3] & SLOT_TAIL) != 0;
    for (int k = 0; k < 10; ++k) w[k] = s[k];
    w[3] &= ~SLOT_TAIL;
    for (int k = 10; k < 16; ++k) w[k + 12] = s[k];
    for (int k = 16; k < 28; ++k) w[k - 6] = tail ? s[k] : 0u;
    for (int k = 28; k < 32; ++k) w[k] = tail ? s[k] : 0u;
    memcpy(&out, w, 128);
}

// Compaction on the device: the position of a read's first record in the compact, ordered array is the exclusive prefix sum of the
// per-read record counts; its 2nd.. records (the extra pool, keyed slot << 16 | rank) follow at + rank.  One thread per record
// converts the slot layout to thj_aln on the way.  (The per-read count saturates at 255: a pass with such a read does not add up
// to n_alns and takes the host path below.)
__device__ __forceinline__ void slot_to_aln_dev(const uint4* src, uint4* dst) {
    uint32_t s[32], w[32];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const uint4 v = src[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
    const bool tail = (s[3] & SLOT_TAIL) != 0;
    if (tail) {
#pragma unroll
        for (int k = 4; k < 8; ++k) { const uint4 v = src[k]; s[4 * k] = v.x; s[4 * k + 1] = v.y; s[4 * k + 2] = v.z; s[4 * k + 3] = v.w; }
    } else {
#pragma unroll
        for (int k = 16; k < 32; ++k) s[k] = 0u;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) w[k] = s[k];
    w[3] &= ~SLOT_TAIL;
#pragma unroll
    for (int k = 10; k < 16; ++k) w[k + 12] = s[k];
#pragma unroll
    for (int k = 16; k < 28; ++k) w[k - 6] = s[k];
#pragma unroll
    for (int k = 28; k < 32; ++k) w[k] = s[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
__global__ __launch_bounds__(256) void thj_k_compact_records(const OutAln* slots, const uint8_t* nrec, const uint32_t* off, int64_t n_slots,
                                                             const OutAln* extra, const u64* extra_key, int64_t n_extra, OutAln* out, int64_t n_out,
                                                             unsigned int* bad) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_slots + n_extra; i += (int64_t)gridDim.x * blockDim.x) {
        const OutAln* src; int64_t at;
        if (i < n_slots) {
            if (i == n_slots - 1 && (int64_t)off[i] + nrec[i] != n_out) atomicExch(bad, 1u);     // the counts do not add up (a saturated count)
            if (!nrec[i]) continue;
            src = slots + i; at = off[i];
        } else {
            const u64 k = extra_key[i - n_slots];
            src = extra + (i - n_slots); at = (int64_t)off[k >> 16] + (int64_t)(k & 0xFFFFu);
        }
        if (at >= n_out) { atomicExch(bad, 1u); continue; }
        slot_to_aln_dev((const uint4*)src, (uint4*)(out + at));
    }
}

static int span_download_host(thj_ctx* c, thj_aln* out);

// the pass's records compacted and ordered on the device: *d_out = n_alns API-layout records (the caller releases it with
// thj_dev_release); THJ_EFALLBACK when the device compaction cannot be used (the caller then takes span_download_host's way)
int thj_span_compact_device(thj_ctx* c, void** d_out_p) {
    *d_out_p = nullptr;
    const int64_t nr = c->span_reads, n = c->n_alns;
    static const bool host_path = getenv("THJ_DOWNLOAD_ON_HOST") != nullptr;
    if (host_path || n >= (1ll << 32) || nr < 1) return THJ_EFALLBACK;
    void *d_off = nullptr, *d_out = nullptr, *d_tmp = nullptr;
    const size_t tmp_bytes = thj_scan::scratch_bytes(nr, 4);          // (thj_scan.h: not hipcub::DeviceScan)
    int rc = thj_dev_alloc(c, &d_off, (size_t)nr * 4 + 16);
    if (!rc) rc = thj_dev_alloc(c, &d_out, (size_t)n * 128);
    if (!rc) rc = thj_dev_alloc(c, &d_tmp, tmp_bytes + 16);
    if (rc) { if (d_off) thj_dev_release(c, d_off); if (d_out) thj_dev_release(c, d_out); if (d_tmp) thj_dev_release(c, d_tmp); return THJ_EFALLBACK; }
    unsigned int* d_bad = (unsigned int*)((char*)d_off + (size_t)nr * 4);
    HIPCHK(hipMemsetAsync(d_bad, 0, 4, c->stream));
    thj_scan::exclusive_sum<uint8_t, uint32_t>(c->stream, (const uint8_t*)c->d_nrec, (uint32_t*)d_off, nr, d_tmp);
    const int64_t items = nr + c->n_ovf;
    int64_t grid = (items + 255) / 256; if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(thj_k_compact_records, dim3((unsigned)grid), dim3(256), 0, c->stream, (const OutAln*)c->d_aln_pool, (const uint8_t*)c->d_nrec, (const uint32_t*)d_off, nr,
                       (const OutAln*)c->d_aln_sorted, (const u64*)c->d_aln_keys, c->n_ovf, (OutAln*)d_out, n, d_bad);
    HIPCHK(hipGetLastError());
    unsigned int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    thj_dev_release(c, d_off); thj_dev_release(c, d_tmp);
    if (bad) { thj_dev_release(c, d_out); return THJ_EFALLBACK; }
    *d_out_p = d_out;
    return THJ_OK;
}

extern "C" int thj_span_download(thj_ctx* c, thj_aln* out) {
    // compact, ordered host copy: compacted on the device, one copy down
    if (!c || (c->n_alns > 0 && !out)) { thj_set_error("thj_span_download: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->n_alns == 0) return THJ_OK;
    void* d_out = nullptr;
    const int rc = thj_span_compact_device(c, &d_out);
    if (rc == THJ_EFALLBACK) return span_download_host(c, out);
    if (rc) return rc;
    const hipError_t e = hipMemcpy(out, d_out, (size_t)c->n_alns * 128, hipMemcpyDeviceToHost);
    thj_dev_release(c, d_out);
    HIPCHK(e);
    return THJ_OK;
}

static int span_download_host(thj_ctx* c, thj_aln* out) {
    // the same on the host: walk the slots, splice in the (rank-ordered) extra records of multihit reads
    const int64_t nr = c->span_reads;
    std::vector<uint8_t> cnt((size_t)nr);
    HIPCHK(hipMemcpy(cnt.data(), c->d_nrec, (size_t)nr, hipMemcpyDeviceToHost));
    std::vector<thj_aln> extra((size_t)c->n_ovf);
    std::vector<uint64_t> ekey((size_t)c->n_ovf);
    if (c->n_ovf) {
        HIPCHK(hipMemcpy(extra.data(), c->d_aln_sorted, (size_t)c->n_ovf * 128, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ekey.data(), c->d_aln_keys, (size_t)c->n_ovf * 8, hipMemcpyDeviceToHost));
    }
    std::vector<uint32_t> eord((size_t)c->n_ovf);
    for (size_t i = 0; i < eord.size(); ++i) eord[i] = (uint32_t)i;
    std::sort(eord.begin(), eord.end(), [&](uint32_t a, uint32_t b) { return ekey[a] < ekey[b]; });
    // slots come over in chunks so the staging buffer stays small
    const int64_t CH = 1 << 20;
    std::vector<thj_aln> buf((size_t)(nr < CH ? nr : CH));
    int64_t w = 0; size_t e = 0;
    for (int64_t r0 = 0; r0 < nr; r0 += CH) {
        int64_t n = nr - r0 < CH ? nr - r0 : CH;
        HIPCHK(hipMemcpy(buf.data(), (const thj_aln*)c->d_aln_pool + r0, (size_t)n * 128, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < n; ++k) {
            if (!cnt[(size_t)(r0 + k)]) continue;
            if (w >= c->n_alns) { thj_set_error("record count mismatch"); return THJ_ESTATE; }
            slot_to_aln(buf[(size_t)k], out[w++]);
            while (e < eord.size() && (ekey[eord[e]] >> 16) == (uint64_t)(r0 + k)) {
                if (w >= c->n_alns) { thj_set_error("record count mismatch"); return THJ_ESTATE; }
                slot_to_aln(extra[eord[e++]], out[w++]);
            }
        }
    }
    if (w != c->n_alns) { thj_set_error("record count mismatch (%lld of %lld)", (long long)w, (long long)c->n_alns); return THJ_ESTATE; }
    return THJ_OK;
}

extern "C" int thj_span_tier_counts(thj_ctx* c, int64_t* counts) {
    // reads the last batch launched handed to the closure kernels (chain entries + thj_k_stitch's list), to tier 2 (multihit reads), to tier 3 (general arrays)
    if (!c || !counts) { thj_set_error("thj_span_tier_counts: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = ensure_span_state(c);
    if (rc) return rc;
    unsigned int h[SPAN_CNT_WORDS] = {};
    HIPCHK(hipMemcpyAsync(h, &c->d_span_status[16 + SPAN_CNT_WORDS * c->span_last_set], sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    counts[0] = h[0]; counts[1] = h[1]; counts[2] = h[2]; counts[3] = h[6]; counts[4] = h[7];
    return THJ_OK;
}

extern "C" int thj_profile_span(thj_ctx* c, int enable, double* avg_ms, int64_t* launches) {
    // avg_ms[8]: thj_k_stitch_contig, thj_k_chains, thj_k_join, thj_k_join_closure, thj_k_finish, thj_k_stitch, thj_k_stitch_pack, thj_k_stitch_generic / _fusion (one set per batch launched)
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    double sum[SPK_N] = {};
    size_t n = c->span_prof_events.size() / SPK_N;
    for (size_t i = 0; i < c->span_prof_events.size(); ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->span_prof_events[i].first, c->span_prof_events[i].second));
        sum[i % SPK_N] += ms;
    }
    for (auto& pr : c->span_prof_events) { c->event_pool.push_back(pr.first); c->event_pool.push_back(pr.second); }
    if (launches) *launches = (int64_t)n;
    if (avg_ms) for (int k = 0; k < SPK_N; ++k) avg_ms[k] = n ? sum[k] / (double)n : 0.0;
    c->span_prof_events.clear();
    c->span_profile = enable != 0;
    return THJ_OK;
}

#include "thj_juncbed_impl.h"

// the runtime loads a translation unit's code object at its first launch (tens of milliseconds): thj_ctx_warm makes that happen early
__global__ void thj_k_warm_span(int* p) { if (p) *p = 0; }
void thj_warm_span(hipStream_t s) { hipLaunchKernelGGL(thj_k_warm_span, dim3(1), dim3(64), 0, s, (int*)nullptr); }
