// thj_segjuncs.hip -- gfx950 kernels + C ABI for the segment_juncs hot path.
//
// One kernel per batch of reads (one thj_segjuncs_run_async call): thj_k_segjuncs walks tiles of 256 reads --
// stage the tile's hits in LDS, drop the unspliced reads, run the mate-anchored rescue (map_read_to_contig) for
// the reads that take it, enumerate RefSeg windows and indel pairs into an LDS task queue, and execute the
// queued tasks in full rounds: one thread per task fetches the two 64-base window ends + the support read and
// does the whole motif/mismatch scan with 64-bit plane arithmetic.  Events go to HBM hash tables (set semantics
// of the reference's std::sets).
// thj_segjuncs_finish: thj_k_compact (table -> dense keys) + hipcub radix sort.
//
// No MFMA: this is integer compare / popcount work bound by HBM record streaming.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <cstddef>
#include <chrono>
#include <vector>

#include "../../include/thj.h"
#include "thj_core.h"
#include "thj_fusion_block.h"
#include "thj_internal.h"

using namespace thj;

static_assert(sizeof(thj_hit) == 16 && sizeof(Hit) == 16, "hit layout");
static_assert(sizeof(thj_params) == sizeof(Params), "params layout");



// ------------------------------------------------------------------ tables

static constexpr u64 EMPTY = ~0ull;

enum { CNT_JUNC = 0, CNT_DEL, CNT_INS, CNT_WINDOWS, CNT_INDEL_PAIRS, CNT_RESCUE_PAIRS, CNT_OVF_BLOCKS, CNT_HITS, CNT_N };

struct Tables {
    u64* junc; u64 junc_mask;
    u64* del;  u64 del_mask;
    u64* ins_key; u64* ins_val; u64 ins_mask;
    u64* junc_list; u64* del_list; u64* ins_list;      // every NEW key (junction / deletion) or slot (insertion), in arrival order
    unsigned int* ovf;              // [3] table-full flags
    unsigned long long* cnt;        // [CNT_N]
};

__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

// The position of a new key in its dense list.  The lanes of a wave that arrive here together (each just won a slot with its own key --
// whatever branch or loop iteration brought them) take their positions with ONE add on the list's counter: a batch whose reads carry
// 190 000 distinct deletions (bench.py's mix: every deletion read has its own) was 190 000 returning atomics on one address in one
// launch of thj_k_sj_tasks (the left side's launch took three times the right side's).  The order of the list does not matter (it is
// sorted when the pass ends).
__device__ __forceinline__ unsigned long long list_append_pos(unsigned long long* count) {
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (;;) {       // (one turn, unless lanes with different lists meet here: each turn serves the list of the first lane still waiting)
        const unsigned long long mine_p = (unsigned long long)count;
        const unsigned long long p0 = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mine_p >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mine_p);
        const bool now = mine_p == p0;
        const unsigned long long grp = __ballot(now);                  // the lanes executing this together, for this list
        if (now) {
            unsigned long long base = 0;
            if (lane == __ffsll((long long)grp) - 1) base = atomicAdd(count, (unsigned long long)__popcll(grp));
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));   // (the group's first lane is the first active lane here)
            return (((unsigned long long)hi << 32) | lo) + (unsigned long long)__popcll(grp & below);
        }
    }
}

// Insert into an open-addressing set; idempotent, so re-emitting an event is harmless.
// A key that was not there yet is also appended to `list` (its position is the insert counter): the distinct
// events are then already dense when the pass ends, and nobody has to scan the whole table for them.
// A table that has filled up is reported through `ovf` and grown by the host (thj_segjuncs_finish: an error for a plain run, grow-and-merge-
// again after an exchange step): once the flag is up nothing this pass inserts is kept, so an insert that sees it leaves at once, and no
// insert probes more than PROBE_CAP slots (at the <= 40 % load the host keeps the tables at, a probe sequence is a handful of slots).
// Before round 6 an insert into a FULL table walked every slot before it gave up: eight ranks' 2.5 M deletions merged into a 1 M-slot table
// (configs[2] at full size, eight contexts) were 1.5 M walks of a million slots each -- 23 s inside the exchange step.  RESCUE (the rehash of
// a growing table): the flag may still be up from the pass that filled the old table; every key must go in.
static constexpr u64 PROBE_CAP = 4096;
template <bool RESCUE = false>
__device__ __forceinline__ void set_insert(u64* tab, u64 mask, u64 key, unsigned long long* count, unsigned int* ovf, u64* list) {
    if (!RESCUE && __hip_atomic_load(ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    u64 h = mix64(key) & mask;
    const u64 limit = RESCUE ? mask : (mask < PROBE_CAP ? mask : PROBE_CAP);
    for (u64 probe = 0; probe <= limit; ++probe) {
        u64 cur = __hip_atomic_load(&tab[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == key) return;
        if (cur == EMPTY) {
            u64 old = atomicCAS((unsigned long long*)&tab[h], EMPTY, key);
            if (old == EMPTY) { const unsigned long long pos = list_append_pos(count); if (pos <= mask) list[pos] = key; return; }
            if (old == key) return;
        }
        h = (h + 1) & mask;
    }
    atomicExch(ovf, 1u);
}

template <bool RESCUE = false>
__device__ __forceinline__ void map_insert_min(u64* keys, u64* vals, u64 mask, u64 key, u64 val,
                                               unsigned long long* count, unsigned int* ovf, u64* list) {
    if (!RESCUE && __hip_atomic_load(ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    u64 h = mix64(key) & mask;
    const u64 limit = RESCUE ? mask : (mask < PROBE_CAP ? mask : PROBE_CAP);
    for (u64 probe = 0; probe <= limit; ++probe) {
        u64 cur = __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == EMPTY) {
            u64 old = atomicCAS((unsigned long long*)&keys[h], EMPTY, key);
            if (old == EMPTY) { const unsigned long long pos = list_append_pos(count); if (pos <= mask) list[pos] = h; cur = key; }
            else cur = old;
        }
        if (cur == key) { atomicMin((unsigned long long*)&vals[h], val); return; }
        h = (h + 1) & mask;
    }
    atomicExch(ovf, 1u);
}

struct EventSink {
    const Genome& g;
    const Tables& t;
    __device__ __forceinline__ void junction(uint32_t ref, uint32_t l, uint32_t r, bool a) {
        set_insert(t.junc, t.junc_mask, junc_key(g, ref, l, r, a), &t.cnt[CNT_JUNC], &t.ovf[0], t.junc_list);
    }
    __device__ __forceinline__ void deletion(uint32_t ref, uint32_t l, uint32_t r) {
        set_insert(t.del, t.del_mask, junc_key(g, ref, l, r, false), &t.cnt[CNT_DEL], &t.ovf[1], t.del_list);
    }
    __device__ __forceinline__ void insertion(uint32_t ref, uint32_t l, int len, uint32_t seq, u64 prio) {
        map_insert_min(t.ins_key, t.ins_val, t.ins_mask, ins_key(g, ref, l, len), (prio << INS_SEQ_BITS) | (u64)(seq & ((1u << INS_SEQ_BITS) - 1u)),
                       &t.cnt[CNT_INS], &t.ovf[2], t.ins_list);
    }
};

// ------------------------------------------------------------------ batch view

struct DevBatch {
    int32_t n_reads, nseg, W, pad;
    const uint32_t* seg_off;
    const Hit* hits;
    const u64* planes;
    const uint16_t* read_len;
    const uint32_t* mate_off;
    const Hit* mate_hits;
    uint32_t ordinal_base, pad2;
};
static_assert(sizeof(DevBatch) == sizeof(thj_seg_batch), "batch layout");

__device__ __forceinline__ ReadView make_view(const DevBatch& b, int r) {
    ReadView v;
    v.hits = b.hits;
    v.so = b.seg_off + (size_t)r * b.nseg;
    v.nseg = b.nseg;
    v.W = b.W;
    v.rp = b.planes + (size_t)r * 3 * b.W;
    v.rl = b.read_len[r];
    v.mate = nullptr; v.n_mate = 0; v.slots = nullptr; v.mscan = nullptr;
    if (b.mate_off) {
        uint32_t m0 = b.mate_off[r], m1 = b.mate_off[r + 1];
        v.mate = b.mate_hits + m0;
        v.n_mate = (int)(m1 - m0);
    }
    v.size = 0; v.rescue = false; v.check_len = 0;
    return v;
}

// ------------------------------------------------------------------ main kernel

static constexpr int TPB = 256;
static constexpr int QCAP = 512;           // task queue entries (LDS); a typical tile of 256 reads adds a few dozen
static constexpr int MANY_CAP = 1 << 20;   // such reads of one launch that can be listed (the rest stay with their threads)
static constexpr int MANY_HITS_LDS = 256;  // hits of such a read staged in LDS (more: read from HBM)

struct Queue {
    uint32_t* a; uint32_t* b; uint32_t* c; uint32_t* d; uint32_t* e;
    unsigned int* n;
};
// The kernels that enumerate from lists (thj_k_sj_general, thj_k_segjuncs_shared, the rescue kernels) queue their tasks in LDS and
// move the queue to ONE list in HBM at the end of every round (flush_tasks: one global add per workgroup and round, the tasks written
// side by side); thj_k_sj_tasks_list executes the list, densely, after the last of them.  None of them holds the code that executes
// a task any more (until round 4 each ran its own queue in rounds of 256 between barriers, and again un-queued when it overflowed).
// A task that finds the LDS queue full goes straight to the list (one global add of its own: rare); a full list sets `ovf` and
// thj_segjuncs_finish fails.
struct XTasks { uint4* q; uint32_t* e; unsigned int* count; unsigned int cap; unsigned int* ovf; };

struct QueueSink {
    Queue q;
    XTasks x;
    uint32_t read;             // batch index of the enumerating read
    uint32_t hbase;            // what to add to the view's hit indices to get batch hit indices
    unsigned int n_windows, n_indels;
    __device__ __forceinline__ void push(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        unsigned int k = atomicAdd(q.n, 1u);
        if (k < (unsigned)QCAP) { q.a[k] = a; q.b[k] = b; q.c[k] = c; q.d[k] = d; q.e[k] = read; }
        else {
            const unsigned int pos = atomicAdd(x.count, 1u);
            if (pos < x.cap) { x.q[pos] = make_uint4(a, b, c, d); x.e[pos] = read; } else atomicExch(x.ovf, 1u);
        }
    }
    __device__ __forceinline__ void window(uint32_t ref, int32_t wl, int32_t wr, bool anti, int start, int slen) {
        ++n_windows;
        push(task_window_word(anti, start, slen), ref, (uint32_t)wl, (uint32_t)wr);
    }
    __device__ __forceinline__ void indel(int i, uint32_t lidx, uint32_t ridx, int li, int ri, bool anti, int plen, bool is_del) {
        ++n_indels;
        push(task_indel_word(anti, is_del, i, plen), lidx + hbase, ridx + hbase, (uint32_t)li | ((uint32_t)ri << 16));
    }
};
// every thread of the workgroup, in converged code: the round's queue to the list
__device__ __forceinline__ void flush_tasks(const Queue& q, const XTasks& x, unsigned int* s_base, unsigned int* full_rounds) {
    __syncthreads();
    const unsigned int have = *q.n, n = have < (unsigned)QCAP ? have : (unsigned)QCAP;
    if (have == 0) return;
    if (threadIdx.x == 0) { *s_base = atomicAdd(x.count, n); if (have > (unsigned)QCAP) atomicAdd(full_rounds, 1u); }
    __syncthreads();
    const unsigned int base = *s_base;
    for (unsigned int k = threadIdx.x; k < n; k += blockDim.x) {
        const unsigned int pos = base + k;
        if (pos < x.cap) { x.q[pos] = make_uint4(q.a[k], q.b[k], q.c[k], q.d[k]); x.e[pos] = q.e[k]; } else atomicExch(x.ovf, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) *q.n = 0;
}

// view of a read for executing a queued task: only what window_exec / indel_exec touch
__device__ __forceinline__ ReadView make_task_view(const DevBatch& b, int r) {
    ReadView v;
    v.hits = b.hits;
    v.so = b.seg_off + (size_t)r * b.nseg;
    v.nseg = b.nseg;
    v.W = b.W;
    v.rp = b.planes + (size_t)r * 3 * b.W;
    v.rl = b.read_len[r];
    v.mate = nullptr; v.n_mate = 0; v.slots = nullptr; v.mscan = nullptr;
    v.size = 0; v.rescue = false; v.check_len = 0;
    return v;
}

// Reads that take the mate-anchored rescue (find_gaps :3330-3497) are few and their map_read_to_contig scans long,
// so the main kernel only lists them -- per workgroup, in its own slice of `list`, no global append counter -- and
// thj_k_segjuncs_rescue handles them densely afterwards.
struct RescueList { uint32_t* list; unsigned int* blk_cnt; int seg_cap; int32_t* slot_pool; unsigned int* heavy_count; uint32_t* heavy_list;
                    unsigned int* many_count; uint32_t* many_list; int own_slice; int many_min; int mid_min; };    // many_*: reads with many hits, for thj_k_segjuncs_shared; own_slice: its slice of `list`

// Stage 1 since round 4: a pipeline of dense kernels instead of one kernel that does everything for a tile of reads.
//   thj_k_sj_flat         every read.  A read with at most one hit per segment ("flat": every read of a uniquely mapping sample) is
//                         finished here -- its hits in registers, find_insertions_and_deletions and find_gaps as straight-line code
//                         (flat_read), its indel pairs and windows written as tasks to HBM; if it takes the mate-anchored rescue it is
//                         listed with one scan pair per mate hit.  Other reads are only listed: for thj_k_sj_general, or (many
//                         hits) for thj_k_segjuncs_shared.
//   thj_k_sj_general      the listed reads (a few hits a segment), one thread each, the read's hits staged in its own piece of LDS;
//                         the general enumeration, LDS task queue and execution as before.
//   thj_k_segjuncs_shared the reads with many hits, a wave each.
//   thj_k_sj_rescue_scan  one thread per (flat rescue read, mate hit): where the read's last bases lie in the mate hit's flank.
//   thj_k_sj_rescue_flat  one thread per flat rescue read: the pseudo-hits against its first-segment hit (flat_rescue) -> tasks.
//   thj_k_segjuncs_rescue(_shared)  the other rescue reads (several hits in the first segment or more than two mate hits).
//   thj_k_sj_tasks        one thread per task of the flat kernels: the two window ends + the support read, or the indel pair.
// Every list is a slice per workgroup of thj_k_sj_flat (no global append counter, the position is an LDS counter; a slice holds
// what the reads its workgroup visits can give at the most, so nothing overflows) and the consumer's workgroup w takes slice w.
struct SjLists {
    uint4* tq; uint32_t* te; unsigned int* task_cnt; int task_cap;     // tasks: words a..d, the read; task_cap entries per slice
    uint32_t* frl; unsigned int* frl_cnt;                              // flat rescue reads: read | (size - 1) << 29; seg_cap per slice
    uint32_t* pairs; unsigned int* pair_cnt;                           // their scan pairs: slot in the slice << 1 | mate hit; 2 seg_cap per slice
    int2* scan;                                                        // [(slice * seg_cap + slot) * 2 + mate hit] = rescue_scan's fwd, rev
    uint32_t* gen; unsigned int* gen_cnt;                              // reads for thj_k_sj_general: read | class << 29; seg_cap per slice
    unsigned int* mid_count; uint32_t* mid_list;                       // the second instance's reads (one list, filled by the first)
};
static constexpr int FLAT_MATES = 2;       // mate hits a flat rescue read can have (more: the general rescue kernel)
static constexpr uint32_t GEN_READ = (1u << 29) - 1;       // in the general list: the read; above it the class -- 0: at most GEN_HITS hits, 1: up to
                                                           // MID_HITS, 2: more (thj_k_segjuncs_shared)

// one task per call and lane at the most, in converged code: a ballot, one LDS add per wave, the tasks written side by side
struct FlatEmit {
    uint4* tq; uint32_t* te; unsigned int* n; unsigned long long below; uint32_t read;
    __device__ __forceinline__ void task(bool ok, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
        const unsigned long long m = __ballot(ok);
        if (m == 0) return;
        unsigned int base = 0;
        if ((threadIdx.x & 63) == 0) base = atomicAdd(n, (unsigned int)__popcll(m));
        base = (unsigned int)__shfl((int)base, 0);
        if (ok) { const unsigned int k = base + (unsigned int)__popcll(m & below); tq[k] = make_uint4(a, b, c, d); te[k] = read; }
    }
};
// position of a lane's entry in a list kept by an LDS (or global) counter: one add per wave
__device__ __forceinline__ unsigned int wave_slot(bool want, unsigned int* counter, unsigned long long below) {
    const unsigned long long m = __ballot(want);
    if (m == 0) return 0u;
    unsigned int base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, (unsigned int)__popcll(m));
    base = (unsigned int)__shfl((int)base, 0);
    return base + (unsigned int)__popcll(m & below);
}

template <int NS>
__global__ __launch_bounds__(TPB, 4) void thj_k_sj_flat(Params p, DevBatch b, RescueList rl, SjLists sl, unsigned long long* cnt) {
    __shared__ unsigned int s_n[4];          // this workgroup's tasks, flat rescue reads, scan pairs, general reads
    __shared__ unsigned int s_stat[3];
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 4) s_n[tid] = 0;
    if (tid < 3) s_stat[tid] = 0;
    __syncthreads();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const size_t slice = (size_t)blockIdx.x * rl.seg_cap;
    uint4* const tq = sl.tq + (size_t)blockIdx.x * sl.task_cap;
    uint32_t* const te = sl.te + (size_t)blockIdx.x * sl.task_cap;
    const int nseg = b.nseg;
    unsigned int my_hits = 0, my_windows = 0, my_indels = 0;
    const int n_tiles = (b.n_reads + TPB - 1) / TPB;
    // consecutive tiles go to consecutive workgroups (-> different XCDs): every XCD's L2 streams its own slices of the hit array
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int r = tile * TPB + tid;
        const bool active = r < b.n_reads;
        uint32_t so[NS + 1];
#pragma unroll
        for (int k = 0; k <= NS; ++k) so[k] = 0;
        if (active) {
            const uint32_t* o = b.seg_off + (size_t)r * nseg;
            if (NS == 4 && nseg == 4) { const uint4 q = *(const uint4*)o; so[0] = q.x; so[1] = q.y; so[2] = q.z; so[3] = q.w; so[4] = o[4]; }
            else {
#pragma unroll
                for (int k = 0; k <= NS; ++k) so[k] = o[k <= nseg ? k : nseg];
            }
        }
        const uint32_t nh = so[NS] - so[0];
        int n_mate = 0;
        if (active && b.mate_off) n_mate = (int)(b.mate_off[r + 1] - b.mate_off[r]);
        // flat: at most one hit per segment, and no more mate hits than the flat rescue takes (a read with more is rare and goes
        // the general way whether it takes the rescue or not)
        bool single = n_mate <= FLAT_MATES;
#pragma unroll
        for (int s = 0; s < NS; ++s) single = single && so[s + 1] - so[s] <= 1u;
        my_hits += nh;
        const bool is_flat = active && single && nh > 0;
        // the reads that are not flat: listed for the kernel that takes them
        {   // (a global list here would be one global atomic per wave and tile on one address: 65 k of them in a launch of the mix held
            // this kernel for 0.4 ms -- they are served one after the other, 6 ns each; thj_k_sj_general moves the reads with many
            // hits to thj_k_segjuncs_shared's list with one add per workgroup and round)
            const bool gen = active && !single;
            const unsigned int gk = wave_slot(gen, &s_n[3], below);
            if (gen) sl.gen[slice + gk] = (uint32_t)r | (nh > (uint32_t)rl.many_min ? 2u << 29 : nh > (uint32_t)rl.mid_min ? 1u << 29 : 0u);
        }
        Hit h[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            uint4 q = make_uint4(0, 0, 0, 0);
            if (is_flat && so[s + 1] != so[s]) q = ((const uint4*)b.hits)[so[s]];
            h[s].ref_id = q.x; h[s].left = (int32_t)q.y; h[s].right = (int32_t)q.z; h[s].meta = q.w;
        }
        int rlen = 0;
        if (is_flat) rlen = b.read_len[r];
        FlatEmit em{tq, te, &s_n[0], below, (uint32_t)r};
        const FlatResult res = flat_read<NS>(p, is_flat ? nseg : 0, so, h, rlen, n_mate, em);
        my_windows += (unsigned int)res.n_windows; my_indels += (unsigned int)res.n_indels;
        // the mate-anchored rescue: a read without a hit in its first segment has no pair to scan and nothing to enumerate
        const bool resc_flat = res.rescue && so[1] != so[0];
        const unsigned int fk = wave_slot(resc_flat, &s_n[1], below);
        if (resc_flat) sl.frl[slice + fk] = (uint32_t)r | ((uint32_t)(res.size - 1 < 7 ? res.size - 1 : 7) << 29);      // (only sizes 2 and 3 give windows: flat_rescue)
#pragma unroll
        for (int m = 0; m < FLAT_MATES; ++m) {
            const bool want = resc_flat && m < n_mate;
            const unsigned int pk = wave_slot(want, &s_n[2], below);
            if (want) sl.pairs[slice * FLAT_MATES + pk] = (fk << 1) | (uint32_t)m;
        }
    }
    if (my_hits) atomicAdd(&s_stat[2], my_hits);
    if (my_windows) atomicAdd(&s_stat[0], my_windows);
    if (my_indels) atomicAdd(&s_stat[1], my_indels);
    __syncthreads();
    // the listed reads with more hits than thj_k_sj_general's first instance stages go on to the second instance's list, or to
    // thj_k_segjuncs_shared's: one global add per workgroup, round and list (the entries stay in the slice; the first instance skips
    // them), so that the three kernels can start together when this one ends
    {
        __shared__ unsigned int s_nlist[2], s_lbase[2];
        const unsigned int n_gen = s_n[3];
        for (unsigned int k0 = 0; k0 < n_gen; k0 += TPB) {
            const unsigned int k = k0 + (unsigned int)tid;
            const uint32_t e = k < n_gen ? sl.gen[slice + k] : 0u;
            const bool mid = (e >> 29) == 1u, many = (e >> 29) == 2u;
            if (tid < 2) s_nlist[tid] = 0;
            __syncthreads();
            const unsigned int dk = wave_slot(mid, &s_nlist[0], below), mk = wave_slot(many, &s_nlist[1], below);
            __syncthreads();
            if (tid == 0 && s_nlist[0]) s_lbase[0] = atomicAdd(sl.mid_count, s_nlist[0]);
            if (tid == 64 && s_nlist[1]) s_lbase[1] = atomicAdd(rl.many_count, s_nlist[1]);
            __syncthreads();
            // (a full list: the read stays with the first instance, which walks its hits in HBM)
            if (mid) { if (s_lbase[0] + dk < (unsigned int)MANY_CAP) sl.mid_list[s_lbase[0] + dk] = e & GEN_READ; else sl.gen[slice + k] = e & GEN_READ; }
            if (many) { if (s_lbase[1] + mk < (unsigned int)MANY_CAP) rl.many_list[s_lbase[1] + mk] = e & GEN_READ; else sl.gen[slice + k] = e & GEN_READ; }
        }
    }
    if (tid == 0) {
        sl.task_cnt[blockIdx.x] = s_n[0]; sl.frl_cnt[blockIdx.x] = s_n[1]; sl.pair_cnt[blockIdx.x] = s_n[2]; sl.gen_cnt[blockIdx.x] = s_n[3];
        rl.blk_cnt[blockIdx.x] = 0;
        if (s_stat[0]) atomicAdd(&cnt[CNT_WINDOWS], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&cnt[CNT_INDEL_PAIRS], (unsigned long long)s_stat[1]);
        if (s_stat[2]) atomicAdd(&cnt[CNT_HITS], (unsigned long long)s_stat[2]);
    }
}

// One task: the two window ends + the support read (window_scan), or the indel pair's read piece and genome pieces.
template <bool WIDE>
__device__ __forceinline__ void exec_task(const Genome& g, const Params& p, const DevBatch& b, EventSink& ev, const uint4 q, const int tr) {
    ReadView tv = make_task_view(b, tr);
    const uint32_t a = q.x;
    const bool anti = task_anti(a);
    if (task_is_indel(a)) {
        const int i = task_indel_i(a);
        indel_exec<WIDE>(g, p, tv, i, q.y, q.z, anti, task_indel_plen(a), task_is_del(a),
                   ins_prio(b.ordinal_base + (uint32_t)tr, i, (int)(q.w & 0xFFFF), (int)(q.w >> 16)), ev);
    } else
        window_exec<WIDE>(g, p, tv, q.y, (int32_t)q.z, (int32_t)q.w, anti, task_window_start(a), task_window_slen(a), ev);
}
// The tasks of the flat kernels, one thread each: workgroup w takes slice w.  Two passes: the windows as they come, the indel pairs
// (a deletion read's: other loads, other code, a 50-step split search) set aside in LDS and run densely afterwards -- mixed in one wave
// (one lane in four on the left side of the mix) every wave ran both paths one after the other.
static constexpr int TASK_DEFER_CAP = 3072;
template <bool WIDE>
__global__ __launch_bounds__(TPB) void thj_k_sj_tasks(Genome g, Params p, DevBatch b, Tables t, SjLists sl) {
    __shared__ uint32_t s_defer[TASK_DEFER_CAP];
    __shared__ unsigned int s_nd;
    const unsigned int n = sl.task_cnt[blockIdx.x];
    const uint4* tq = sl.tq + (size_t)blockIdx.x * sl.task_cap;
    const uint32_t* te = sl.te + (size_t)blockIdx.x * sl.task_cap;
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    if (threadIdx.x == 0) s_nd = 0;
    __syncthreads();
    EventSink ev{g, t};
    for (unsigned int k0 = 0; k0 < n; k0 += TPB) {
        const unsigned int k = k0 + threadIdx.x;
        const bool active = k < n;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (active) q = tq[k];
        const bool indel = active && task_is_indel(q.x);
        const unsigned int slot = wave_slot(indel, &s_nd, below);
        if (indel && slot < (unsigned int)TASK_DEFER_CAP) s_defer[slot] = k;
        else if (active) exec_task<WIDE>(g, p, b, ev, q, (int)te[k]);          // a window -- or an indel pair beyond the room (never on the mix)
    }
    __syncthreads();
    const unsigned int nd = s_nd < (unsigned int)TASK_DEFER_CAP ? s_nd : (unsigned int)TASK_DEFER_CAP;
    for (unsigned int j = threadIdx.x; j < nd; j += TPB) {
        const unsigned int k = s_defer[j];
        const uint4 q = tq[k];
        const int tr = (int)te[k];
        ReadView tv = make_task_view(b, tr);
        const int i = task_indel_i(q.x);
        indel_exec<WIDE>(g, p, tv, i, q.y, q.z, task_anti(q.x), task_indel_plen(q.x), task_is_del(q.x),
                         ins_prio(b.ordinal_base + (uint32_t)tr, i, (int)(q.w & 0xFFFF), (int)(q.w >> 16)), ev);
    }
}
// ... of the kernels that enumerate from lists: one list.
template <bool WIDE>
__global__ __launch_bounds__(TPB) void thj_k_sj_tasks_list(Genome g, Params p, DevBatch b, Tables t, XTasks x) {
    const unsigned int n = *x.count < x.cap ? *x.count : x.cap;
    EventSink ev{g, t};
    for (unsigned int k = blockIdx.x * TPB + threadIdx.x; k < n; k += gridDim.x * TPB) exec_task<WIDE>(g, p, b, ev, x.q[k], (int)x.e[k]);
}

// One thread per (flat rescue read, mate hit): rescue_scan, whatever the left hit is (it is the same scan for every left hit;
// thj_k_sj_rescue_flat applies the left hit's contig / strand test).
__global__ __launch_bounds__(TPB) void thj_k_sj_rescue_scan(Genome g, Params p, DevBatch b, SjLists sl, int seg_cap) {
    const unsigned int n = sl.pair_cnt[blockIdx.x];
    const size_t slice = (size_t)blockIdx.x * seg_cap;
    for (unsigned int k = threadIdx.x; k < n; k += TPB) {
        const uint32_t e = sl.pairs[slice * FLAT_MATES + k];
        const uint32_t slot = e >> 1, m = e & 1u;
        const int r = (int)(sl.frl[slice + slot] & 0x1FFFFFFFu);
        const uint4 q = ((const uint4*)b.mate_hits)[b.mate_off[r] + m];
        Hit rh; rh.ref_id = q.x; rh.left = (int32_t)q.y; rh.right = (int32_t)q.z; rh.meta = q.w;
        int32_t f, rv;
        const bool scanned = rescue_scan(g, p, b.planes + (size_t)r * 3 * b.W, b.W, (int)b.read_len[r], rh, f, rv);
        if (!scanned && f != SLOT_BREAK) f = SLOT_UNSCANNED;
        sl.scan[(slice + slot) * FLAT_MATES + m] = make_int2(f, rv);
    }
}

// One thread per flat rescue read: its first-segment hit against the pseudo-hits of its (at most two) mate hits -> window tasks,
// appended to the slice thj_k_sj_flat's workgroup w filled (thj_k_sj_tasks runs after this kernel).
__global__ __launch_bounds__(TPB) void thj_k_sj_rescue_flat(Params p, DevBatch b, SjLists sl, int seg_cap, unsigned long long* cnt, unsigned int* n_flat_pairs) {
    __shared__ unsigned int s_ntask, s_stat[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned int n = sl.frl_cnt[blockIdx.x];
    if (tid == 0) { s_ntask = sl.task_cnt[blockIdx.x]; s_stat[0] = 0; s_stat[1] = 0; }
    __syncthreads();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const size_t slice = (size_t)blockIdx.x * seg_cap;
    unsigned int my_pairs = 0, my_windows = 0;
    for (unsigned int k0 = 0; k0 < n; k0 += TPB) {
        const unsigned int k = k0 + (unsigned int)tid;
        const bool active = k < n;
        int r = 0, size = 0, n_mate = 0, rlen = 0;
        Hit bh{0, 0, 0, 0}, mh[FLAT_MATES];
        int32_t sc[2 * FLAT_MATES];
#pragma unroll
        for (int m = 0; m < FLAT_MATES; ++m) { mh[m] = Hit{0, 0, 0, 0}; sc[2 * m] = SLOT_NONE; sc[2 * m + 1] = SLOT_NONE; }
        if (active) {
            const uint32_t e = sl.frl[slice + k];
            r = (int)(e & 0x1FFFFFFFu); size = (int)(e >> 29) + 1;
            const uint4 q = ((const uint4*)b.hits)[b.seg_off[(size_t)r * b.nseg]];
            bh.ref_id = q.x; bh.left = (int32_t)q.y; bh.right = (int32_t)q.z; bh.meta = q.w;
            rlen = b.read_len[r];
            const uint32_t m0 = b.mate_off[r];
            n_mate = (int)(b.mate_off[r + 1] - m0);
#pragma unroll
            for (int m = 0; m < FLAT_MATES; ++m)
                if (m < n_mate) {
                    const uint4 w = ((const uint4*)b.mate_hits)[m0 + m];
                    mh[m].ref_id = w.x; mh[m].left = (int32_t)w.y; mh[m].right = (int32_t)w.z; mh[m].meta = w.w;
                    const int2 o = sl.scan[(slice + k) * FLAT_MATES + m];
                    sc[2 * m] = o.x; sc[2 * m + 1] = o.y;
                }
        }
        FlatEmit em{sl.tq + (size_t)blockIdx.x * sl.task_cap, sl.te + (size_t)blockIdx.x * sl.task_cap, &s_ntask, below, (uint32_t)r};
        int nw = 0;
        my_pairs += (unsigned int)flat_rescue<FLAT_MATES>(p, active, bh, size, rlen, mh, n_mate, sc, em, nw);
        my_windows += (unsigned int)nw;
    }
    if (my_pairs) atomicAdd(&s_stat[0], my_pairs);
    if (my_windows) atomicAdd(&s_stat[1], my_windows);
    __syncthreads();
    if (tid == 0) {
        sl.task_cnt[blockIdx.x] = s_ntask;
        if (s_stat[0]) { atomicAdd(&cnt[CNT_RESCUE_PAIRS], (unsigned long long)s_stat[0]); atomicAdd(n_flat_pairs, s_stat[0]); }
        if (s_stat[1]) atomicAdd(&cnt[CNT_WINDOWS], (unsigned long long)s_stat[1]);
    }
}

// The listed reads that are not flat: one thread per read, the read's offsets and hits copied to the thread's own piece of LDS with
// independent loads, so that the list walks of the general enumeration (dozens of dependent round trips from HBM) run on LDS.
// Two instances: SLICED (256 threads, reads of at most GEN_HITS hits) takes the slices thj_k_sj_flat filled and hands the reads
// with more hits on -- up to MID_HITS to the second instance's list (64 threads a workgroup, the same thing with more LDS per
// thread: a read with eight hits a segment costs a thread 40 us of LDS walks, but 64 of them run side by side; given a wave each
// in thj_k_segjuncs_shared they cost 25 us apiece, most of it the wave's own chain of global loads), beyond that to
// thj_k_segjuncs_shared's, one global add per workgroup, round and list.  Rescue reads go on to thj_k_segjuncs_rescue; windows and
// indel pairs are queued in LDS and executed in whole rounds as before.
#ifndef THJ_GEN_HITS
#define THJ_GEN_HITS 8
#endif
static constexpr int GEN_HITS = THJ_GEN_HITS;        // hits of a read the first instance stages
static constexpr int MID_HITS = 32;        // ... the second (more: thj_k_segjuncs_shared, or rl.many_min if that is smaller)
static constexpr int MID_T = 256;
static constexpr int MID_G = 8;            // lanes a read of the second instance's list is shared by
static constexpr int MID_GRID = 2048;
#ifndef THJ_GEN_G
#define THJ_GEN_G 4
#endif
static constexpr int GEN_G = THJ_GEN_G;    // ... of the first instance's
template <int HITS, int T, bool SLICED, int SO, int G = 1>
__global__ __launch_bounds__(T) void thj_k_sj_general(Params p, DevBatch b, RescueList rl, SjLists sl, XTasks x, unsigned long long* cnt) {
    // G: lanes that share a read (a power of two <= 64, adjacent lanes of one wave).  1: a thread per read.  More -- the second instance,
    // reads of three to eight hits a segment --: lane g of the group takes every G-th hit of each sweep (gaps_prepare_shared,
    // indels_enumerate / gaps_enumerate with (first, stride)), so that a read's sweeps of k x k hit pairs are k steps long, not k x k:
    // with a thread per read the find_gaps sweeps were 0.18 of this instance's 0.26 ms (THJ_EXP build, flag 1 << 18).
    static_assert(G >= 1 && G <= 64 && (G & (G - 1)) == 0 && T % G == 0, "lanes per read");
    constexpr int RPR = T / G;             // reads per round
    constexpr int STRIDE = HITS + 1;       // uint4 per read (an odd count: the threads of a wave spread over the banks)
    __shared__ uint32_t q_a[QCAP], q_b[QCAP], q_c[QCAP], q_d[QCAP], q_e[QCAP];
    __shared__ uint4 s_hits[RPR * STRIDE];
    __shared__ uint32_t s_so[RPR * SO];                     // SO: words of a read's CSR row (nseg + 1 <= 9, or <= 17 for reads of more than eight segments)
    __shared__ unsigned int q_n, s_nresc, s_base[3], s_xbase;
    __shared__ unsigned int s_stat[4];
    const int tid = threadIdx.x, lane = tid & 63, slot = tid / G, g = tid % G;
    if (tid < 4) s_stat[tid] = 0;
    if (tid == 0) { q_n = 0; s_nresc = 0; }
    __syncthreads();
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const Queue qq{q_a, q_b, q_c, q_d, q_e, &q_n};
    const unsigned int n = SLICED ? sl.gen_cnt[blockIdx.x] : (*sl.mid_count < (unsigned int)MANY_CAP ? *sl.mid_count : (unsigned int)MANY_CAP);
    const uint32_t* list = SLICED ? sl.gen + (size_t)blockIdx.x * rl.seg_cap : sl.mid_list;
    const unsigned int first = SLICED ? 0u : blockIdx.x * RPR, step = SLICED ? (unsigned int)RPR : gridDim.x * RPR;
    unsigned int my_windows = 0, my_indels = 0;
    for (unsigned int k0 = first; k0 < n; k0 += step) {
        __syncthreads();
        const unsigned int k = k0 + (unsigned int)slot;
        bool active = k < n;
        ReadView v;
        bool do_gaps = false, to_rescue = false;
        int r = 0;
        uint32_t hbase = 0;
        const uint32_t e = active ? list[k] : 0u;
        r = (int)(e & GEN_READ);
        if (SLICED && (e >> 29) != 0u) active = false;          // thj_k_sj_flat moved it to another kernel's list
        if (active) {
            v = make_view(b, r);
            const uint32_t h0 = v.so[0], nh = v.so[v.nseg] - h0;
            if (nh <= (uint32_t)HITS) {              // (a read a full list left here has more: it walks its hits in HBM)
                for (int i = g; i <= v.nseg; i += G) s_so[slot * SO + i] = v.so[i] - h0;
                for (uint32_t i = (uint32_t)g; i < nh; i += (uint32_t)G) s_hits[slot * STRIDE + i] = ((const uint4*)b.hits)[h0 + i];
                v.so = s_so + slot * SO; v.hits = (const Hit*)(s_hits + slot * STRIDE); hbase = h0;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the copy is read back as Hit records
        if (active) {
            bool wants = false;
            if (G == 1) do_gaps = !THJ_EXPF(1 << 18) && gaps_prepare(p, v, wants);
            else do_gaps = !THJ_EXPF(1 << 18) && gaps_prepare_shared(p, v, wants, g, G, [&](bool f) { return ((__ballot(f) >> (lane - g)) & (~0ull >> (64 - G))) != 0ull; });
            to_rescue = do_gaps && wants;
            if (!to_rescue) {
                QueueSink qs{qq, x, (uint32_t)r, hbase, 0u, 0u};
                if (!THJ_EXPF(1 << 17)) indels_enumerate(p, v, qs, g, G);
                if (do_gaps) gaps_enumerate(p, v, qs, g, G);
                my_windows += qs.n_windows; my_indels += qs.n_indels;
            }
        }
        const bool resc = to_rescue && g == 0;                  // (one lane lists the read)
        if (SLICED) {        // rescue reads: the slice of the same number (thj_k_sj_flat left it empty)
            const unsigned int rk = wave_slot(resc, &s_nresc, below);
            if (resc) rl.list[(size_t)blockIdx.x * rl.seg_cap + rk] = (uint32_t)r;
        } else {             // ... thj_k_segjuncs_shared's slice, a range of it per round
            if (tid == 0) s_nresc = 0;
            __syncthreads();
            const unsigned int rk = wave_slot(resc, &s_nresc, below);
            __syncthreads();
            if (tid == 0 && s_nresc) s_base[2] = atomicAdd(&rl.blk_cnt[rl.own_slice], s_nresc);
            __syncthreads();
            if (resc) rl.list[(size_t)rl.own_slice * rl.seg_cap + s_base[2] + rk] = (uint32_t)r;
        }
        flush_tasks(qq, x, &s_xbase, &s_stat[3]);
    }
    if (my_windows) atomicAdd(&s_stat[0], my_windows);
    if (my_indels) atomicAdd(&s_stat[1], my_indels);
    __syncthreads();
    if (tid == 0) {
        if (SLICED) rl.blk_cnt[blockIdx.x] = s_nresc;
        if (s_stat[0]) atomicAdd(&cnt[CNT_WINDOWS], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&cnt[CNT_INDEL_PAIRS], (unsigned long long)s_stat[1]);
        if (s_stat[3]) atomicAdd(&cnt[CNT_OVF_BLOCKS], (unsigned long long)s_stat[3]);
    }
}

struct SjWave {            // the wave operations wave_read_enumerate is written against
    int lane;
    __device__ __forceinline__ unsigned long long ballot(bool p) { return __ballot(p); }
    __device__ __forceinline__ uint32_t bcast(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane(src)); }
};
// The reads with many hits (listed by the main kernel, unclassified): one WAVE per read.  The read's hits are staged in LDS, the partner
// search of find_gaps' head is shared over the lanes; a read that takes the mate-anchored rescue goes to this kernel's own slice of
// the rescue list (the rescue kernels run next), the others are enumerated here, lane = hit, into the usual queue.
__global__ __launch_bounds__(TPB, 4) void thj_k_segjuncs_shared(Params p, DevBatch b, RescueList rl, XTasks x, unsigned long long* cnt) {
    __shared__ uint32_t q_a[QCAP], q_b[QCAP], q_c[QCAP], q_d[QCAP], q_e[QCAP];
    __shared__ unsigned int q_n;
    __shared__ unsigned int s_stat[4];
    __shared__ Hit s_h[TPB / 64][MANY_HITS_LDS];
    __shared__ uint32_t s_o[TPB / 64][20];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 4) s_stat[tid] = 0;
    if (tid == 0) q_n = 0;
    __syncthreads();
    const Queue qq{q_a, q_b, q_c, q_d, q_e, &q_n};
    const unsigned int n = *rl.many_count < (unsigned int)MANY_CAP ? *rl.many_count : (unsigned int)MANY_CAP;
    // A workgroup takes SHARED_BATCH reads at a time and its waves draw them one by one (s_next): a read with forty hits a segment
    // costs a hundred times one with four, and with one read per wave and a barrier per read three waves in four waited for it
    // (0.64 ms per launch for 65 k reads, of which the barrier-bound waiting was most).
    const unsigned int SHARED_BATCH = n / gridDim.x >= 32u ? 32u : n / gridDim.x >= 4u ? n / gridDim.x : 4u;      // a few reads per wave and turn; small launches spread over the chip
    __shared__ unsigned int s_next, s_batch, s_nr, s_rbase, s_xbase;
    __shared__ uint32_t s_resc[32];                         // the rescue reads of a batch (SHARED_BATCH <= 32)
    if (tid == 0) s_nr = 0;
    unsigned int my_windows = 0, my_indels = 0;
    for (;;) {                                              // ... and the workgroups draw the batches (the word after the list's count)
        __syncthreads();
        if (tid == 0) { s_next = 0; s_batch = atomicAdd(rl.many_count + 1, 1u); }
        __syncthreads();
        const unsigned int base = s_batch * SHARED_BATCH;
        if (base >= n) break;
        for (;;) {
            unsigned int k = 0;
            if (lane == 0) k = atomicAdd(&s_next, 1u);
            k = (unsigned int)__shfl((int)k, 0);
            if (k >= SHARED_BATCH || base + k >= n) break;
            const int r = (int)rl.many_list[base + k];
            ReadView v = make_view(b, r);
            bool do_gaps = false, wants = false, staged = false;
            uint32_t hbase = 0;
            const uint32_t h0 = v.so[0], nh = v.so[v.nseg] - h0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the wave's last read is done with s_h
            if (nh <= (uint32_t)MANY_HITS_LDS && v.nseg < 20) {                 // the hits and their offsets into LDS
                for (uint32_t i = (uint32_t)lane; i < nh; i += 64u) ((uint4*)s_h[wave])[i] = ((const uint4*)b.hits)[h0 + i];
                if (lane <= v.nseg) s_o[wave][lane] = v.so[lane] - h0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                v.hits = s_h[wave]; v.so = s_o[wave]; hbase = h0; staged = true;
            }
            if (THJ_EXPF(1 << 26)) continue;                    // (ablation: the draw and the staging alone)
            if (staged && wave_read_fits(v)) {
                // the hits are staged and no segment holds more than 64: the sweeps run on registers (wave_read_enumerate)
                SjWave xw{lane};
                QueueSink qs{qq, x, (uint32_t)r, hbase, 0u, 0u};
                do_gaps = wave_read_enumerate(xw, p, v, qs, wants);
                if (do_gaps && wants) { if (lane == 0) s_resc[atomicAdd(&s_nr, 1u)] = (uint32_t)r; }
                else { my_windows += qs.n_windows; my_indels += qs.n_indels; }
                continue;
            }
            do_gaps = gaps_prepare_shared(p, v, wants, lane, 64, [](bool f) { return __any((int)f) != 0; });
            if (THJ_EXPF(1 << 27)) continue;                    // (... and the partner search)
            if (do_gaps && wants) {
                // the rescue kernels take it from here (their list, this kernel's slice)
                if (lane == 0) s_resc[atomicAdd(&s_nr, 1u)] = (uint32_t)r;
            } else {
                QueueSink qs{qq, x, (uint32_t)r, hbase, 0u, 0u};
                if (!THJ_EXPF(1 << 17)) indels_enumerate(p, v, qs, lane, 64);
                if (do_gaps && !THJ_EXPF(1 << 25)) gaps_enumerate(p, v, qs, lane, 64);
                my_windows += qs.n_windows; my_indels += qs.n_indels;
            }
        }
        __syncthreads();
        if (s_nr) {                                          // the batch's rescue reads: one global add for all of them
            if (tid == 0) s_rbase = atomicAdd(&rl.blk_cnt[rl.own_slice], s_nr);
            __syncthreads();
            if ((unsigned int)tid < s_nr) rl.list[(size_t)rl.own_slice * rl.seg_cap + s_rbase + tid] = s_resc[tid];
            __syncthreads();
            if (tid == 0) s_nr = 0;
        }
        flush_tasks(qq, x, &s_xbase, &s_stat[3]);
    }
    if (my_windows) atomicAdd(&s_stat[0], my_windows);
    if (my_indels) atomicAdd(&s_stat[1], my_indels);
    __syncthreads();
    if (tid == 0) {
        if (s_stat[0]) atomicAdd(&cnt[CNT_WINDOWS], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&cnt[CNT_INDEL_PAIRS], (unsigned long long)s_stat[1]);
        if (s_stat[3]) atomicAdd(&cnt[CNT_OVF_BLOCKS], (unsigned long long)s_stat[3]);
    }
}

// Rescue kernel: one thread per listed read, all lanes scanning.  The read's (hit, mate hit) pairs are mapped
// (map_read_to_contig over the mate flank, rescue_pair) into a small per-thread LDS slot area -- reads with more
// pairs than fit recompute them as rv_foreach walks the pseudo-hit list -- then the general enumeration runs with the
// rescued hits in place and its windows are queued and executed as in the main kernel.
static constexpr int RPT = 4;             // rescue pairs per thread kept in LDS
static constexpr int MSCAN = 128;          // mate hits of a read with more pairs that a wave keeps the scans of (thj_k_segjuncs_rescue_shared)
static constexpr int RESCUE_GRID = 1024;  // workgroups of the rescue kernel at the most
static constexpr int HEAVY_CAP = 1 << 18;  // rescue reads of one launch that can have a pool slice (the rest recompute their pairs as they go)
static constexpr int MAX_LISTS = 2048;    // workgroups of the main kernel = slices of the rescue list

__global__ __launch_bounds__(TPB, 4) void thj_k_segjuncs_rescue(Genome g, Params p, DevBatch b, RescueList rl, int n_lists, XTasks x, unsigned long long* cnt) {
    __shared__ uint32_t q_a[QCAP], q_b[QCAP], q_c[QCAP], q_d[QCAP], q_e[QCAP];
    __shared__ int32_t s_slots[TPB * RPT * 2];
    __shared__ unsigned int s_off[MAX_LISTS + 1];
    __shared__ unsigned int q_n;
    __shared__ unsigned int s_stat[4];
    __shared__ Genome s_g;            // copies the on-the-fly rescue can point at (kernel arguments have no address)
    __shared__ Params s_p;
    typedef hipcub::BlockScan<unsigned int, TPB> Scan;
    __shared__ typename Scan::TempStorage scan_tmp;
    __shared__ unsigned int s_nheavy, s_heavy_base, s_xbase;
    const int tid = threadIdx.x;
    const unsigned long long below = (tid & 63) ? (~0ull >> (64 - (tid & 63))) : 0ull;
    if (tid < 4) s_stat[tid] = 0;
    if (tid == 0) { q_n = 0; s_g = g; s_p = p; }
    // offsets of the per-workgroup slices in their concatenation
    unsigned int total;
    {
        constexpr int IPT = MAX_LISTS / TPB;
        unsigned int cnt[IPT], sum = 0, excl;
#pragma unroll
        for (int k = 0; k < IPT; ++k) { const int j = tid * IPT + k; cnt[k] = j < n_lists ? rl.blk_cnt[j] : 0u; sum += cnt[k]; }
        Scan(scan_tmp).ExclusiveSum(sum, excl, total);
#pragma unroll
        for (int k = 0; k < IPT; ++k) { s_off[tid * IPT + k] = excl; excl += cnt[k]; }
        if (tid == 0) s_off[MAX_LISTS] = total;
    }
    __syncthreads();
    const Queue qq{q_a, q_b, q_c, q_d, q_e, &q_n};
    const unsigned int per_round = gridDim.x * TPB;
    const unsigned int rounds = (total + per_round - 1) / per_round;
    unsigned int my_pairs = 0, my_windows = 0, my_indels = 0;      // statistics: per thread, added to LDS once at the end
    for (unsigned int it = 0; it < rounds; ++it) {
        const unsigned int i = (it * gridDim.x + blockIdx.x) * TPB + tid;
        const bool active = i < total;
        __syncthreads();
        ReadView v;
        bool do_gaps = false, heavy = false;
        int r = 0;
        if (active) {
            int lo = 0, hi = n_lists;                    // slice holding entry i: last one with s_off <= i
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
            r = (int)rl.list[(size_t)lo * rl.seg_cap + (i - s_off[lo])];
            v = make_view(b, r);
            // a read with more pairs than the LDS slots hold (multihits) only has its pairs' outcomes computed here, into its slice of
            // the HBM pool; thj_k_segjuncs_rescue_shared enumerates it with a wave (one thread walking 40 x 40 pairs held its tile up)
            heavy = (int64_t)rv_count_raw(v, 0) * v.n_mate > RPT && v.n_mate <= MSCAN && !THJ_EXPF(1 << 24);
        }
        unsigned int hk = 0;
        {   // positions in the heavy list: one global add per workgroup and round (per read they queued on one address)
            if (tid == 0) s_nheavy = 0;
            __syncthreads();
            hk = wave_slot(heavy, &s_nheavy, below);
            __syncthreads();
            if (tid == 0 && s_nheavy) s_heavy_base = atomicAdd(rl.heavy_count, s_nheavy);
            __syncthreads();
            hk += s_heavy_base;
            heavy = heavy && hk < (unsigned int)HEAVY_CAP;
        }
        if (active) {
            QueueSink qs{qq, x, (uint32_t)r, 0u, 0u, 0u};
            if (heavy) rl.heavy_list[hk] = (uint32_t)r;
            if (!heavy) indels_enumerate(p, v, qs);
            // every listed read takes the rescue (that is why it was listed): the partner search -- 40 x 40 dependent loads for one
            // thread when the read came from thj_k_segjuncs_shared, 0.4 ms of this kernel on the mix -- is not repeated
            const bool wants = true;
            gaps_prepare_listed(p, v);
            do_gaps = true;
            if (do_gaps && !heavy) {
                if (wants) {
                    const int n_left = rv_count_raw(v, 0);
                    // the pairs' outcomes are kept (LDS, or HBM for a read with many hits): gaps_enumerate asks for them once per
                    // (hit, partner) it looks at, and recomputing a flank scan each time made a 16-copy read cost 200 of them
                    const bool fits = (int64_t)n_left * v.n_mate <= RPT;          // (more, and more mate hits than a wave's table holds: computed as they are asked for)
                    int32_t* mine = s_slots + tid * RPT * 2;
                    unsigned int local = 0;
                    // the scan of a mate hit's flank is the same for every left hit on the mate's contig and opposite strand: kept
                    // for the first two mate hits (a read of a repeat family has tens of left hits and one or two mate hits)
                    int32_t c0_f = SLOT_NONE, c0_rv = SLOT_NONE, c1_f = SLOT_NONE, c1_rv = SLOT_NONE; bool c0_ok = false, c1_ok = false; unsigned int sc_have = 0;
                    for (int l = 0; l < n_left; ++l)
                        for (int m = 0; m < v.n_mate; ++m) {
                            int32_t f = SLOT_NONE, rv = SLOT_NONE;
                            if (THJ_EXPF(1 << 22)) { }
                            else {
                                const Hit lh = v.hits[v.so[0] + l], rh = v.mate[m];
                                if (lh.ref_id == rh.ref_id && hit_anti(lh) != hit_anti(rh)) {        // :3414
                                    bool scanned;
                                    if (m == 0) {
                                        if (!(sc_have & 1u)) { c0_ok = rescue_scan(g, p, v.rp, v.W, v.rl, rh, c0_f, c0_rv); sc_have |= 1u; }
                                        f = c0_f; rv = c0_rv; scanned = c0_ok;
                                    } else if (m == 1) {
                                        if (!(sc_have & 2u)) { c1_ok = rescue_scan(g, p, v.rp, v.W, v.rl, rh, c1_f, c1_rv); sc_have |= 2u; }
                                        f = c1_f; rv = c1_rv; scanned = c1_ok;
                                    } else scanned = rescue_scan(g, p, v.rp, v.W, v.rl, rh, f, rv);
                                    if (scanned) ++local;
                                }
                            }
                            if (fits) { mine[2 * (l * v.n_mate + m)] = f; mine[2 * (l * v.n_mate + m) + 1] = rv; }
                            if (f == SLOT_BREAK) break;                  // the reference leaves the mate loop here (:3431-3450)
                        }
                    my_pairs += local;
                    v.rescue = true;
                    v.slots = fits ? mine : nullptr;
                    v.lazy_g = &s_g; v.lazy_p = &s_p;
                }
                if (!THJ_EXPF(1 << 23)) gaps_enumerate(p, v, qs);
            }
            my_windows += qs.n_windows; my_indels += qs.n_indels;
        }
        flush_tasks(qq, x, &s_xbase, &s_stat[3]);
    }
    if (my_pairs) atomicAdd(&s_stat[2], my_pairs);
    if (my_windows) atomicAdd(&s_stat[0], my_windows);
    if (my_indels) atomicAdd(&s_stat[1], my_indels);
    __syncthreads();
    if (tid == 0) {
        if (s_stat[0]) atomicAdd(&cnt[CNT_WINDOWS], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&cnt[CNT_INDEL_PAIRS], (unsigned long long)s_stat[1]);
        if (s_stat[2]) atomicAdd(&cnt[CNT_RESCUE_PAIRS], (unsigned long long)s_stat[2]);
        if (s_stat[3]) atomicAdd(&cnt[CNT_OVF_BLOCKS], (unsigned long long)s_stat[3]);
    }
}

// The rescue reads with many pairs (listed by thj_k_segjuncs_rescue): one wave per read.  Round 6: the read's hits, its mate hits,
// the scan of every mate hit's flank and the pseudo-hit list itself live in the wave's piece of LDS -- a lane per mate hit scans, a
// lane per left hit walks the mate hits once and writes the pseudo-hits it contributes at its prefix-sum position (the reference's
// order, rescue_pseudo_hits), then lane = left hit enumerates against that list.  Before, every look at the list (the bowtie2
// count, the three walks of gaps_enumerate per left hit) re-derived it pair by pair from hit records in HBM: k x k dependent loads
// a look, k^3 for a read of k copies whose mate has k hits -- 1.0 ms per launch on the e2e files, whatever the number of reads.
static constexpr int PLIST_CAP = 128;      // pseudo-hits of a read kept in LDS (bowtie2 runs drop a read with more than max_seg_multihits = 40 of them)
__global__ __launch_bounds__(TPB, 3) void thj_k_segjuncs_rescue_shared(Genome g, Params p, DevBatch b, RescueList rl, XTasks x, unsigned long long* cnt) {
    __shared__ uint32_t q_a[QCAP], q_b[QCAP], q_c[QCAP], q_d[QCAP], q_e[QCAP];
    __shared__ unsigned int q_n;
    __shared__ unsigned int s_stat[4];
    __shared__ Genome s_g;
    __shared__ Params s_p;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < 4) s_stat[tid] = 0;
    if (tid == 0) { q_n = 0; s_g = g; s_p = p; }
    __syncthreads();
    const Queue qq{q_a, q_b, q_c, q_d, q_e, &q_n};
    const unsigned int n = *rl.heavy_count < (unsigned int)HEAVY_CAP ? *rl.heavy_count : (unsigned int)HEAVY_CAP;
    // the waves of a workgroup draw the reads of its batch one by one, as in thj_k_segjuncs_shared
    const unsigned int BATCH = n / gridDim.x >= 32u ? 32u : n / gridDim.x >= 4u ? n / gridDim.x : 4u;
    __shared__ unsigned int s_next, s_xbase;
    __shared__ int32_t s_mscan[TPB / 64][2 * MSCAN];
    __shared__ Hit s_h[TPB / 64][MANY_HITS_LDS];
    __shared__ uint32_t s_o[TPB / 64][20];
    __shared__ Hit s_mate[TPB / 64][MSCAN];
    __shared__ PHit s_pl[TPB / 64][PLIST_CAP];
    const int wave = tid >> 6;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    unsigned int my_windows = 0, my_indels = 0, my_pairs = 0;
    for (unsigned int base = blockIdx.x * BATCH; base < n; base += gridDim.x * BATCH) {
        __syncthreads();
        if (tid == 0) s_next = 0;
        __syncthreads();
        for (;;) {
            unsigned int k = 0;
            if (lane == 0) k = atomicAdd(&s_next, 1u);
            k = (unsigned int)__shfl((int)k, 0);
            if (k >= BATCH || base + k >= n) break;
            const unsigned int h = base + k;
            const int r = (int)rl.heavy_list[h];
            ReadView v = make_view(b, r);
            gaps_prepare_listed(p, v);                             // a listed read takes the rescue: no second partner search
            uint32_t hbase = 0;
            const uint32_t h0 = v.so[0], nh = v.so[v.nseg] - h0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the wave's last read is done with its tables
            if (nh <= (uint32_t)MANY_HITS_LDS && v.nseg < 20) {   // the read's hits and their offsets
                for (uint32_t i = (uint32_t)lane; i < nh; i += 64u) ((uint4*)s_h[wave])[i] = ((const uint4*)b.hits)[h0 + i];
                if (lane <= v.nseg) s_o[wave][lane] = v.so[lane] - h0;
                v.hits = s_h[wave]; v.so = s_o[wave]; hbase = h0;
            }
            // where the read's last bases lie in each mate hit's flank: one scan per mate hit, a lane each -- the scan does not look at
            // the left hit, so the n_left x n_mate pairs of the reference's loop (:3406-3492) are n_mate scans and a contig / strand test
            // per pair (until round 4 a read with more than 64 pairs scanned again for every pair its enumeration looked at: 0.3 s per
            // launch of thj_k_segjuncs_rescue on files with a 41-copy family whose mates carry 82 hits)
            int32_t* ms = s_mscan[wave];
            for (int m = lane; m < v.n_mate; m += 64) {           // (n_mate <= MSCAN: thj_k_segjuncs_rescue lists no other read)
                const uint4 q = ((const uint4*)v.mate)[m];
                Hit rh; rh.ref_id = q.x; rh.left = (int32_t)q.y; rh.right = (int32_t)q.z; rh.meta = q.w;
                s_mate[wave][m] = rh;
                int32_t f, rv;
                const bool scanned = rescue_scan(g, p, v.rp, v.W, v.rl, rh, f, rv);
                if (!scanned && f != SLOT_BREAK) f = SLOT_UNSCANNED;
                ms[2 * m] = f; ms[2 * m + 1] = rv;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v.mate = s_mate[wave];
            // the pseudo-hit list: lane = left hit, rounds of 64 left hits; a lane's hits land behind those of the left hits before it
            const int n_left = rv_count_raw(v, 0);
            int n_pl = 0;
            bool listed = true;
            for (int l0 = 0; l0 < n_left; l0 += 64) {
                const int l = l0 + lane;
                int mine = 0, sc = 0;
                Hit lh{0, 0, 0, 0};
                if (l < n_left) { lh = v.hits[v.so[0] + l]; mine = rescue_pseudo_hits(lh, v.mate, v.n_mate, ms, nullptr, sc); }
                my_pairs += (unsigned int)sc;
                int incl = mine;                                   // inclusive prefix sum over the wave
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
                const int total = __shfl(incl, 63);
                const int at = n_pl + incl - mine;
                if (n_pl + total <= PLIST_CAP) { if (mine) { int dummy = 0; rescue_pseudo_hits(lh, v.mate, v.n_mate, ms, s_pl[wave] + at, dummy); } }
                else listed = false;
                n_pl += total;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v.rescue = true; v.slots = nullptr; v.lazy_g = &s_g; v.lazy_p = &s_p;
            QueueSink qs{qq, x, (uint32_t)r, hbase, 0u, 0u};
            indels_enumerate(p, v, qs, lane, 64);
            if (listed) { v.plist = s_pl[wave]; v.n_plist = n_pl; v.mscan = nullptr; gaps_enumerate(p, v, qs, lane, 64); }
            else if (p.bowtie2 && n_pl > p.max_seg_multihits) { }  // find_gaps returns at its multihit test (:3499-3506) whatever else the read holds
            else { v.mscan = ms; gaps_enumerate(p, v, qs, lane, 64); }      // a list beyond the LDS piece (bowtie1, or a raised --max-seg-multihits): the pairs walked as before
            my_windows += qs.n_windows; my_indels += qs.n_indels;
        }
        flush_tasks(qq, x, &s_xbase, &s_stat[3]);
    }
    (void)below;
    if (my_windows) atomicAdd(&s_stat[0], my_windows);
    if (my_indels) atomicAdd(&s_stat[1], my_indels);
    if (my_pairs) atomicAdd(&s_stat[2], my_pairs);
    __syncthreads();
    if (tid == 0) {
        if (s_stat[0]) atomicAdd(&cnt[CNT_WINDOWS], (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&cnt[CNT_INDEL_PAIRS], (unsigned long long)s_stat[1]);
        if (s_stat[2]) atomicAdd(&cnt[CNT_RESCUE_PAIRS], (unsigned long long)s_stat[2]);
        if (s_stat[3]) atomicAdd(&cnt[CNT_OVF_BLOCKS], (unsigned long long)s_stat[3]);
    }
}

// ------------------------------------------------------------------ fusion search

struct FusionSink {
    thj_fusion* buf; unsigned long long* count; unsigned long long cap; unsigned int* ovf;
    const uint8_t* ignore; uint32_t n_ignore;             // per ref id: 1 = --fusion-ignore-chromosomes names it
    __device__ __forceinline__ bool ignored(uint32_t ref) const { return ignore && ref < n_ignore && ignore[ref]; }
    __device__ __forceinline__ void fusion(uint32_t r1, uint32_t r2, uint32_t l, uint32_t r, uint32_t dir, uint32_t ed) {
        unsigned long long pos = atomicAdd(count, 1ull);
        if (pos < cap) { thj_fusion f{r1, r2, l, r, dir, 1u, ed, 0u}; buf[pos] = f; }
        else atomicExch(ovf, 1u);
    }
    // thj_fusion_block.h's wave-aggregated form: room for n events with one add, then every lane writes its own
    __device__ __forceinline__ unsigned long long reserve(unsigned long long n) { return atomicAdd(count, n); }
    __device__ __forceinline__ void put(unsigned long long pos, uint32_t r1, uint32_t r2, uint32_t l, uint32_t r, uint32_t dir, uint32_t ed) {
        if (pos < cap) { thj_fusion f{r1, r2, l, r, dir, 1u, ed, 0u}; buf[pos] = f; }
        else atomicExch(ovf, 1u);
    }
};

// find_fusions + detect_fusion: thj_fusion_block.h has the workgroup's algorithm (and the history of its phases); here the device's execution
// context for it and the kernel.  Candidate events are appended raw and reduced in thj_fusion_finish.
// (the queue is full -- a tile of multihit reads: the pair where it is found, out of line so that the kernel's registers are not sized for it)
__device__ __noinline__ void detect_fusion_now(const Genome* g, const Params* p, const u64* rp, int W, int rl, bool rc, const Hit* lh, const Hit* rh, int dir, FusionSink* out) {
    detect_fusion(*g, *p, rp, W, rl, rc, *lh, *rh, dir, *out);
}
struct FusBlockDev {
    int tid, lane;
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ unsigned long long ballot(bool q) { return __ballot(q); }
    __device__ __forceinline__ int shfl(int v, int src) { return __shfl(v, src); }
    __device__ __forceinline__ int shfl_up(int v, int d) { return __shfl_up(v, d); }
    __device__ __forceinline__ uint32_t atomic_add(uint32_t* q, uint32_t v) { return atomicAdd(q, v); }
    __device__ __forceinline__ void atomic_or(uint32_t* q, uint32_t v) { atomicOr(q, v); }
    __device__ __forceinline__ void detect_now(const Genome& g, const Params& p, const u64* rp, int W, int rl, bool rc, const Hit& lh, const Hit& rh, int dir, FusionSink& out) {
        detect_fusion_now(&g, &p, rp, W, rl, rc, &lh, &rh, dir, &out);
    }
};
struct FusDevBatch {
    const DevBatch& b; int W;
    __device__ __forceinline__ ReadView view(int r) const { return make_view(b, r); }
    __device__ __forceinline__ int read_len(uint32_t r) const { return (int)b.read_len[r]; }
    __device__ __forceinline__ const u64* planes(uint32_t r) const { return b.planes + (size_t)r * 3 * b.W; }
};
__global__ __launch_bounds__(256, 3) void thj_k_fusion(Genome g, Params p, DevBatch b, FusionSink sink) {
    __shared__ FusBlockShared sh;
    FusBlockDev x{(int)threadIdx.x, (int)(threadIdx.x & 63u)};
    FusDevBatch db{b, b.W};
    fusion_block(x, g, p, db, b.n_reads, (int)blockIdx.x, (int)gridDim.x, sink, sh);
}

__global__ __launch_bounds__(256) void thj_k_ins_gather(const u64* slots, int64_t n, const u64* keys, const u64* vals, u64* out_keys, u64* out_vals) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const u64 h = slots[i];
        out_keys[i] = keys[h];
        out_vals[i] = vals[h];
    }
}

// table growth: every occupied slot of the old table goes into the new one
__global__ __launch_bounds__(256) void thj_k_rehash_keys(const u64* old_tab, u64 old_cap, u64* tab, u64 mask, unsigned long long* count, unsigned int* ovf,
                                                         u64* list) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < old_cap; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = old_tab[i];
        if (k != EMPTY) set_insert<true>(tab, mask, k, count, ovf, list);
    }
}
__global__ __launch_bounds__(256) void thj_k_rehash_ins(const u64* old_keys, const u64* old_vals, u64 old_cap, u64* keys, u64* vals, u64 mask,
                                                        unsigned long long* count, unsigned int* ovf, u64* list) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < old_cap; i += (u64)gridDim.x * blockDim.x) {
        const u64 k = old_keys[i];
        if (k != EMPTY) map_insert_min<true>(keys, vals, mask, k, old_vals[i], count, ovf, list);
    }
}

__global__ __launch_bounds__(256) void thj_k_merge_keys(u64* tab, u64 mask, const u64* keys, int64_t n,
                                                        unsigned long long* count, unsigned int* ovf, u64* list) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        set_insert(tab, mask, keys[i], count, ovf, list);
}

__global__ __launch_bounds__(256) void thj_k_merge_ins(u64* keys, u64* vals, u64 mask, const u64* in_keys, const u64* in_vals,
                                                       int64_t n, unsigned long long* count, unsigned int* ovf, u64* list) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        map_insert_min(keys, vals, mask, in_keys[i], in_vals[i], count, ovf, list);
}

// ------------------------------------------------------------------ context

#include "thj_ctx.h"


static int next_pow2(int64_t x, int64_t* out) {
    int64_t p = 1024;
    while (p < x) p <<= 1;
    *out = p;
    return 0;
}

// dense lists of the distinct events, side by side in d_tmp_keys: junctions | deletions | insertion slots
static inline u64* junc_list(thj_ctx* c) { return c->d_tmp_keys; }
static inline u64* del_list(thj_ctx* c) { return c->d_tmp_keys + c->junc_cap; }
static inline u64* ins_list(thj_ctx* c) { return c->d_tmp_keys + c->junc_cap + c->indel_cap; }

static int free_tables(thj_ctx* c) {
    hipFree(c->d_junc); hipFree(c->d_del); hipFree(c->d_ins_key); hipFree(c->d_ins_val);
    hipFree(c->d_junc_sorted); hipFree(c->d_del_sorted); hipFree(c->d_ins_key_sorted); hipFree(c->d_ins_val_sorted);
    hipFree(c->d_tmp_keys); hipFree(c->d_tmp_vals); hipFree(c->d_tmp_keys2); hipFree(c->d_sort_tmp);
    c->d_junc = c->d_del = c->d_ins_key = c->d_ins_val = nullptr;
    c->d_junc_sorted = c->d_del_sorted = c->d_ins_key_sorted = c->d_ins_val_sorted = nullptr;
    c->d_tmp_keys = c->d_tmp_vals = c->d_tmp_keys2 = nullptr; c->d_sort_tmp = nullptr; c->sort_tmp_bytes = 0;
    return 0;
}

static int alloc_tables(thj_ctx* c, int64_t junc_cap, int64_t indel_cap) {
    free_tables(c);
    next_pow2(junc_cap, &c->junc_cap);
    next_pow2(indel_cap, &c->indel_cap);
    HIPCHK(hipMalloc(&c->d_junc, (size_t)c->junc_cap * 8));
    HIPCHK(hipMalloc(&c->d_del, (size_t)c->indel_cap * 8));
    HIPCHK(hipMalloc(&c->d_ins_key, (size_t)c->indel_cap * 8));
    HIPCHK(hipMalloc(&c->d_ins_val, (size_t)c->indel_cap * 8));
    // a table never holds more than 75 % + one wave of entries
    c->out_cap_junc = c->junc_cap; c->out_cap_indel = c->indel_cap;
    HIPCHK(hipMalloc(&c->d_junc_sorted, (size_t)c->out_cap_junc * 8));
    HIPCHK(hipMalloc(&c->d_tmp_keys, (size_t)(c->out_cap_junc + 2 * c->out_cap_indel) * 8));   // the three event lists side by side
    HIPCHK(hipMalloc(&c->d_del_sorted, (size_t)c->out_cap_indel * 8));
    HIPCHK(hipMalloc(&c->d_ins_key_sorted, (size_t)c->out_cap_indel * 8));
    HIPCHK(hipMalloc(&c->d_ins_val_sorted, (size_t)c->out_cap_indel * 8));
    HIPCHK(hipMalloc(&c->d_tmp_vals, (size_t)c->out_cap_indel * 8));
    HIPCHK(hipMalloc(&c->d_tmp_keys2, (size_t)c->out_cap_indel * 8));
    size_t need = 0, need2 = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, need, (const u64*)nullptr, (u64*)nullptr, (int64_t)c->out_cap_junc, 0, 64, c->stream));
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need2, (const u64*)nullptr, (u64*)nullptr, (const u64*)nullptr, (u64*)nullptr,
                                              (int64_t)c->out_cap_indel, 0, 64, c->stream));
    c->sort_tmp_bytes = need > need2 ? need : need2;
    HIPCHK(hipMalloc(&c->d_sort_tmp, c->sort_tmp_bytes ? c->sort_tmp_bytes : 16));
    return THJ_OK;
}

static int reset_tables_async(thj_ctx* c) {
    HIPCHK(hipMemsetAsync(c->d_junc, 0xFF, (size_t)c->junc_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_del, 0xFF, (size_t)c->indel_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_ins_key, 0xFF, (size_t)c->indel_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_ins_val, 0xFF, (size_t)c->indel_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_ovf, 0, 4 * sizeof(unsigned int), c->stream));
    HIPCHK(hipMemsetAsync(c->d_cnt, 0, CNT_N * sizeof(unsigned long long), c->stream));
    c->n_junc = c->n_del = c->n_ins = 0;
    c->probe_pending = false;                 // counters of the previous pass say nothing about the emptied tables
    c->xchg = nullptr;
    return THJ_OK;
}

extern "C" int thj_device_count(void) {
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    return n;
}

extern "C" int thj_ctx_create(int device, void* stream, thj_ctx** out) {
    if (!out) { thj_set_error("thj_ctx_create: null out"); return THJ_EINVAL; }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) { thj_set_error("device %d not present (%d devices)", device, ndev); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(device));
    thj_ctx* c = new thj_ctx();
    c->device = device;
    if (stream) c->stream = (hipStream_t)stream;
    else { HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    HIPCHK(hipMalloc(&c->d_ovf, 4 * sizeof(unsigned int)));
    HIPCHK(hipMalloc(&c->d_cnt, CNT_N * sizeof(unsigned long long)));
    HIPCHK(hipMalloc(&c->d_out_n, 4 * sizeof(unsigned long long)));
    HIPCHK(hipHostMalloc(&c->h_pinned, 48 * sizeof(unsigned long long)));
    // THJ_TABLE_CAPS=j,i: test knob -- small event tables (slots, powers of two), so that the growth paths run on a small case
    int64_t cap_j = 1ll << 24, cap_i = 1ll << 20;
    if (const char* e = getenv("THJ_TABLE_CAPS")) {
        long long a = 0, b = 0;
        if (sscanf(e, "%lld,%lld", &a, &b) == 2 && a >= 64 && b >= 64 && !(a & (a - 1)) && !(b & (b - 1))) { cap_j = a; cap_i = b; }
    }
    int rc = alloc_tables(c, cap_j, cap_i);
    if (rc) { delete c; return rc; }
    rc = reset_tables_async(c);
    if (rc) { delete c; return rc; }
    *out = c;
    return THJ_OK;
}

// The code objects of the translation units `parts` names (THJ_WARM_*), loaded now by an empty launch from each instead of at the
// first real launch: an executable calls this on the thread that creates the context, while its other threads still read the
// reference and plan the shards (each code object is tens of milliseconds the first shard would otherwise wait for under the
// GPU's lock).  No effect on results.
__global__ void thj_k_warm_segjuncs(int* p) { if (p) *p = 0; }
extern "C" int thj_ctx_warm(thj_ctx* c, int parts) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (parts & 1) hipLaunchKernelGGL(thj_k_warm_segjuncs, dim3(1), dim3(64), 0, c->stream, (int*)nullptr);
    if (parts & 2) thj_warm_span(c->stream);
    if (parts & 4) thj_warm_ingest(c->stream);
    if (parts & 8) thj_warm_bamout(c->stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return THJ_OK;
}

static void cov_free(thj_ctx* c);
extern "C" void thj_ctx_destroy(thj_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    free_tables(c);
    if (c->own_blocks) hipFree((void*)c->d_blocks);
    hipFree(c->d_contig_blk); hipFree(c->d_contig_len);
    hipFree(c->d_ovf); hipFree(c->d_cnt); hipFree(c->d_out_n); for (int i = 0; i < 2; ++i) { hipFree(c->d_rescue_list[i]); hipFree(c->d_rescue_slots[i]); hipFree(c->d_many[i]); hipFree(c->d_sj_lists[i]); } hipFree(c->d_fus_ignore);
    if (c->probe_ev) hipEventDestroy(c->probe_ev);
    hipHostFree(c->h_pinned);
    thj_span_free(c); thj_bamout_free(c);
    cov_free(c);
    if (c->fus_probe_ev) (void)hipEventDestroy(c->fus_probe_ev);
    hipFree(c->d_fus); hipFree(c->d_fus_count); hipFree(c->d_ing0); hipFree(c->d_ing1); hipFree(c->d_infl_tmp); thj_dev_cache_free(c);
    for (hipEvent_t e : c->prof_all) hipEventDestroy(e);
    for (auto e : c->event_pool) hipEventDestroy(e);
    for (int i = 0; i < 3; ++i) if (c->aux_stream[i]) hipStreamDestroy(c->aux_stream[i]);
    for (int i = 0; i < 10; ++i) if (c->aux_ev[i]) hipEventDestroy(c->aux_ev[i]);
    if (c->own_stream) hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int thj_ctx_sync(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipStreamSynchronize(c->stream));
    return THJ_OK;
}

extern "C" void* thj_ctx_stream(thj_ctx* c) { return c ? (void*)c->stream : nullptr; }

static int set_contigs(thj_ctx* c, const uint32_t* contig_blk, const int64_t* lens, int32_t n_contigs, int64_t n_blocks) {
    hipFree(c->d_contig_blk); hipFree(c->d_contig_len);
    c->d_contig_blk = nullptr; c->d_contig_len = nullptr;
    c->h_contig_blk.assign(contig_blk, contig_blk + n_contigs + 1);
    c->h_lens.assign(lens, lens + n_contigs);
    std::vector<int32_t> l32(n_contigs > 0 ? n_contigs : 1);
    for (int32_t i = 0; i < n_contigs; ++i) {
        if (lens[i] < 0 || lens[i] > 0x7fffffff) { thj_set_error("contig %d length unsupported", i); return THJ_EINVAL; }
        l32[i] = (int32_t)lens[i];
    }
    HIPCHK(hipMalloc(&c->d_contig_blk, (size_t)(n_contigs + 1) * 4));
    HIPCHK(hipMalloc(&c->d_contig_len, (size_t)(n_contigs > 0 ? n_contigs : 1) * 4));
    HIPCHK(hipMemcpyAsync(c->d_contig_blk, contig_blk, (size_t)(n_contigs + 1) * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_contig_len, l32.data(), (size_t)n_contigs * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->n_contigs = n_contigs; c->n_blocks = n_blocks;
    return THJ_OK;
}

extern "C" int thj_genome_upload(thj_ctx* c, const uint64_t* blocks, int64_t n_blocks, const uint32_t* contig_blk,
                                 const int64_t* lens, int32_t n_contigs) {
    if (!c || !blocks || !contig_blk || !lens || n_blocks <= 0) { thj_set_error("thj_genome_upload: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (c->own_blocks) hipFree((void*)c->d_blocks);
    c->d_blocks = nullptr; c->own_blocks = false;
    u64* d = nullptr;
    HIPCHK(hipMalloc(&d, (size_t)n_blocks * 32));
    HIPCHK(hipMemcpyAsync(d, blocks, (size_t)n_blocks * 32, hipMemcpyHostToDevice, c->stream));
    c->d_blocks = d; c->own_blocks = true;
    return set_contigs(c, contig_blk, lens, n_contigs, n_blocks);
}

extern "C" int thj_genome_adopt(thj_ctx* c, const void* d_blocks, int64_t n_blocks, const uint32_t* contig_blk,
                                const int64_t* lens, int32_t n_contigs) {
    if (!c || !d_blocks || !contig_blk || !lens) { thj_set_error("thj_genome_adopt: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (c->own_blocks) hipFree((void*)c->d_blocks);
    c->d_blocks = (const u64*)d_blocks; c->own_blocks = false;
    return set_contigs(c, contig_blk, lens, n_contigs, n_blocks);
}

// (thj_ensure_aux_streams: thj_streams.hip -- a translation unit of its own, so that its spin kernel does not pull this file's code object in)

// ------------------------------------------------------------------ batches

struct OwnedBatch {
    thj_seg_batch desc;        // device pointers; MUST be the first member
    void* ptrs[6];
};

extern "C" int thj_batch_upload(thj_ctx* c, const thj_seg_batch* h, int64_t n_hits, int64_t n_mate_hits, thj_seg_batch** out) {
    if (!c || !h || !out) { thj_set_error("thj_batch_upload: null argument"); return THJ_EINVAL; }
    if (h->n_reads < 0 || h->nseg < 1 || h->nseg > 16 || h->words_per_plane < 1) { thj_set_error("thj_batch_upload: bad shape (nseg must be 1..16)"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    OwnedBatch* ob = new OwnedBatch();
    memset(ob, 0, sizeof *ob);
    ob->desc = *h;
    const int64_t n = h->n_reads;
    const size_t sizes[6] = {(size_t)(n * h->nseg + 1) * 4, (size_t)n_hits * 16, (size_t)n * 3 * h->words_per_plane * 8,
                             (size_t)n * 2, h->mate_off ? (size_t)(n + 1) * 4 : 0, h->mate_off ? (size_t)n_mate_hits * 16 : 0};
    const void* src[6] = {h->seg_off, h->hits, h->read_planes, h->read_len, h->mate_off, h->mate_hits};
    for (int i = 0; i < 6; ++i) {
        if (i >= 4 && !h->mate_off) { ob->ptrs[i] = nullptr; continue; }
        { int rc_ = thj_dev_alloc(c, &ob->ptrs[i], sizes[i] ? sizes[i] : 16); if (rc_) return rc_; }
        if (sizes[i]) HIPCHK(hipMemcpyAsync(ob->ptrs[i], src[i], sizes[i], hipMemcpyHostToDevice, c->stream));
    }
    ob->desc.seg_off = (const uint32_t*)ob->ptrs[0];
    ob->desc.hits = (const thj_hit*)ob->ptrs[1];
    ob->desc.read_planes = (const uint64_t*)ob->ptrs[2];
    ob->desc.read_len = (const uint16_t*)ob->ptrs[3];
    ob->desc.mate_off = (const uint32_t*)ob->ptrs[4];
    ob->desc.mate_hits = (const thj_hit*)ob->ptrs[5];
    HIPCHK(hipStreamSynchronize(c->stream));     // host buffers may be released by the caller
    *out = &ob->desc;
    return THJ_OK;
}

extern "C" int thj_batch_free(thj_ctx* c, thj_seg_batch* dev) {
    if (!c || !dev) return THJ_OK;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    OwnedBatch* ob = (OwnedBatch*)dev;
    for (int i = 0; i < 6; ++i) thj_dev_release(c, ob->ptrs[i]);
    delete ob;
    return THJ_OK;
}

// ------------------------------------------------------------------ run

// The event tables grow by themselves: after every run the insert counters are copied to pinned memory (no
// synchronisation); the next run looks at them and, past 40 % load, rehashes into tables four times the size.
// thj_segjuncs_finish still refuses a table that ended up more than 75 % full (one batch adding > 35 % of a table).
static int grow_tables(thj_ctx* c, bool grow_junc, bool grow_indel) {
    HIPCHK(hipStreamSynchronize(c->stream));
    u64* oj = c->d_junc; u64* od = c->d_del; u64* oik = c->d_ins_key; u64* oiv = c->d_ins_val;
    const int64_t ojc = c->junc_cap, oic = c->indel_cap;
    c->d_junc = c->d_del = c->d_ins_key = c->d_ins_val = nullptr;          // alloc_tables frees what the context still holds
    int rc = alloc_tables(c, grow_junc ? ojc * 4 : ojc, grow_indel ? oic * 4 : oic);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(c->d_junc, 0xFF, (size_t)c->junc_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_del, 0xFF, (size_t)c->indel_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_ins_key, 0xFF, (size_t)c->indel_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_ins_val, 0xFF, (size_t)c->indel_cap * 8, c->stream));
    HIPCHK(hipMemsetAsync(&c->d_cnt[CNT_JUNC], 0, 3 * sizeof(unsigned long long), c->stream));      // JUNC, DEL, INS: recounted by the rehash
    auto blocks_for = [](int64_t cap) { int64_t b = (cap + 255) / 256; return (unsigned)(b > 4096 ? 4096 : b); };
    hipLaunchKernelGGL(thj_k_rehash_keys, dim3(blocks_for(ojc)), dim3(256), 0, c->stream, (const u64*)oj, (u64)ojc, c->d_junc,
                       (u64)c->junc_cap - 1, &c->d_cnt[CNT_JUNC], &c->d_ovf[0], junc_list(c));
    hipLaunchKernelGGL(thj_k_rehash_keys, dim3(blocks_for(oic)), dim3(256), 0, c->stream, (const u64*)od, (u64)oic, c->d_del,
                       (u64)c->indel_cap - 1, &c->d_cnt[CNT_DEL], &c->d_ovf[1], del_list(c));
    hipLaunchKernelGGL(thj_k_rehash_ins, dim3(blocks_for(oic)), dim3(256), 0, c->stream, (const u64*)oik, (const u64*)oiv, (u64)oic,
                       c->d_ins_key, c->d_ins_val, (u64)c->indel_cap - 1, &c->d_cnt[CNT_INS], &c->d_ovf[2], ins_list(c));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(oj); hipFree(od); hipFree(oik); hipFree(oiv);
    return THJ_OK;
}

static int maybe_grow_tables(thj_ctx* c) {
    if (!c->probe_pending || hipEventQuery(c->probe_ev) != hipSuccess) return THJ_OK;
    c->probe_pending = false;
    const unsigned long long* n = &c->h_pinned[32];
    const bool gj = n[CNT_JUNC] * 5 > (unsigned long long)c->junc_cap * 2;
    const bool gi = (n[CNT_DEL] > n[CNT_INS] ? n[CNT_DEL] : n[CNT_INS]) * 5 > (unsigned long long)c->indel_cap * 2;
    return (gj || gi) ? grow_tables(c, gj, gi) : THJ_OK;
}

extern "C" int thj_segjuncs_configure(thj_ctx* c, int64_t junc_capacity, int64_t indel_capacity) {
    if (!c || junc_capacity < 1 || indel_capacity < 1) { thj_set_error("thj_segjuncs_configure: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    int rc = alloc_tables(c, junc_capacity, indel_capacity);
    if (rc) return rc;
    return reset_tables_async(c);
}

extern "C" int thj_segjuncs_reset_async(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    return reset_tables_async(c);
}

static int check_params(const thj_params* p, const thj_seg_batch* b) {
    if (p->segment_length < 8 || p->segment_length > 64) {
        thj_set_error("segment_length %d unsupported by the device path (8..64: the 8 bases either side of a segment boundary "
                      "that make a support read must lie inside the segments, segment_juncs.cpp:3580-3584; a 2L read piece and an "
                      "L+16 support read must fit one 128-bit plane word)", p->segment_length);
        return THJ_EINVAL;
    }
    if (p->max_insertion_length > 6 || p->max_insertion_length < 0) { thj_set_error("max_insertion_length %d unsupported (0..6)", p->max_insertion_length); return THJ_EINVAL; }
    if (p->max_deletion_length < 0 || p->max_deletion_length > 1000000) { thj_set_error("max_deletion_length out of range"); return THJ_EINVAL; }
    if (p->max_segment_intron + p->segment_length + 64 >= (1 << 29)) { thj_set_error("max_segment_intron too large for the packed key"); return THJ_EINVAL; }
    if (b->nseg < 1 || b->nseg > 16) { thj_set_error("nseg %d unsupported (1..16)", b->nseg); return THJ_EINVAL; }
    if (b->words_per_plane < 1 || b->words_per_plane > 8) { thj_set_error("words_per_plane %d unsupported (1..8: reads of up to 512 bases)", b->words_per_plane); return THJ_EINVAL; }
    if (b->n_reads < 0 || (int64_t)b->n_reads + b->ordinal_base >= (1ll << 29)) { thj_set_error("batch too large: read ordinals must stay below 2^29"); return THJ_EINVAL; }
    return THJ_OK;
}

int thj_dev_alloc(thj_ctx* c, void** out, size_t bytes) {
    if (bytes < 256) bytes = 256;
    int best = -1;
    for (size_t i = 0; i < c->dev_cache.size(); ++i) {
        const thj_ctx::DevBlock& b = c->dev_cache[i];
        if (!b.used && b.cap >= bytes && b.cap <= 2 * bytes + 65536 && (best < 0 || b.cap < c->dev_cache[(size_t)best].cap)) best = (int)i;
    }
    if (best >= 0) { c->dev_cache[(size_t)best].used = true; *out = c->dev_cache[(size_t)best].p; return THJ_OK; }
    // keep the cache bounded: drop idle blocks once it holds more than 16 GiB
    if (c->dev_cache_bytes > ((size_t)16 << 30)) {
        for (size_t i = 0; i < c->dev_cache.size();) {
            if (!c->dev_cache[i].used) { hipFree(c->dev_cache[i].p); c->dev_cache_bytes -= c->dev_cache[i].cap; c->dev_cache.erase(c->dev_cache.begin() + (ptrdiff_t)i); }
            else ++i;
        }
    }
    const size_t cap = bytes + bytes / 8;
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, cap));
    c->dev_cache.push_back({p, cap, true});
    c->dev_cache_bytes += cap;
    *out = p;
    return THJ_OK;
}
void thj_dev_release(thj_ctx* c, void* p) {
    if (!p) return;
    for (auto& b : c->dev_cache) if (b.p == p) { b.used = false; return; }
    hipFree(p);                                 // not one of ours
}
void thj_dev_cache_free(thj_ctx* c) {
    for (auto& b : c->dev_cache) hipFree(b.p);
    c->dev_cache.clear(); c->dev_cache_bytes = 0;
}

hipEvent_t thj_get_event(thj_ctx* c) {
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}

// One batch's kernels.  `set` (0 / 1) names the scratch lists and side streams the launch uses: two batches launched one after
// the other on different sets run their side chains beside each other (thj_segjuncs_run_pair_async); *joined is set when the
// launch left work on side streams that sj_join has to bring back to the context's stream.
struct SjState {
    Genome g; Params p; DevBatch b; Tables t; RescueList rl; SjLists sl; XTasks x;
    int grid, n_tiles; bool wide, serial;
    hipStream_t sm, sa, sb, sc; hipEvent_t* aev; hipEvent_t m0, m1;
};
// first half: the scratch lists, thj_k_sj_flat on the context's stream, the side streams told to wait for it
static bool beside_env() { static const bool v = getenv("THJ_SJ_SHARED_BESIDE") != nullptr; return v; }
static int sj_launch_flat(thj_ctx* c, const thj_params* tp, const thj_seg_batch* db, int set, SjState& st) {
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    Params p;
    memcpy(&p, tp, sizeof p);
    DevBatch b;
    memcpy(&b, db, sizeof b);
    Tables t{c->d_junc, (u64)c->junc_cap - 1, c->d_del, (u64)c->indel_cap - 1, c->d_ins_key, c->d_ins_val,
             (u64)c->indel_cap - 1, junc_list(c), del_list(c), ins_list(c), c->d_ovf, c->d_cnt};
    const int n = b.n_reads;

#ifdef THJ_EXP
    { int f = getenv("THJ_EXP_FLAGS") ? atoi(getenv("THJ_EXP_FLAGS")) : 0; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(thj_exp_flags), &f, sizeof f)); }
#endif
    const int n_tiles = (n + TPB - 1) / TPB;
    int grid = n_tiles < 256 * 8 - 1 ? n_tiles : 256 * 8 - 1;       // 256 CUs x 8 resident workgroups, grid-stride the rest (one rescue-list slice is thj_k_segjuncs_shared's)
    // rescue list: one slice per workgroup, sized for all the reads the workgroup visits
    RescueList rl;
    rl.seg_cap = (n_tiles + grid - 1) / grid * TPB;
    const int64_t need = (int64_t)grid * rl.seg_cap + 2 * MANY_CAP + MAX_LISTS;   // the workgroups' slices, the one thj_k_segjuncs_shared and thj_k_sj_general's second instance share (room for every read they may get), the counts
    if (c->rescue_list_cap[set] < need) {
        hipFree(c->d_rescue_list[set]); c->d_rescue_list[set] = nullptr;
        HIPCHK(hipMalloc(&c->d_rescue_list[set], (size_t)need * 4));
        c->rescue_list_cap[set] = need;
    }
    rl.list = c->d_rescue_list[set];
    rl.blk_cnt = c->d_rescue_list[set] + (int64_t)grid * rl.seg_cap + 2 * MANY_CAP;
    rl.own_slice = grid;
    static const int many_min = getenv("THJ_MANY_HITS") ? atoi(getenv("THJ_MANY_HITS")) : MID_HITS;
    rl.many_min = many_min;
    rl.mid_min = many_min < GEN_HITS ? many_min : GEN_HITS;
    // [8 words: the count of the reads with many hits, of the batches thj_k_segjuncs_shared has drawn, of thj_k_sj_general's second list, of the tasks in the
    // list, of the flat rescue pairs][the two lists]
    if (!c->d_many[set]) HIPCHK(hipMalloc((void**)&c->d_many[set], 32 + (size_t)MANY_CAP * 8));
    rl.many_count = (unsigned int*)c->d_many[set];
    rl.many_list = c->d_many[set] + 8;
    HIPCHK(hipMemsetAsync(c->d_many[set], 0, 32, c->stream));
    HIPCHK(hipMemsetAsync(rl.blk_cnt + grid, 0, 4, c->stream));
    // [16 bytes: the count of listed reads][HEAVY_CAP read indices]
    if (b.mate_off && !c->d_rescue_slots[set]) HIPCHK(hipMalloc((void**)&c->d_rescue_slots[set], 16 + (size_t)HEAVY_CAP * 4));
    rl.heavy_count = (unsigned int*)c->d_rescue_slots[set];
    rl.heavy_list = c->d_rescue_slots[set] ? (uint32_t*)c->d_rescue_slots[set] + 4 : nullptr;
    rl.slot_pool = nullptr;
    if (b.mate_off) HIPCHK(hipMemsetAsync(c->d_rescue_slots[set], 0, 16, c->stream));
    // the flat kernels' lists: a slice per workgroup, each sized for the most its reads can give (a flat read: nseg - 2 indel
    // pairs and nseg - 1 windows, or 2 per mate hit when it takes the rescue) -- sparse in a large allocation, never overflowing
    SjLists sl;
    XTasks x;
    size_t xcap;
    {
        const int64_t S = (int64_t)grid * rl.seg_cap;
        const int tmax = (b.nseg > 2 ? b.nseg - 2 : 0) + (b.nseg - 1 > 2 * FLAT_MATES ? b.nseg - 1 : 2 * FLAT_MATES);
        sl.task_cap = rl.seg_cap * tmax;
        xcap = (size_t)2 * (size_t)n > ((size_t)1 << 20) ? (size_t)2 * (size_t)n : ((size_t)1 << 20);     // the list of the kernels that enumerate from lists: two tasks per read of the batch
        if (getenv("THJ_XTASK_CAP")) xcap = (size_t)atoll(getenv("THJ_XTASK_CAP"));                       // (tests: a full list must fail loudly)
        const size_t bytes = (size_t)S * tmax * 20 + (size_t)S * 4 * (1 + FLAT_MATES + 1) + (size_t)S * FLAT_MATES * 8 + (size_t)grid * 16 + xcap * 20 + 512;
        if (c->sj_lists_cap[set] < bytes) {
            hipFree(c->d_sj_lists[set]); c->d_sj_lists[set] = nullptr; c->sj_lists_cap[set] = 0;
            HIPCHK(hipMalloc(&c->d_sj_lists[set], bytes + bytes / 8));
            c->sj_lists_cap[set] = bytes + bytes / 8;
        }
        char* q = (char*)c->d_sj_lists[set];
        sl.tq = (uint4*)q; q += (size_t)S * tmax * 16;
        x.q = (uint4*)q; q += xcap * 16;
        sl.scan = (int2*)q; q += (size_t)S * FLAT_MATES * 8;
        sl.te = (uint32_t*)q; q += (size_t)S * tmax * 4;
        sl.frl = (uint32_t*)q; q += (size_t)S * 4;
        sl.pairs = (uint32_t*)q; q += (size_t)S * FLAT_MATES * 4;
        sl.gen = (uint32_t*)q; q += (size_t)S * 4;
        x.e = (uint32_t*)q; q += xcap * 4;
        x.count = (unsigned int*)c->d_many[set] + 3; x.cap = (unsigned int)(xcap < 0xFFFFFFFFull ? xcap : 0xFFFFFFFFull); x.ovf = c->d_ovf + 3;
        sl.task_cnt = (unsigned int*)q; sl.frl_cnt = sl.task_cnt + grid; sl.pair_cnt = sl.frl_cnt + grid; sl.gen_cnt = sl.pair_cnt + grid;
        sl.mid_count = (unsigned int*)c->d_many[set] + 2; sl.mid_list = c->d_many[set] + 8 + MANY_CAP;      // [4]: the launch's flat rescue pairs (statistics)
    }
    const bool wide = p.segment_length > 32;
    // Two chains after thj_k_sj_flat, side by side on two streams: the flat reads' (rescue scan, rescue enumeration, tasks:
    // dense, bound by the genome lines they fetch) on the context's stream, and the reads with several hits a segment (general x 2,
    // rescue x 2, tasks: few waves per CU, each waiting on its own chain of loads) on a side stream, with thj_k_segjuncs_shared
    // beside them on another.  Everything is joined on the context's stream again before the caller returns; the event tables
    // take inserts from any of them.  THJ_SJ_SERIAL=1: one stream.
    static const bool serial_env = getenv("THJ_SJ_SERIAL") && atoi(getenv("THJ_SJ_SERIAL")) != 0;
    const bool serial = serial_env || c->serial_launch;
    if (!serial) { const int src = thj_ensure_aux_streams(c, beside_env() ? 3 : set + 1); if (src) return src; }
    // the set's chain on the set's side stream: thj_k_segjuncs_shared, both general instances, then the rescue kernels and the tasks.
    // (THJ_SJ_SHARED_BESIDE: developer switch -- thj_k_segjuncs_shared beside the chain on a third side stream, as before round 5; the third
    // stream shares a hardware queue with one of the others, and what was enqueued behind it waited: 5.7 against 5.55 ms per step)
    static const bool beside = beside_env();
    hipStream_t sm = c->stream, sa = serial ? c->stream : c->aux_stream[set], sb = serial ? c->stream : beside ? c->aux_stream[2] : sa, sc = sa;
    hipEvent_t* const aev = c->aux_ev + 5 * set;
    // profiling: one pair of events around every kernel (pairs of kernels where the second is the first's tail), on the stream it runs on;
    // SJ_PROF_N intervals per launch, in the order thj_profile_segjuncs documents
    auto mark = [&](hipStream_t st) -> hipEvent_t { if (!c->profile) return nullptr; hipEvent_t e = thj_get_event(c); hipEventRecord(e, st); c->prof_all.push_back(e); return e; };
    hipEvent_t m0 = mark(sm);
    if (b.nseg <= 4) hipLaunchKernelGGL(thj_k_sj_flat<4>, dim3(grid), dim3(TPB), 0, sm, p, b, rl, sl, c->d_cnt);
    else if (b.nseg <= 8) hipLaunchKernelGGL(thj_k_sj_flat<8>, dim3(grid), dim3(TPB), 0, sm, p, b, rl, sl, c->d_cnt);
    else hipLaunchKernelGGL(thj_k_sj_flat<16>, dim3(grid), dim3(TPB), 0, sm, p, b, rl, sl, c->d_cnt);
    hipEvent_t m1 = mark(sm);
    if (!serial) { HIPCHK(hipEventRecord(aev[0], sm)); HIPCHK(hipStreamWaitEvent(sa, aev[0], 0)); }      // (sb waits where its kernel is launched: it is shared by the sets)
    st.g = g; st.p = p; st.b = b; st.t = t; st.rl = rl; st.sl = sl; st.x = x; st.grid = grid; st.n_tiles = n_tiles; st.wide = wide; st.serial = serial;
    st.sm = sm; st.sa = sa; st.sb = sb; st.sc = sc; st.aev = aev; st.m0 = m0; st.m1 = m1;
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

// second half: the side chains on their streams, the flat reads' rescue and tasks on the context's
static int sj_launch_rest(thj_ctx* c, SjState& st, int set, bool* joined) {
    Genome& g = st.g; Params& p = st.p; DevBatch& b = st.b; Tables& t = st.t; RescueList& rl = st.rl; SjLists& sl = st.sl; XTasks& x = st.x;
    const int grid = st.grid, n_tiles = st.n_tiles; const bool wide = st.wide, serial = st.serial;
    hipStream_t sm = st.sm, sa = st.sa, sb = st.sb, sc = st.sc; hipEvent_t* const aev = st.aev; hipEvent_t m0 = st.m0, m1 = st.m1;
    auto mark = [&](hipStream_t s_) -> hipEvent_t { if (!c->profile) return nullptr; hipEvent_t e = thj_get_event(c); hipEventRecord(e, s_); c->prof_all.push_back(e); return e; };
    auto span = [&](hipEvent_t a, hipEvent_t z) { if (c->profile) c->prof_events.emplace_back(a, z); };
    // ---- the reads with several hits a segment
    const int rgrid = grid < RESCUE_GRID ? grid : RESCUE_GRID;
    hipEvent_t b0 = mark(sb);
    if (!serial) HIPCHK(hipStreamWaitEvent(sb, aev[0], 0));
    hipLaunchKernelGGL(thj_k_segjuncs_shared, dim3(rgrid), dim3(TPB), 0, sb, p, b, rl, x, c->d_cnt);     // the reads with many hits: a wave each (the longest of the three: first)
    hipEvent_t b1 = mark(sb);
    if (!serial) HIPCHK(hipEventRecord(aev[2], sb));
    hipEvent_t c0 = mark(sc);
    {
        const int mgrid = n_tiles < MID_GRID ? n_tiles : MID_GRID;
        if (b.nseg <= 8) hipLaunchKernelGGL((thj_k_sj_general<MID_HITS, MID_T, false, 9, MID_G>), dim3(mgrid), dim3(MID_T), 0, sc, p, b, rl, sl, x, c->d_cnt);
        else hipLaunchKernelGGL((thj_k_sj_general<MID_HITS, MID_T, false, 17, MID_G>), dim3(mgrid), dim3(MID_T), 0, sc, p, b, rl, sl, x, c->d_cnt);
    }
    hipEvent_t c1 = mark(sc);
    hipEvent_t a0 = mark(sa);
    if (b.nseg <= 8) hipLaunchKernelGGL((thj_k_sj_general<GEN_HITS, TPB, true, 9, GEN_G>), dim3(grid), dim3(TPB), 0, sa, p, b, rl, sl, x, c->d_cnt);
    else hipLaunchKernelGGL((thj_k_sj_general<GEN_HITS, TPB, true, 17, GEN_G>), dim3(grid), dim3(TPB), 0, sa, p, b, rl, sl, x, c->d_cnt);
    hipEvent_t a1 = mark(sa);
    if (!serial) HIPCHK(hipStreamWaitEvent(sa, aev[2], 0));                // (thj_k_segjuncs_shared's reads for the rescue are listed)
    hipEvent_t a3 = mark(sa);                    // (after the wait for thj_k_segjuncs_shared)
    if (b.mate_off) {
        hipLaunchKernelGGL(thj_k_segjuncs_rescue, dim3(rgrid), dim3(TPB), 0, sa, g, p, b, rl, grid + 1, x, c->d_cnt);
        hipLaunchKernelGGL(thj_k_segjuncs_rescue_shared, dim3(rgrid), dim3(TPB), 0, sa, g, p, b, rl, x, c->d_cnt);
    }
    hipEvent_t a4 = mark(sa);
    // ... and their tasks
    if (wide) hipLaunchKernelGGL(thj_k_sj_tasks_list<true>, dim3(rgrid), dim3(TPB), 0, sa, g, p, b, t, x);
    else hipLaunchKernelGGL(thj_k_sj_tasks_list<false>, dim3(rgrid), dim3(TPB), 0, sa, g, p, b, t, x);
    hipEvent_t a5 = mark(sa);
    if (!serial) HIPCHK(hipEventRecord(aev[3], sa));
    // ---- the flat reads
    hipEvent_t m2 = mark(sm);
    if (b.mate_off) {
        hipLaunchKernelGGL(thj_k_sj_rescue_scan, dim3(grid), dim3(TPB), 0, sm, g, p, b, sl, rl.seg_cap);
        hipLaunchKernelGGL(thj_k_sj_rescue_flat, dim3(grid), dim3(TPB), 0, sm, p, b, sl, rl.seg_cap, c->d_cnt, sl.mid_count + 2);
    }
    hipEvent_t m3 = mark(sm);
    if (wide) hipLaunchKernelGGL(thj_k_sj_tasks<true>, dim3(grid), dim3(TPB), 0, sm, g, p, b, t, sl);
    else hipLaunchKernelGGL(thj_k_sj_tasks<false>, dim3(grid), dim3(TPB), 0, sm, g, p, b, t, sl);
    hipEvent_t m4 = mark(sm);
    span(m0, m1); span(a0, a1); span(c0, c1); span(b0, b1); span(a3, a4); span(a4, a5); span(m2, m3); span(m3, m4);
    if (c->profile) c->prof_sets.push_back(set);
    *joined = !serial;
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

static int sj_launch(thj_ctx* c, const thj_params* tp, const thj_seg_batch* db, int set, bool* joined) {
    SjState st;
    int rc = sj_launch_flat(c, tp, db, set, st);
    return rc ? rc : sj_launch_rest(c, st, set, joined);
}

static int sj_join(thj_ctx* c, int set) { HIPCHK(hipStreamWaitEvent(c->stream, c->aux_ev[5 * set + 3], 0)); return THJ_OK; }
static int sj_probe(thj_ctx* c) {
    // insert counters for the next run's growth decision (asynchronous)
    if (!c->probe_ev) HIPCHK(hipEventCreateWithFlags(&c->probe_ev, hipEventDisableTiming));
    if (!c->probe_pending) {
        HIPCHK(hipMemcpyAsync(&c->h_pinned[32], c->d_cnt, CNT_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->probe_ev, c->stream));
        c->probe_pending = true;
    }
    return THJ_OK;
}

extern "C" int thj_segjuncs_run_async(thj_ctx* c, const thj_params* tp, const thj_seg_batch* db) {
    if (!c || !tp || !db) { thj_set_error("thj_segjuncs_run_async: null argument"); return THJ_EINVAL; }
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    int rc = check_params(tp, db);
    if (rc) return rc;
    HIPCHK(hipSetDevice(c->device));
    if (db->n_reads == 0) return THJ_OK;
    if ((rc = maybe_grow_tables(c))) return rc;
    bool joined = false;
    if ((rc = sj_launch(c, tp, db, 0, &joined))) return rc;
    if (joined && (rc = sj_join(c, 0))) return rc;
    return sj_probe(c);
}

// Two batches (the two sides of a pass) as one call: the same as two thj_segjuncs_run_async calls in this order, but the second
// batch's kernels do not wait for the first batch's side chains (the reads with several hits a segment: a long tail of small,
// latency-bound kernels that leave most of the GPU idle) -- each batch has its own scratch lists and side streams, the event
// tables take inserts from all of them, and everything is back on the context's stream when the call returns.
extern "C" int thj_segjuncs_run_pair_async(thj_ctx* c, const thj_params* tp0, const thj_seg_batch* db0, const thj_params* tp1, const thj_seg_batch* db1) {
    if (!c || !tp0 || !db0 || !tp1 || !db1) { thj_set_error("thj_segjuncs_run_pair_async: null argument"); return THJ_EINVAL; }
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    int rc = check_params(tp0, db0);
    if (rc || (rc = check_params(tp1, db1))) return rc;
    HIPCHK(hipSetDevice(c->device));
    if ((rc = maybe_grow_tables(c))) return rc;
    bool j0 = false, j1 = false;
    // both thj_k_sj_flat first, then the rest of each: the second batch's side chains start as early as they can
    SjState s0, s1;
    if (db0->n_reads && (rc = sj_launch_flat(c, tp0, db0, 0, s0))) return rc;
    if (db1->n_reads && (rc = sj_launch_flat(c, tp1, db1, 1, s1))) return rc;
    if (db0->n_reads && (rc = sj_launch_rest(c, s0, 0, &j0))) return rc;
    if (db1->n_reads && (rc = sj_launch_rest(c, s1, 1, &j1))) { if (j0) sj_join(c, 0); return rc; }
    if (j0 && (rc = sj_join(c, 0))) return rc;
    if (j1 && (rc = sj_join(c, 1))) return rc;
    return (db0->n_reads || db1->n_reads) ? sj_probe(c) : THJ_OK;
}

extern "C" int thj_profile_serial(thj_ctx* c, int on) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->serial_launch = on != 0;
    return THJ_OK;
}

static constexpr int SJ_PROF_N = 8;
extern "C" int thj_profile_segjuncs(thj_ctx* c, int enable, double* avg_ms, int64_t* launches, double* stats) {
    // avg_ms[8], per thj_segjuncs_run_async (a pair call counts as two): thj_k_sj_flat; thj_k_sj_general, first instance; second instance;
    // thj_k_segjuncs_shared; thj_k_segjuncs_rescue + _rescue_shared; thj_k_sj_tasks_list; thj_k_sj_rescue_scan + thj_k_sj_rescue_flat; thj_k_sj_tasks.
    // stats[4], averages per launch over the launches whose lists are still there (the last on each scratch set): reads in
    // thj_k_segjuncs_shared's list, in thj_k_sj_general's second list, tasks in the list thj_k_sj_tasks_list ran, flat rescue pairs.
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    double sum[SJ_PROF_N] = {0};
    const size_t n = c->prof_events.size() / SJ_PROF_N;
    for (size_t i = 0; i < c->prof_events.size(); ++i) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_events[i].first, c->prof_events[i].second));
        sum[i % SJ_PROF_N] += ms;
    }
    for (hipEvent_t e : c->prof_all) c->event_pool.push_back(e);
    if (launches) *launches = (int64_t)n;
    if (avg_ms) for (int k = 0; k < SJ_PROF_N; ++k) avg_ms[k] = n ? sum[k] / (double)n : 0.0;
    if (stats) {
        bool used[2] = {false, false};
        for (int st : c->prof_sets) used[st & 1] = true;
        double acc[4] = {0, 0, 0, 0}; int ns = 0;
        for (int st = 0; st < 2; ++st) if (used[st] && c->d_many[st]) {
            unsigned int h[8];
            HIPCHK(hipMemcpy(h, c->d_many[st], sizeof h, hipMemcpyDeviceToHost));
            acc[0] += h[0]; acc[1] += h[2]; acc[2] += h[3]; acc[3] += h[4]; ++ns;
        }
        for (int k = 0; k < 4; ++k) stats[k] = ns ? acc[k] / ns : 0.0;
    }
    c->prof_events.clear(); c->prof_all.clear(); c->prof_sets.clear();
    c->profile = enable != 0;
    return THJ_OK;
}

extern "C" int thj_segjuncs_merge_keys_async(thj_ctx* c, int kind, const uint64_t* d_keys, int64_t n) {
    if (!c || (kind != 0 && kind != 1) || (n > 0 && !d_keys)) { thj_set_error("thj_segjuncs_merge_keys_async: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return THJ_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (kind == 0)
        hipLaunchKernelGGL(thj_k_merge_keys, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_junc, (u64)c->junc_cap - 1,
                           (const u64*)d_keys, n, &c->d_cnt[CNT_JUNC], &c->d_ovf[0], junc_list(c));
    else
        hipLaunchKernelGGL(thj_k_merge_keys, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_del, (u64)c->indel_cap - 1,
                           (const u64*)d_keys, n, &c->d_cnt[CNT_DEL], &c->d_ovf[1], del_list(c));
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

static void fusion_drop_device_set(thj_ctx* c);
extern "C" int thj_fusion_reset_async(thj_ctx* c) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_fus_count) {
        HIPCHK(hipMalloc(&c->d_fus_count, 16));
        c->fus_cap = 1 << 22;                   // (grows: thj_fusion_run_async)
        if (const char* e = getenv("THJ_FUSION_CAP")) { const long long v = atoll(e); if (v >= 64) c->fus_cap = v; }      // test knob: a small buffer, so that it has to grow
        HIPCHK(hipMalloc(&c->d_fus, (size_t)c->fus_cap * sizeof(thj_fusion)));
    }
    HIPCHK(hipMemsetAsync(c->d_fus_count, 0, 16, c->stream));
    if (c->fus_probe_pending) { (void)hipEventSynchronize(c->fus_probe_ev); c->fus_probe_pending = false; }
    c->h_fusions.clear();
    fusion_drop_device_set(c);
    return THJ_OK;
}

extern "C" int thj_fusion_set_ignored(thj_ctx* c, const uint32_t* ref_ids, int32_t n) {
    // --fusion-ignore-chromosomes: pairs with a hit on one of these contigs are skipped (segment_juncs.cpp:3214-3231)
    if (!c || (n > 0 && !ref_ids) || n < 0) { thj_set_error("thj_fusion_set_ignored: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    hipFree(c->d_fus_ignore); c->d_fus_ignore = nullptr; c->n_fus_ignore = 0;
    if (n == 0) return THJ_OK;
    uint32_t mx = 0;
    for (int32_t i = 0; i < n; ++i) if (ref_ids[i] > mx) mx = ref_ids[i];
    std::vector<uint8_t> flags((size_t)mx + 1, 0);
    for (int32_t i = 0; i < n; ++i) flags[ref_ids[i]] = 1;
    HIPCHK(hipMalloc(&c->d_fus_ignore, flags.size()));
    HIPCHK(hipMemcpy(c->d_fus_ignore, flags.data(), flags.size(), hipMemcpyHostToDevice));
    c->n_fus_ignore = (int64_t)flags.size();
    return THJ_OK;
}

extern "C" int thj_fusion_run_async(thj_ctx* c, const thj_params* tp, const thj_seg_batch* db) {
    if (!c || !tp || !db) { thj_set_error("thj_fusion_run_async: null argument"); return THJ_EINVAL; }
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    int rc = check_params(tp, db);
    if (rc) return rc;
    if (db->words_per_plane > 4 || db->nseg > 8) { thj_set_error("reads longer than 256 bases or of more than eight segments are not supported by the fusion kernel"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (!c->d_fus_count) { rc = thj_fusion_reset_async(c); if (rc) return rc; }
    if (db->n_reads == 0) return THJ_OK;
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    Params p; memcpy(&p, tp, sizeof p);
    DevBatch b; memcpy(&b, db, sizeof b);
    // The raw candidate events of a pass pile up until thj_fusion_finish reduces them (round 6: configs[3] at full size -- 50 M pairs of the
    // mix -- has 9.9 M of them, the buffer held 1 M: a loud error, but an error).  The buffer now grows ahead of the count: after every launch
    // the count sets out for the host, the next call looks at what has arrived and, past half the room, moves the events to a buffer four
    // times the size (one stream synchronisation per growth; no wait otherwise).  A single batch that adds more than the room left still
    // overflows: thj_fusion_finish then enlarges the buffer to the count and answers THJ_ERETRY.
    if (!c->fus_probe_ev) HIPCHK(hipEventCreateWithFlags(&c->fus_probe_ev, hipEventDisableTiming));
    if (c->fus_probe_pending && hipEventQuery(c->fus_probe_ev) == hipSuccess) {
        c->fus_probe_pending = false;
        if ((int64_t)c->h_pinned[44] * 2 > c->fus_cap && c->fus_cap < (1ll << 30)) {
            unsigned long long h[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(h, c->d_fus_count, 16, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if (!(unsigned int)h[1]) {                                  // (an overflow that has happened stays one)
                int64_t ncap = c->fus_cap * 4;
                while (ncap < (int64_t)h[0] * 4) ncap *= 2;
                thj_fusion* nb = nullptr;
                HIPCHK(hipMalloc(&nb, (size_t)ncap * sizeof(thj_fusion)));
                if (h[0]) HIPCHK(hipMemcpy(nb, c->d_fus, (size_t)h[0] * sizeof(thj_fusion), hipMemcpyDeviceToDevice));
                hipFree(c->d_fus);
                c->d_fus = nb; c->fus_cap = ncap;
            }
        }
    }
    FusionSink sink{c->d_fus, c->d_fus_count, (unsigned long long)c->fus_cap, (unsigned int*)(c->d_fus_count + 1),
                    c->d_fus_ignore, (uint32_t)c->n_fus_ignore};
    int64_t blocks = ((int64_t)b.n_reads + 255) / 256;
    if (blocks > 768) blocks = 768;         // three workgroups per CU (168 VGPRs); each walks ~n_tiles / 768 tiles, its candidate pairs pile up
    hipLaunchKernelGGL(thj_k_fusion, dim3((unsigned)blocks), dim3(256), 0, c->stream, g, p, b, sink);
    HIPCHK(hipGetLastError());
    if (!c->fus_probe_pending) {
        HIPCHK(hipMemcpyAsync(&c->h_pinned[44], c->d_fus_count, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipEventRecord(c->fus_probe_ev, c->stream));
        c->fus_probe_pending = true;
    }
    return THJ_OK;
}

// FusionSimpleSet (segment_juncs.cpp:2791-2803) on the device: the raw candidate events are ordered by Fusion::operator<
// (fusions.h:38-69: ref1, ref2, left, right, dir) with three stable radix sorts of an index (least significant key first), runs of
// equal keys are numbered by a prefix sum over their heads, and every event adds itself to its run's record (count += 1, the
// smallest edit distance wins).  Only the distinct fusions come down.  (Round 1 brought every raw event to the host and sorted
// there: 36 of the 62 ms of a fusion-search step on configs[3]'s shape.)
__global__ __launch_bounds__(256) void thj_k_fus_keys(const thj_fusion* ev, int64_t n, uint32_t* kdir, uint32_t* idx) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) { kdir[i] = ev[i].dir; idx[i] = (uint32_t)i; }
}
__global__ __launch_bounds__(256) void thj_k_fus_gather(const thj_fusion* ev, const uint32_t* idx, int64_t n, int which, u64* key) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const thj_fusion e = ev[idx[i]];
        key[i] = which == 0 ? ((u64)e.left << 32) | (u64)e.right : ((u64)e.ref_id1 << 32) | (u64)e.ref_id2;
    }
}
__global__ __launch_bounds__(256) void thj_k_fus_heads(const thj_fusion* ev, const uint32_t* idx, int64_t n, uint32_t* head) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t h = 1;
        if (i > 0) {
            const thj_fusion a = ev[idx[i - 1]], b = ev[idx[i]];
            h = (a.ref_id1 != b.ref_id1 || a.ref_id2 != b.ref_id2 || a.left != b.left || a.right != b.right || a.dir != b.dir) ? 1u : 0u;
        }
        head[i] = h;
    }
}
__global__ __launch_bounds__(256) void thj_k_fus_reduce(const thj_fusion* ev, const uint32_t* idx, const uint32_t* head, const uint32_t* run, int64_t n, thj_fusion* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const thj_fusion e = ev[idx[i]];
        thj_fusion* o = out + (run[i] - 1);                           // run[] = inclusive prefix sum of the heads
        if (head[i]) { o->ref_id1 = e.ref_id1; o->ref_id2 = e.ref_id2; o->left = e.left; o->right = e.right; o->dir = e.dir; o->reserved = 0; }
        atomicAdd(&o->count, 1u);
        atomicMin(&o->edit_dist, e.edit_dist);
    }
}

static int fusion_reduce_on_device(thj_ctx* c, int64_t n) {
    void *d_kdir = nullptr, *d_kdir2 = nullptr, *d_idx = nullptr, *d_idx2 = nullptr, *d_key = nullptr, *d_key2 = nullptr, *d_out = nullptr;
    auto release = [&]() { for (void* p : {d_kdir, d_kdir2, d_idx, d_idx2, d_key, d_key2, d_out}) if (p) thj_dev_release(c, p); };
    int rc = 0;
    for (void** p : {&d_kdir, &d_kdir2, &d_idx, &d_idx2}) if (!rc) rc = thj_dev_alloc(c, p, (size_t)n * 4);
    for (void** p : {&d_key, &d_key2}) if (!rc) rc = thj_dev_alloc(c, p, (size_t)n * 8);
    if (!rc) rc = thj_dev_alloc(c, &d_out, (size_t)n * sizeof(thj_fusion));
    if (rc) { release(); return rc; }
    size_t need32 = 0, need64 = 0, needscan = 0;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need32, (const uint32_t*)d_kdir, (uint32_t*)d_kdir2, (const uint32_t*)d_idx, (uint32_t*)d_idx2, (int)n, 0, 8, c->stream));
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, need64, (const u64*)d_key, (u64*)d_key2, (const uint32_t*)d_idx, (uint32_t*)d_idx2, (int)n, 0, 64, c->stream));
    HIPCHK(hipcub::DeviceScan::InclusiveSum(nullptr, needscan, (const uint32_t*)d_kdir, (uint32_t*)d_kdir2, (int)n, c->stream));
    size_t need = std::max(need32, std::max(need64, needscan));
    if (need > c->sort_tmp_bytes) { HIPCHK(hipStreamSynchronize(c->stream)); hipFree(c->d_sort_tmp); c->d_sort_tmp = nullptr; HIPCHK(hipMalloc(&c->d_sort_tmp, need)); c->sort_tmp_bytes = need; }
    int64_t grid = (n + 255) / 256; if (grid > 8192) grid = 8192;
    const thj_fusion* ev = c->d_fus;
    size_t tmp = c->sort_tmp_bytes;
    hipLaunchKernelGGL(thj_k_fus_keys, dim3((unsigned)grid), dim3(256), 0, c->stream, ev, n, (uint32_t*)d_kdir, (uint32_t*)d_idx);
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, tmp, (const uint32_t*)d_kdir, (uint32_t*)d_kdir2, (const uint32_t*)d_idx, (uint32_t*)d_idx2, (int)n, 0, 8, c->stream));
    hipLaunchKernelGGL(thj_k_fus_gather, dim3((unsigned)grid), dim3(256), 0, c->stream, ev, (const uint32_t*)d_idx2, n, 0, (u64*)d_key);
    tmp = c->sort_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, tmp, (const u64*)d_key, (u64*)d_key2, (const uint32_t*)d_idx2, (uint32_t*)d_idx, (int)n, 0, 64, c->stream));
    hipLaunchKernelGGL(thj_k_fus_gather, dim3((unsigned)grid), dim3(256), 0, c->stream, ev, (const uint32_t*)d_idx, n, 1, (u64*)d_key);
    tmp = c->sort_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, tmp, (const u64*)d_key, (u64*)d_key2, (const uint32_t*)d_idx, (uint32_t*)d_idx2, (int)n, 0, 64, c->stream));
    // d_idx2: the events in Fusion::operator< order
    hipLaunchKernelGGL(thj_k_fus_heads, dim3((unsigned)grid), dim3(256), 0, c->stream, ev, (const uint32_t*)d_idx2, n, (uint32_t*)d_kdir);
    tmp = c->sort_tmp_bytes;
    HIPCHK(hipcub::DeviceScan::InclusiveSum(c->d_sort_tmp, tmp, (const uint32_t*)d_kdir, (uint32_t*)d_kdir2, (int)n, c->stream));
    uint32_t n_runs = 0;
    HIPCHK(hipMemcpyAsync(&n_runs, (const uint32_t*)d_kdir2 + (n - 1), 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    // count = 0, edit_dist = all ones, the rest is written by each run's head
    HIPCHK(hipMemsetAsync(d_out, 0, (size_t)n_runs * sizeof(thj_fusion), c->stream));
    HIPCHK(hipMemset2DAsync((char*)d_out + offsetof(thj_fusion, edit_dist), sizeof(thj_fusion), 0xFF, 4, n_runs, c->stream));
    hipLaunchKernelGGL(thj_k_fus_reduce, dim3((unsigned)grid), dim3(256), 0, c->stream, ev, (const uint32_t*)d_idx2, (const uint32_t*)d_kdir, (const uint32_t*)d_kdir2, n, (thj_fusion*)d_out);
    HIPCHK(hipGetLastError());
    // the set stays on the device (thj_span_fusions_from_segjuncs hands it to the spanning stage); it comes down when asked for
    c->d_fus_out = (thj_fusion*)d_out; d_out = nullptr;
    c->n_fus_out = (int64_t)n_runs;
    c->h_fus_stale = true;
    release();
    return THJ_OK;
}

int thj_fusions_to_host(thj_ctx* c) {
    if (!c->h_fus_stale) return THJ_OK;
    c->h_fusions.resize((size_t)c->n_fus_out);
    if (c->n_fus_out) HIPCHK(hipMemcpyAsync(c->h_fusions.data(), c->d_fus_out, (size_t)c->n_fus_out * sizeof(thj_fusion), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->h_fus_stale = false;
    return THJ_OK;
}
static void fusion_drop_device_set(thj_ctx* c) {
    if (c->d_fus_out) thj_dev_release(c, c->d_fus_out);
    c->d_fus_out = nullptr; c->n_fus_out = 0; c->h_fus_stale = false;
}

extern "C" int thj_fusion_finish(thj_ctx* c, int64_t* n_fusions) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    c->h_fusions.clear();
    fusion_drop_device_set(c);
    if (c->d_fus_count) {
        unsigned long long h[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(h, c->d_fus_count, 16, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if ((unsigned int)h[1]) {
            // a batch added more than the buffer had room for (thj_fusion_run_async grows it ahead of the count, but only between batches).  The
            // counter kept counting, so the need is known: make the buffer that large and ask for the pass again
            const int64_t need = (int64_t)h[0], ncap = need + need / 4 + 4096;
            if (need >= (1ll << 30)) { thj_set_error("fusion event buffer overflow (%llu candidate events, capacity %lld)", h[0], (long long)c->fus_cap); return THJ_EOVERFLOW; }
            thj_fusion* nb = nullptr;
            HIPCHK(hipMalloc(&nb, (size_t)ncap * sizeof(thj_fusion)));
            hipFree(c->d_fus);
            c->d_fus = nb; c->fus_cap = ncap;
            thj_set_error("the buffer for raw fusion candidates was too small (%lld needed); it has been enlarged: run the pass again "
                          "(thj_fusion_reset_async, the thj_fusion_run_async calls, thj_fusion_finish)", (long long)need);
            return THJ_ERETRY;
        }
        static const bool on_host = getenv("THJ_FUSION_REDUCE_ON_HOST") != nullptr;
        if (h[0] && !on_host && h[0] < (1ull << 31)) {
            const int rc = fusion_reduce_on_device(c, (int64_t)h[0]);
            if (rc) return rc;
        } else if (h[0]) {
        std::vector<thj_fusion> ev((size_t)h[0]);
        HIPCHK(hipMemcpy(ev.data(), c->d_fus, (size_t)h[0] * sizeof(thj_fusion), hipMemcpyDeviceToHost));
        // the same on the host: count the occurrences, keep the smallest edit distance; iteration order = Fusion::operator<
        auto less = [](const thj_fusion& a, const thj_fusion& b) {
            if (a.ref_id1 != b.ref_id1) return a.ref_id1 < b.ref_id1;
            if (a.ref_id2 != b.ref_id2) return a.ref_id2 < b.ref_id2;
            if (a.left != b.left) return a.left < b.left;
            if (a.right != b.right) return a.right < b.right;
            return a.dir < b.dir;
        };
        std::sort(ev.begin(), ev.end(), less);
        for (auto& e : ev) {
            if (!c->h_fusions.empty() && !less(c->h_fusions.back(), e) && !less(e, c->h_fusions.back())) {
                c->h_fusions.back().count += 1;
                if (e.edit_dist < c->h_fusions.back().edit_dist) c->h_fusions.back().edit_dist = e.edit_dist;
            } else c->h_fusions.push_back(e);
        }
        }
    }
    if (n_fusions) *n_fusions = c->h_fus_stale ? c->n_fus_out : (int64_t)c->h_fusions.size();
    return THJ_OK;
}

extern "C" int thj_fusion_download(thj_ctx* c, thj_fusion* out) {
    if (!c) { thj_set_error("thj_fusion_download: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    { const int rc = thj_fusions_to_host(c); if (rc) return rc; }
    if (!c->h_fusions.empty() && !out) { thj_set_error("thj_fusion_download: bad argument"); return THJ_EINVAL; }
    if (!c->h_fusions.empty()) memcpy(out, c->h_fusions.data(), c->h_fusions.size() * sizeof(thj_fusion));
    return THJ_OK;
}

extern "C" int thj_segjuncs_device_insertions(thj_ctx* c, const uint64_t** d_keys, const uint64_t** d_vals, int64_t* n) {
    if (!c || !d_keys || !d_vals || !n) { thj_set_error("thj_segjuncs_device_insertions: bad argument"); return THJ_EINVAL; }
    *d_keys = (const uint64_t*)c->d_ins_key_sorted; *d_vals = (const uint64_t*)c->d_ins_val_sorted; *n = c->n_ins;
    return THJ_OK;
}

extern "C" int thj_segjuncs_merge_insertions_async(thj_ctx* c, const uint64_t* d_keys, const uint64_t* d_vals, int64_t n) {
    if (!c || (n > 0 && (!d_keys || !d_vals))) { thj_set_error("thj_segjuncs_merge_insertions_async: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return THJ_OK;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(thj_k_merge_ins, dim3((unsigned)blocks), dim3(256), 0, c->stream, c->d_ins_key, c->d_ins_val,
                       (u64)c->indel_cap - 1, (const u64*)d_keys, (const u64*)d_vals, n, &c->d_cnt[CNT_INS], &c->d_ovf[2], ins_list(c));
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

static int x_finish_check(thj_ctx* c, const unsigned int* ovf_now);
static unsigned long long* x_host_headers(thj_ctx* c, const u64** d_hdr, size_t* bytes);

extern "C" int thj_segjuncs_finish(thj_ctx* c, thj_segjuncs_counts* counts) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    for (;;) {
        // the distinct events are already dense (see set_insert): ONE round trip brings their counts -- and, after an
        // exchange step, the headers every rank gathered
        HIPCHK(hipMemcpyAsync(&c->h_pinned[4], c->d_ovf, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipMemcpyAsync(&c->h_pinned[8], c->d_cnt, CNT_N * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        if (c->xchg) {
            const u64* d_hdr = nullptr; size_t hb = 0;
            unsigned long long* h = x_host_headers(c, &d_hdr, &hb);
            HIPCHK(hipMemcpyAsync(h, d_hdr, hb, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        const unsigned int* ovf = (const unsigned int*)&c->h_pinned[4];
        if (ovf[3]) {            // (before the exchange step's own look at the flags: a full task list has lost tasks on this rank whatever the others say)
            c->xchg = nullptr;
            thj_set_error("the task list of a thj_segjuncs_run_async call filled up (more than two window / indel tasks per read of the batch): split the batch");
            return THJ_EOVERFLOW;
        }
        if (c->xchg) {
            const int rc = x_finish_check(c, ovf);
            if (rc < 0) return rc;
            if (rc > 0) continue;                 // the step was repeated (larger message / larger table): look again
        } else if (ovf[0] || ovf[1] || ovf[2]) {
            thj_set_error("event table overflow (junc=%u del=%u ins=%u): call thj_segjuncs_configure with larger capacities and re-run",
                          ovf[0], ovf[1], ovf[2]);
            return THJ_EOVERFLOW;
        }
        c->h_pinned[0] = c->h_pinned[8 + CNT_JUNC]; c->h_pinned[1] = c->h_pinned[8 + CNT_DEL]; c->h_pinned[2] = c->h_pinned[8 + CNT_INS];
        c->n_junc = (int64_t)c->h_pinned[0];
        c->n_del = (int64_t)c->h_pinned[1];
        c->n_ins = (int64_t)c->h_pinned[2];
        // a pass that left a table more than 40 % full (merged key sets of other ranks, a large coverage pass, one huge
        // batch) moves to tables four times the size now: the rehash keeps every key, and the next pass starts with room
        const bool gj = c->n_junc * 5 > c->junc_cap * 2;
        const bool gi = (c->n_del > c->n_ins ? c->n_del : c->n_ins) * 5 > c->indel_cap * 2;
        if (gj || gi) { int rc = grow_tables(c, gj, gi); if (rc) return rc; continue; }
        break;
    }
    // sorted output (stream-ordered: consumers on the context stream need no further synchronisation).  Only the bits a key of this
    // genome can have are sorted on (a radix pass is a kernel or two, and these lists are short: the passes are launch gaps, not work):
    // junction / deletion keys = global position + 1 above 30 bits of length and strand, insertion keys = the same above 4 bits of length
    int pos_bits = 1;
    while (pos_bits < 34 && (1ll << pos_bits) <= c->n_blocks * 64 + 1) ++pos_bits;
    const int junc_bits = 30 + pos_bits > 64 ? 64 : 30 + pos_bits, ins_bits = 4 + pos_bits;
    size_t tmp = c->sort_tmp_bytes;
    if (c->n_junc > 0)
        HIPCHK(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, tmp, (const u64*)junc_list(c), c->d_junc_sorted, c->n_junc, 0, junc_bits, c->stream));
    tmp = c->sort_tmp_bytes;
    if (c->n_del > 0)
        HIPCHK(hipcub::DeviceRadixSort::SortKeys(c->d_sort_tmp, tmp, (const u64*)del_list(c), c->d_del_sorted, c->n_del, 0, junc_bits, c->stream));
    if (c->n_ins > 0) {
        int64_t blocks = (c->n_ins + 255) / 256; if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(thj_k_ins_gather, dim3((unsigned)blocks), dim3(256), 0, c->stream, (const u64*)ins_list(c), c->n_ins,
                           (const u64*)c->d_ins_key, (const u64*)c->d_ins_val, c->d_tmp_keys2, c->d_tmp_vals);
        tmp = c->sort_tmp_bytes;
        HIPCHK(hipcub::DeviceRadixSort::SortPairs(c->d_sort_tmp, tmp, (const u64*)c->d_tmp_keys2, c->d_ins_key_sorted,
                                                  (const u64*)c->d_tmp_vals, c->d_ins_val_sorted, c->n_ins, 0, ins_bits, c->stream));
    }
    if (counts) {
        const unsigned long long* cnt = &c->h_pinned[8];
        counts->n_juncs = c->n_junc; counts->n_deletions = c->n_del; counts->n_insertions = c->n_ins;
        counts->n_windows = (int64_t)cnt[CNT_WINDOWS]; counts->n_indel_pairs = (int64_t)cnt[CNT_INDEL_PAIRS];
        counts->n_rescue_pairs = (int64_t)cnt[CNT_RESCUE_PAIRS]; counts->n_overflow_blocks = (int64_t)cnt[CNT_OVF_BLOCKS];
        counts->n_hits_read = (int64_t)cnt[CNT_HITS];
    }
    return THJ_OK;
}

extern "C" int thj_segjuncs_device_keys(thj_ctx* c, int kind, const uint64_t** d_keys, int64_t* n) {
    if (!c || !d_keys || !n || (kind != 0 && kind != 1)) { thj_set_error("thj_segjuncs_device_keys: bad argument"); return THJ_EINVAL; }
    *d_keys = kind == 0 ? (const uint64_t*)c->d_junc_sorted : (const uint64_t*)c->d_del_sorted;
    *n = kind == 0 ? c->n_junc : c->n_del;
    return THJ_OK;
}

static void decode_gpos(const thj_ctx* c, uint64_t gpos1, uint32_t* ref_id, uint32_t* pos) {
    // gpos1 = contig_blk[ref-1]*64 + pos + 1, pos >= -1
    int64_t gp = (int64_t)gpos1 - 1;
    int lo = 0, hi = c->n_contigs;      // last contig whose start <= gp (+1 tolerance for pos = -1)
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if ((int64_t)c->h_contig_blk[mid] * 64 <= gp + 1) lo = mid; else hi = mid;
    }
    // pos = -1 of contig k would alias the guard block of contig k-1; guard blocks make that unambiguous
    *ref_id = (uint32_t)lo + 1;
    *pos = (uint32_t)(int32_t)(gp - (int64_t)c->h_contig_blk[lo] * 64);
}

extern "C" int thj_segjuncs_download(thj_ctx* c, thj_junction* juncs, thj_junction* dels, thj_insertion* ins) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    std::vector<uint64_t> k, v;
    auto get = [&](const u64* d, int64_t n, std::vector<uint64_t>& h) -> int {
        h.resize((size_t)n);
        if (n) HIPCHK(hipMemcpyAsync(h.data(), d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        return THJ_OK;
    };
    int rc;
    if (c->n_junc && !juncs) { thj_set_error("null juncs buffer"); return THJ_EINVAL; }
    if ((rc = get(c->d_junc_sorted, c->n_junc, k))) return rc;
    for (int64_t i = 0; i < c->n_junc; ++i) {
        uint32_t ref, pos;
        decode_gpos(c, k[i] >> 30, &ref, &pos);
        uint32_t len = (uint32_t)((k[i] >> 1) & ((1ull << 29) - 1));
        juncs[i].ref_id = ref; juncs[i].left = pos; juncs[i].right = pos + len; juncs[i].antisense = (uint32_t)(k[i] & 1);
    }
    if (c->n_del && !dels) { thj_set_error("null deletions buffer"); return THJ_EINVAL; }
    if ((rc = get(c->d_del_sorted, c->n_del, k))) return rc;
    for (int64_t i = 0; i < c->n_del; ++i) {
        uint32_t ref, pos;
        decode_gpos(c, k[i] >> 30, &ref, &pos);
        uint32_t len = (uint32_t)((k[i] >> 1) & ((1ull << 29) - 1));
        dels[i].ref_id = ref; dels[i].left = pos; dels[i].right = pos + len; dels[i].antisense = 0;
    }
    if (c->n_ins && !ins) { thj_set_error("null insertions buffer"); return THJ_EINVAL; }
    if ((rc = get(c->d_ins_key_sorted, c->n_ins, k))) return rc;
    if ((rc = get(c->d_ins_val_sorted, c->n_ins, v))) return rc;
    static const char code[8] = {'A', 'C', 'G', 'T', 'N', '?', '?', '?'};
    for (int64_t i = 0; i < c->n_ins; ++i) {
        uint32_t ref, pos;
        decode_gpos(c, k[i] >> 4, &ref, &pos);
        int len = (int)(k[i] & 15);
        memset(&ins[i], 0, sizeof ins[i]);
        ins[i].ref_id = ref; ins[i].left = pos;
        uint32_t seq = (uint32_t)(v[i] & ((1u << INS_SEQ_BITS) - 1u));
        for (int b = 0; b < len && b < 7; ++b) ins[i].seq[b] = code[(seq >> (3 * b)) & 7];
        ins[i].prio = v[i] >> INS_SEQ_BITS;
    }
    return THJ_OK;
}

// ------------------------------------------------------------------ juncs_db gather

struct DevPiece { uint32_t ref_id; int32_t start, len; uint32_t flags; };
static_assert(sizeof(DevPiece) == sizeof(thj_piece), "piece layout");

__global__ __launch_bounds__(256) void thj_k_gather_pieces(Genome g, const DevPiece* pieces, int64_t n, const int64_t* out_off, char* out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const DevPiece pc = pieces[i];
        piece_text(g, pc.ref_id, pc.start, pc.len, (pc.flags & THJ_PIECE_RC) != 0, out + out_off[i]);
    }
}

extern "C" int thj_genome_gather(thj_ctx* c, const thj_piece* pieces, int64_t n, const int64_t* out_off, char* out, int64_t out_bytes) {
    if (!c || n < 0 || (n > 0 && (!pieces || !out_off || !out)) || out_bytes < 0) { thj_set_error("thj_genome_gather: bad argument"); return THJ_EINVAL; }
    if (!c->d_blocks) { thj_set_error("no genome resident: call thj_genome_upload/adopt first"); return THJ_ESTATE; }
    if (n == 0) return THJ_OK;
    for (int64_t i = 0; i < n; ++i) {
        const thj_piece& p = pieces[i];
        if (p.ref_id == 0 || (int32_t)p.ref_id > c->n_contigs || p.start < 0 || p.len < 0 || (int64_t)p.start + p.len > c->h_lens[p.ref_id - 1]) {
            thj_set_error("piece %lld lies outside contig %u", (long long)i, p.ref_id); return THJ_EINVAL;
        }
        if (out_off[i] < 0 || out_off[i] + p.len > out_bytes) { thj_set_error("piece %lld does not fit the output buffer", (long long)i); return THJ_EINVAL; }
    }
    HIPCHK(hipSetDevice(c->device));
    DevPiece* d_p = nullptr; int64_t* d_off = nullptr; char* d_out = nullptr;
    HIPCHK(hipMalloc(&d_p, (size_t)n * sizeof(DevPiece)));
    HIPCHK(hipMalloc(&d_off, (size_t)n * 8));
    HIPCHK(hipMalloc(&d_out, (size_t)(out_bytes ? out_bytes : 1)));
    HIPCHK(hipMemcpyAsync(d_p, pieces, (size_t)n * sizeof(DevPiece), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_off, out_off, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    Genome g{c->d_blocks, c->d_contig_blk, c->d_contig_len, c->n_contigs};
    int64_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(thj_k_gather_pieces, dim3((unsigned)blocks), dim3(256), 0, c->stream, g, (const DevPiece*)d_p, n, (const int64_t*)d_off, d_out);
    // only the bytes the pieces cover are defined on the device; the caller's other bytes (headers, newlines) stay as they are
    std::vector<char> tmp((size_t)out_bytes);
    HIPCHK(hipMemcpyAsync(tmp.data(), d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < n; ++i) memcpy(out + out_off[i], tmp.data() + out_off[i], (size_t)pieces[i].len);
    hipFree(d_p); hipFree(d_off); hipFree(d_out);
    return THJ_OK;
}

#include "thj_covsearch_impl.h"
#include "thj_exchange_impl.h"

static unsigned long long* x_host_headers(thj_ctx* c, const u64** d_hdr, size_t* bytes) {
    *d_hdr = c->xchg->d_hdr; *bytes = (size_t)c->xchg->n * 64;
    return c->xchg->h_hdr;
}
