// thj_ctx.h -- the context object shared by the translation units of libthj_hip.so (internal)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <utility>
#include <vector>

#include "../../include/thj.h"
#include "thj_core.h"
#include "thj_internal.h"

using thj::u64;

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            thj_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return THJ_EHIP;                                                                  \
        }                                                                                     \
    } while (0)

// a span batch this library made (thj_span_batch_upload, thj_ingest_span_*): the API descriptor first, then what it owns
struct OwnedSpanBatch {
    thj_span_batch desc;
    void* ptrs[8];               // 0 seg_off, 1 hits, 2 read planes, 3 read lengths, 4 qualities, 5 hit heads,
                                 // 6 the reads' own (inflated) BAM records, 7 uint32 per row: where its record starts in [6]
    size_t reads_infl_bytes;     // size of [6]
};

struct thj_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // the side streams for everything that runs beside the context's stream: [0] / [1] = the first / second batch of a pair call, in
    // stage 1 (sj_launch in thj_segjuncs.hip) and in stage 2 (span_stream below points at the same); [2] only behind developer switches.
    // Two in use, not one per chain: HIP spreads streams over GPU_MAX_HW_QUEUES (4) hardware queues round robin, two streams on one
    // queue run one after the other, and a process has other streams too -- with ten streams the second side's chain of stage 1 sat
    // behind the first side's (profiles/r05_e_timeline.txt)
    hipStream_t aux_stream[3] = {}; hipEvent_t aux_ev[10] = {};
    // what the probe measured when it took side stream k: a spin kernel on it beside one on the context's stream and on every side stream
    // taken before, over a spin kernel alone (1.0 = fully beside each other, 2.0 = one queue shared); 0 = not measured.  aux_independent
    // = measured and below 1.5 (thj_ctx_stream_info, thj_streams.hip)
    double aux_ratio[3] = {0, 0, 0}; bool aux_independent[3] = {false, false, false};
    // genome
    const u64* d_blocks = nullptr; bool own_blocks = false;
    uint32_t* d_contig_blk = nullptr; int32_t* d_contig_len = nullptr;
    std::vector<uint32_t> h_contig_blk; std::vector<int64_t> h_lens;
    int32_t n_contigs = 0; int64_t n_blocks = 0;
    // tables
    int64_t junc_cap = 0, indel_cap = 0;
    u64 *d_junc = nullptr, *d_del = nullptr, *d_ins_key = nullptr, *d_ins_val = nullptr;
    unsigned int* d_ovf = nullptr;
    unsigned long long* d_cnt = nullptr;
    // sorted outputs
    u64 *d_junc_sorted = nullptr, *d_del_sorted = nullptr, *d_ins_key_sorted = nullptr, *d_ins_val_sorted = nullptr;
    u64 *d_tmp_keys = nullptr, *d_tmp_vals = nullptr, *d_tmp_keys2 = nullptr;   // event lists (see set_insert) / insertion gather
    int64_t out_cap_junc = 0, out_cap_indel = 0;
    unsigned long long* d_out_n = nullptr;      // [3]
    unsigned long long* h_pinned = nullptr;     // [16] pinned staging
    void* d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
    int64_t n_junc = 0, n_del = 0, n_ins = 0;
    hipEvent_t probe_ev = nullptr; bool probe_pending = false;            // insert counters on their way to h_pinned[32..]
    uint8_t* d_fus_ignore = nullptr; int64_t n_fus_ignore = 0;            // --fusion-ignore-chromosomes flags per ref id
    uint32_t* d_rescue_list[2] = {}; int64_t rescue_list_cap[2] = {};      // reads taking the mate-anchored rescue + per-workgroup counts
    uint32_t* d_many[2] = {};                                            // reads with many hits of a launch (thj_k_segjuncs_shared): count, list
    void* d_sj_lists[2] = {}; size_t sj_lists_cap[2] = {};                  // the flat kernels' task / rescue / general-read lists (thj_k_sj_flat)
    int32_t* d_rescue_slots[2] = {};                                    // rescue outcomes of reads with many hits (thj_k_segjuncs_rescue)
    // long_spanning_reads (thj_span.hip)
    uint32_t* d_junc_bucket = nullptr; int64_t n_junc_buckets = 0;     // coarse index over d_span_junc (junc_range)
    u64* d_span_cat = nullptr;                                            // junction ++ deletion keys before their sort
    u64* d_span_junc = nullptr; int64_t n_span_junc = 0; int64_t cap_span_junc = 0;
    u64* d_span_ins_key = nullptr; uint32_t* d_span_ins_seq = nullptr; int64_t n_span_ins = 0; int64_t cap_span_ins = 0;
    void* d_huge_ws = nullptr; uint32_t* d_huge_list = nullptr; int huge_blocks = 0, huge_list_cap = 0;      // reads with too many joined alignments for a thread's array (thj_k_stitch_huge)
    void* d_span_fus = nullptr; int64_t n_span_fus = 0; int64_t cap_span_fus = 0;       // --fusion-search: the sorted .fusions list (thj_span_fusions_upload)
    void* d_aln_pool = nullptr; void* d_aln_sorted = nullptr; int64_t aln_cap = 0;
    u64* d_aln_keys = nullptr;
    unsigned long long* d_aln_count = nullptr; unsigned int* d_span_status = nullptr;
    int64_t n_alns = 0, n_ovf = 0, span_reads = 0, ovf_cap = 0;
    uint8_t* d_nrec = nullptr;
    std::vector<thj_aln> h_alns;
    // scratch of a batch in flight (span_launch in thj_span.hip): two sets, so that the two sides of a pass can run beside each other
    struct SpanSet { uint32_t* d_worklist = nullptr; int64_t worklist_cap = 0; void* d_ent = nullptr; int64_t ent_cap = 0; void* d_joined = nullptr; int64_t joined_cap = 0; uint32_t* d_defer = nullptr; int64_t defer_cap = 0; };
    SpanSet span_set[2]; int span_last_set = 0;
    // thj_span_tier0_pair_async ran the pair's tier 0 ahead of thj_span_run_pair_async (which then only does what is behind it)
    bool span_t0_pending = false; int64_t span_t0_n[2] = {0, 0}; hipEvent_t span_t0_prof[2][32] = {};
    hipStream_t span_stream[3] = {}; hipEvent_t span_ev[10] = {}; bool span_stream_own = false;      // (= aux_stream unless THJ_SPAN_PRIO)
    bool span_profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> span_prof_events;
    // coverage search (thj_covsearch_impl.h)
    u64* d_cov = nullptr; int32_t* d_cov_size = nullptr;                  // 8 bitmaps of n_blocks words; max(right) + 1 per contig
    uint32_t* d_ext_key = nullptr; u64* d_ext_val = nullptr; uint32_t* d_ext_key_sorted = nullptr; u64* d_ext_val_sorted = nullptr;
    uint32_t* d_ext_off = nullptr; int64_t n_ext = 0, ext_cap = 0, ext_sorted_cap = 0;        // extension table of the unmapped reads
    unsigned long long* d_cov_found = nullptr;
    u64* d_cov_filter = nullptr; int64_t cov_filter_bytes = 0;          // Bloom filter over the extension table
    u64* d_cov_jkey = nullptr; uint32_t* d_cov_jskip = nullptr; int64_t cov_jcap = 0;      // junctions found: key, skip count
    u64* d_cov_jkey2 = nullptr; uint32_t* d_cov_jskip2 = nullptr;                           // ... sort buffers for the cut
    void* d_mx_cand = nullptr; int64_t n_mx_cand = 0, mx_cand_cap = 0;        // microexon search: candidate windows of the pass's reads
    u64 cov_filter_mask = 0; int32_t cov_min_intron = 0, cov_max_intron = 0; bool cov_pending = false;
    // fusion search
    thj_fusion* d_fus = nullptr; unsigned long long* d_fus_count = nullptr; int64_t fus_cap = 0;
    hipEvent_t fus_probe_ev = nullptr; bool fus_probe_pending = false;      // the raw event count on its way to h_pinned[44] (the buffer grows ahead of it)
    std::vector<thj_fusion> h_fusions;
    thj_fusion* d_fus_out = nullptr; int64_t n_fus_out = 0;            // the reduced set on the device (Fusion::operator< order); null: only h_fusions holds it
    bool h_fus_stale = false;                                          // h_fusions not yet copied down from d_fus_out
    // batch arrays come and go once per shard / batch: hipMalloc and (synchronising) hipFree per array cost more than the
    // kernels of a small shard, so released blocks are kept and handed out again (thj_dev_alloc / thj_dev_release)
    struct DevBlock { void* p; size_t cap; bool used; };
    std::vector<DevBlock> dev_cache; size_t dev_cache_bytes = 0;
    // device-side ingest scratch (thj_ingest.hip)
    void* d_ing0 = nullptr; size_t ing_cap0 = 0; void* d_ing1 = nullptr; size_t ing_cap1 = 0;
    // the BAM writer's device side (thj_bamout.hip): the pass's encoded records, their offsets; the deflater's scratch
    uint8_t* d_bam = nullptr; size_t bam_cap = 0; int64_t bam_bytes = 0;
    void* d_bam_tmp = nullptr; size_t bam_tmp_cap = 0;
    void* d_infl_tmp = nullptr; size_t infl_tmp_cap = 0;                  // token streams of the two-kernel inflater
    // junction consensus (thj_juncbed_impl.h)
    u64* d_jb_key = nullptr; uint32_t* d_jb_u32 = nullptr; u64* d_jb_list = nullptr; u64* d_jb_sorted = nullptr;
    unsigned long long* d_jb_cnt = nullptr; void* d_jb_occ = nullptr;
    int64_t jb_cap = 0, jb_occ_cap = 0, jb_occ_used = 0, jb_want = 0;
    std::vector<thj_juncstat> h_jb;
    // multi-GPU exchange step pending a look at its gathered headers (thj_exchange_impl.h)
    struct thj_comm* xchg = nullptr;
    // profiling
    bool profile = false;
    bool serial_launch = false;                 // thj_profile_serial: no side streams
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<hipEvent_t> prof_all; std::vector<int> prof_sets;       // every event of prof_events once; the scratch set of each profiled launch
    std::vector<hipEvent_t> event_pool;
};
hipEvent_t thj_get_event(struct thj_ctx* c);
int thj_ensure_aux_streams(struct thj_ctx* c, int need);            // thj_streams.hip: aux_stream[0 .. need), aux_ev[10]
void thj_warm_span(hipStream_t s); void thj_warm_ingest(hipStream_t s); void thj_warm_bamout(hipStream_t s);      // one empty launch from the translation unit: its code object is loaded now
int thj_dev_alloc(struct thj_ctx* c, void** out, size_t bytes);     // like hipMalloc, from the context's block cache
void thj_dev_release(struct thj_ctx* c, void* p);                  // like hipFree, but the block stays with the context
void thj_dev_cache_free(struct thj_ctx* c);
void thj_span_free(struct thj_ctx* c);
int thj_span_compact_device(struct thj_ctx* c, void** d_out);       // thj_span.hip: the pass's records, ordered, on the device
void thj_bamout_free(struct thj_ctx* c);                            // thj_bamout.hip

int thj_fusions_to_host(struct thj_ctx* c);        // h_fusions <- d_fus_out when it has not come down yet (thj_segjuncs.hip)
