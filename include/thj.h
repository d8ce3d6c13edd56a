/*
 * thj.h -- C ABI of libthj_hip.so: the MI355X-native junction-discovery hot path
 * that sits behind TopHat's `segment_juncs` and `long_spanning_reads` binaries.
 *
 * The reference (DaehwanKimLab/tophat v2.1.2) has no plugin / FFI surface for
 * this path: its boundary is process + argv + files (tophat.py:3070-3129 and
 * :3133-3200).  The drop-in executables keep that boundary; this header is the
 * thin C ABI between their C++ host code and the HIP kernels, and it is what a
 * binding in any other host language would bind.  Each entry point names the
 * reference code it replaces.
 *
 * Conventions: extern "C"; plain pointers and sizes; every function returns 0
 * on success and a negative THJ_E* code on failure with a message available
 * from thj_last_error() (thread-local); opaque handles; caller-owned buffers;
 * one context per (host thread, device); no global state.  Functions whose
 * name ends in _async only enqueue work on the context's HIP stream.
 * Pointers documented as DEVICE pointers must be HIP device allocations on the
 * context's device (e.g. hipMalloc or a torch tensor's data_ptr()).
 */
#ifndef THJ_H
#define THJ_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THJ_OK            0
#define THJ_EINVAL       -1   /* bad argument / unsupported parameter value */
#define THJ_ENOMEM       -2
#define THJ_EHIP         -3   /* a HIP runtime call failed */
#define THJ_EOVERFLOW    -4   /* an event table filled up; re-configure larger and re-run */
#define THJ_ESTATE       -5   /* call sequence violated (e.g. run before genome upload) */
#define THJ_ERETRY       -7   /* a device pool was too small for this pass and has been enlarged: make the same calls again */
#define THJ_EFALLBACK    -6   /* the device-side ingest cannot take this input (records straddle BGZF members, reads > 256 bases ...):
                                 nothing was done, use the host readers */

typedef struct thj_ctx thj_ctx;

/* One alignment record: the fields of BowtieHit (bwt_map.h:36-536) the path
 * reads, produced by BAMHitFactory::get_hit_from_buf (bwt_map.cpp:1101-1452).
 * `right` = BowtieHit::right() (bwt_map.h:213-243), `read_len` =
 * BowtieHit::read_len() (bwt_map.h:141-163). 16 bytes. */
typedef struct {
    uint32_t ref_id;      /* 1-based, @SQ order of --sam-header (bwt_map.h:608-632) */
    int32_t  left;
    int32_t  right;
    uint8_t  flags;       /* THJ_HIT_* */
    uint8_t  edit_dist;
    uint8_t  mismatches;
    uint8_t  read_len;
} thj_hit;

#define THJ_HIT_ANTISENSE 1u   /* BowtieHit::antisense_align() */
#define THJ_HIT_END       2u   /* BowtieHit::end() (bwt_map.cpp:1133-1140) */

/* The globals of common.cpp:79-180 that change hot-path results (SURVEY.md
 * Appendix A).  All int32, same order as tophat_amd/params.py. */
typedef struct {
    int32_t segment_length;
    int32_t segment_mismatches;
    int32_t min_segment_intron;
    int32_t max_segment_intron;
    int32_t max_insertion_length;
    int32_t max_deletion_length;
    int32_t max_seg_multihits;
    int32_t inner_dist_mean;
    int32_t inner_dist_std_dev;
    int32_t library_type;          /* common.h:155-167 */
    int32_t bowtie2;
    int32_t read_side;             /* segments.h:13-18: 1 READ_LEFT, 2 READ_RIGHT */
    int32_t min_report_intron;
    int32_t max_report_intron;
    int32_t min_anchor_len;
    int32_t read_mismatches;
    int32_t read_gap_length;
    int32_t read_edit_dist;
    int32_t bowtie2_max_penalty;
    int32_t bowtie2_min_penalty;
    int32_t bowtie2_penalty_for_N;
    int32_t bowtie2_read_gap_open;
    int32_t bowtie2_read_gap_cont;
    int32_t bowtie2_ref_gap_open;
    int32_t bowtie2_ref_gap_cont;
    int32_t fusion_anchor_length;  /* common.cpp:172 */
    int32_t fusion_min_dist;       /* common.cpp:173 */
    int32_t fusion_search;         /* common.cpp:171 (--fusion-search); long_spanning_reads only: thj_span_run_async */
} thj_params;

/* Fills `p` with the defaults of common.cpp:79-180. */
void thj_params_default(thj_params* p);

const char* thj_last_error(void);
const char* thj_version(void);

/* ------------------------------------------------------------ context */

/* Number of HIP devices visible to the process (< 0: error). */
int  thj_device_count(void);
/* `stream` is a hipStream_t to launch on (e.g. torch's current stream) or NULL
 * to let the context create its own non-blocking stream.  Work that runs beside that stream (the side chains of
 * thj_segjuncs_run*_async and thj_span_run*_async) runs on two side streams of the context's own, made at the first such
 * call and chosen by a ~2 ms measurement so that they do not share a hardware queue with `stream` or each other (HIP maps
 * streams to GPU_MAX_HW_QUEUES queues round robin); every call returns with its work joined on `stream` again. */
int  thj_ctx_create(int device, void* stream, thj_ctx** out);
/* Loads the kernels of the named parts now (an empty launch from each translation unit) instead of at their first use; blocks
 * until they are there.  For a process that has something else to do meanwhile.  No effect on results. */
#define THJ_WARM_SEGJUNCS 1   /* stage 1, event sets, juncs_db gather, exchange */
#define THJ_WARM_SPAN     2   /* stage 2, junction consensus */
#define THJ_WARM_INGEST   4   /* BGZF inflate, BAM parse */
#define THJ_WARM_BAMOUT   8   /* BAM record encoding, DEFLATE */
int  thj_ctx_warm(thj_ctx* ctx, int parts);
/* The side streams in use and what the measurement found for each: independent[k] = 1 when side stream k was measured to run beside
 * `stream` and the side streams before it; ratio[k] = (a 150 us spin kernel on each of them at once) / (one alone): 1.0 beside each
 * other, 2.0 one hardware queue shared, 0 not made yet.  The measurement happens inside the first thj_segjuncs_run*_async /
 * thj_span_run*_async / thj_span_tier0_pair_async call that needs a side stream (and again when a pair call needs the second):
 * that call BLOCKS the host for the ~2 ms it takes (it synchronises `stream` and runs spin kernels on it), so `stream` must not be
 * under graph capture then -- call thj_ctx_probe_streams first in such a setting.  A stream that shares a queue is still used
 * (same results, the sides partly one after the other); a warning goes to stderr and independent[k] stays 0. */
int  thj_ctx_stream_info(thj_ctx* ctx, int32_t* n_side, int32_t* independent /*[3]*/, double* ratio /*[3]*/);
/* Makes and measures `need` (1 or 2) side streams now, blocking, instead of inside the first run call. */
int  thj_ctx_probe_streams(thj_ctx* ctx, int32_t need);
/* ABI revision of this header: bumped whenever an entry point's argument layout changes (2: thj_span_tier_counts writes 5 int64,
 * thj_profile_span 8 doubles).  A caller built against another revision must not call in. */
#define THJ_ABI_VERSION 2
int  thj_abi_version(void);
void thj_ctx_destroy(thj_ctx* ctx);
int  thj_ctx_sync(thj_ctx* ctx);            /* hipStreamSynchronize on the context stream */
void* thj_ctx_stream(thj_ctx* ctx);         /* the hipStream_t in use */

/* ------------------------------------------------------------- genome */
/* Replaces RefSequenceTable + get_seqs (bwt_map.h:579-788,
 * segment_juncs.cpp:64-88): the whole reference as one HBM-resident array of
 * 32-byte blocks of 64 bases {lo bit-plane, hi bit-plane, N-mask, 0}; A=00 C=01
 * G=10 T=11, N has lo=hi=0 and its N-mask bit set, so dropping the mask gives the
 * Dna5->Dna "N becomes A" view the window copies use (segment_juncs.cpp:2157)
 * and keeping it gives the Dna5 view (:2417, long_spanning_reads.cpp:1146).
 * Contig i (ref_id i+1) starts at block contig_blk[i]; one zero guard block
 * follows every contig.  lens[i] == 0 marks an @SQ entry with no FASTA record
 * (rt.get_seq() == NULL, segment_juncs.cpp:2105-2108). */

/* Number of 32-byte blocks the packed genome needs. */
int thj_genome_layout(int32_t n_contigs, const int64_t* lens, uint32_t* contig_blk /*[n_contigs+1]*/,
                      int64_t* n_blocks);
/* Packs ASCII contigs (acgtACGT, anything else = N; seqs[i] may be NULL when
 * lens[i]==0) into `blocks` (4*n_blocks uint64, host memory). */
int thj_genome_pack(int32_t n_contigs, const char* const* seqs, const int64_t* lens,
                    const uint32_t* contig_blk, uint64_t* blocks, int64_t n_blocks);
/* Copies a packed genome to the device (H2D) and keeps it resident. */
int thj_genome_upload(thj_ctx* ctx, const uint64_t* blocks, int64_t n_blocks,
                      const uint32_t* contig_blk, const int64_t* lens, int32_t n_contigs);
/* Adopts an already device-resident packed genome (DEVICE pointer `d_blocks`,
 * not owned); contig_blk/lens are host arrays. */
int thj_genome_adopt(thj_ctx* ctx, const void* d_blocks, int64_t n_blocks,
                     const uint32_t* contig_blk, const int64_t* lens, int32_t n_contigs);

/* -------------------------------------------------------------- reads */
/* Replaces ReadStream::getRead's product (reads.cpp:528-630): read r becomes
 * 3*W uint64 {lo[W], hi[W], N[W]} bit-planes (base i = bit i%64 of word i/64)
 * plus its length.  Characters other than ACGT are N. */
int thj_reads_pack(int64_t n_reads, const int64_t* read_off, const char* bases,
                   int32_t words_per_plane, uint64_t* planes /*[n*3*W]*/, uint16_t* lens /*[n]*/);

/* ------------------------------------------------ segment_juncs batch */
/* The hits_for_read vectors that look_for_hit_group / process_next_hit_group
 * (segment_juncs.cpp:3823-4123) hand to the finders, for reads in visiting
 * (increasing id) order.  All array pointers are DEVICE pointers. */
typedef struct {
    int32_t n_reads;
    int32_t nseg;                /* number of segment maps (hits_for_read.size()): 1..16 */
    int32_t words_per_plane;     /* W of thj_reads_pack: 1..8 (reads of up to 512 bases) */
    int32_t reserved;
    const uint32_t* seg_off;     /* [n_reads*nseg+1] CSR into hits, index r*nseg+s */
    const thj_hit*  hits;
    const uint64_t* read_planes; /* [n_reads*3*W] */
    const uint16_t* read_len;    /* [n_reads] */
    const uint32_t* mate_off;    /* [n_reads+1] CSR into mate_hits, or NULL: no mates */
    const thj_hit*  mate_hits;   /* partner_hit_group of find_gaps (segment_juncs.cpp:3321-3348) */
    uint32_t ordinal_base;       /* visiting ordinal of read 0 (first-inserted-wins priority of
                                    std::set<Insertion>, insertions.h:52-67) */
    uint32_t reserved2;
} thj_seg_batch;

/* Convenience: allocates device memory for a batch given HOST arrays, copies
 * (H2D, on the context stream) and returns a device-resident descriptor that
 * thj_batch_free releases.  n_hits / n_mate_hits are the CSR totals. */
int thj_batch_upload(thj_ctx* ctx, const thj_seg_batch* host, int64_t n_hits, int64_t n_mate_hits,
                     thj_seg_batch** out);
int thj_batch_free(thj_ctx* ctx, thj_seg_batch* dev);

/* Capacity (entries, rounded up to a power of two) of the device event tables;
 * default 1<<24 junctions, 1<<20 each for deletions and insertions.  Resets. */
int thj_segjuncs_configure(thj_ctx* ctx, int64_t junc_capacity, int64_t indel_capacity);
/* Empties the event tables (start of a segment_juncs run). */
int thj_segjuncs_reset_async(thj_ctx* ctx);
/* find_insertions_and_deletions + find_gaps + juncs_from_ref_segs<RecordSegmentJuncs>
 * x {GT-AG, GC-AG, AT-AC} (segment_juncs.cpp:2807-2942, :3293-3650, :2052-2377)
 * for every read of the batch; events accumulate in the context's tables the
 * way the reference accumulates into its std::sets (:4911-4916). */
int thj_segjuncs_run_async(thj_ctx* ctx, const thj_params* p, const thj_seg_batch* dev_batch);
/* Two batches (the two sides of a pass: find_gaps over left_segmap_fnames, then over right_segmap_fnames,
 * segment_juncs.cpp:4911-4916) as one call.  The events are those of two thj_segjuncs_run_async calls in this
 * order; the second batch's kernels do not wait for the first batch's side chains. */
int thj_segjuncs_run_pair_async(thj_ctx* ctx, const thj_params* p0, const thj_seg_batch* dev_batch0,
                                const thj_params* p1, const thj_seg_batch* dev_batch1);

typedef struct { uint32_t ref_id, left, right, antisense; } thj_junction;   /* junctions.h:27-57 */
typedef struct { uint32_t ref_id, left; char seq[8]; uint64_t prio; } thj_insertion; /* insertions.h:31-67 */

typedef struct {
    int64_t n_juncs, n_deletions, n_insertions;
    int64_t n_windows;        /* RefSeg windows scanned */
    int64_t n_indel_pairs;    /* hit pairs sent to detect_small_insertion/deletion */
    int64_t n_rescue_pairs;   /* (hit, mate hit) pairs scanned by map_read_to_contig */
    int64_t n_overflow_blocks;/* workgroups that fell back to un-queued processing */
    int64_t n_hits_read;      /* thj_hit records the kernels were handed */
} thj_segjuncs_counts;

/* Compacts + sorts the tables into the order of the reference's sets
 * (junctions.h:39-57; insertions.h:52-67) and synchronises the stream.
 * Returns THJ_EOVERFLOW if a table overflowed during the runs. */
int thj_segjuncs_finish(thj_ctx* ctx, thj_segjuncs_counts* counts);
/* Copies the sorted events to host arrays sized from thj_segjuncs_finish's counts. */
int thj_segjuncs_download(thj_ctx* ctx, thj_junction* juncs, thj_junction* deletions, thj_insertion* insertions);
/* DEVICE pointers to the sorted packed 64-bit event keys after finish (for
 * on-device merging / RCCL all-gather): kind 0 junctions, 1 deletions. */
int thj_segjuncs_device_keys(thj_ctx* ctx, int kind, const uint64_t** d_keys, int64_t* n);
/* Inserts packed keys (DEVICE pointer) produced by another context/rank into this
 * context's table: the merge step of segment_juncs.cpp:4911-4916 across GPUs.
 * Precondition: the keys were made on a context holding the SAME genome layout (their position field lies below this
 * genome's n_blocks * 64 + 2): thj_segjuncs_finish sorts the event lists on as many key bits as this genome's positions
 * need, and a key beyond them would be mis-sorted silently. */
int thj_segjuncs_merge_keys_async(thj_ctx* ctx, int kind, const uint64_t* d_keys, int64_t n);

/* The sorted insertion table after finish: packed keys and values (value = first-wins
 * priority << 20 | bases), DEVICE pointers; and its merge counterpart (atomicMin on the
 * value = "earlier insertion wins", insertions.h:52-67 + std::set::insert). */
int thj_segjuncs_device_insertions(thj_ctx* ctx, const uint64_t** d_keys, const uint64_t** d_vals, int64_t* n);
int thj_segjuncs_merge_insertions_async(thj_ctx* ctx, const uint64_t* d_keys, const uint64_t* d_vals, int64_t n);

/* ------------------------------------------------ fusion search (--fusion-search) */
typedef struct { uint32_t ref_id1, ref_id2, left, right, dir, count, edit_dist, reserved; } thj_fusion;   /* fusions.h:24-116 */
#define THJ_FUSION_FF 7u
#define THJ_FUSION_FR 8u
#define THJ_FUSION_RF 9u
#define THJ_FUSION_RR 10u
/* find_fusions + detect_fusion (segment_juncs.cpp:2976-3291, :2629-2805) for every read of the batch -- which for
 * this call must hold ALL visited reads, including those whose only mapped segment is the first
 * (:3994-4028).  Candidate events accumulate on the device. */
int thj_fusion_reset_async(thj_ctx* ctx);
/* --fusion-ignore-chromosomes (segment_juncs.cpp:3214-3231): hit pairs touching one of these contigs (1-based ref ids)
 * are not examined.  Stays in force until called again (n = 0 clears). */
int thj_fusion_set_ignored(thj_ctx* ctx, const uint32_t* ref_ids, int32_t n);
int thj_fusion_run_async(thj_ctx* ctx, const thj_params* p, const thj_seg_batch* dev_batch);
/* Synchronises and reduces the events to the FusionSimpleSet (count, smallest edit distance) in
 * Fusion::operator< order (fusions.h:38-69).  THJ_ERETRY: a batch had more raw candidates than the buffer held (it grows between batches,
 * ahead of the count); the buffer now has the room the count asked for: thj_fusion_reset_async, the run calls and this again. */
int thj_fusion_finish(thj_ctx* ctx, int64_t* n_fusions);
int thj_fusion_download(thj_ctx* ctx, thj_fusion* out);

/* Average durations (ms) of the kernels of a run, measured with HIP events on the stream each runs on (the reads
 * with several hits a segment have streams of their own beside the flat reads' kernels), over the runs since the
 * last call (a pair call counts as two).  avg_ms[8]: `thj_k_sj_flat`; `thj_k_sj_general`, first instance; second
 * instance; `thj_k_segjuncs_shared`; `thj_k_segjuncs_rescue` + `_rescue_shared`; `thj_k_sj_tasks_list`;
 * `thj_k_sj_rescue_scan` + `thj_k_sj_rescue_flat`; `thj_k_sj_tasks`.  stats[4] (may be null), averages over the
 * launches whose lists are still there: reads given to `thj_k_segjuncs_shared`, to the second instance of
 * `thj_k_sj_general`, tasks `thj_k_sj_tasks_list` ran, flat rescue pairs.  Enables event recording when `enable` != 0. */
int thj_profile_segjuncs(thj_ctx* ctx, int enable, double* avg_ms, int64_t* launches, double* stats);
/* Measurement aid: with `on` != 0 the stage calls of this context launch every kernel on the context's stream, one after the
 * other (no side streams) -- the durations thj_profile_segjuncs / thj_profile_span then report are each kernel's own, not those of
 * kernels sharing the GPU.  Results are the same either way.  Default 0. */
int thj_profile_serial(thj_ctx* ctx, int on);


/* --------------------------------------------------- juncs_db
 * The step between the two executables (juncs_db.cpp:73-233): FASTA records of the sequence around every junction /
 * deletion / insertion / fusion, cut from the genome.  The resident bit-plane genome makes this a gather: the caller
 * lists the pieces it wants, the device writes their bases as ASCII (A C G T N; reverse-complemented when asked, N
 * stays N) at the byte offsets given, and the caller interleaves its header lines. */
typedef struct thj_piece {
    uint32_t ref_id;      /* 1-based */
    int32_t  start;       /* 0-based first base */
    int32_t  len;         /* bases; the piece must lie inside the contig */
    uint32_t flags;       /* THJ_PIECE_RC */
} thj_piece;
#define THJ_PIECE_RC 1u
/* HOST arrays in, HOST bytes out: out[out_off[i] .. out_off[i] + pieces[i].len) receives piece i.  Synchronous. */
int thj_genome_gather(thj_ctx* ctx, const thj_piece* pieces, int64_t n, const int64_t* out_off, char* out, int64_t out_bytes);

/* --------------------------------------------------- long_spanning_reads */

/* CigarOpCode values of bwt_map.h:36-55, packed as (op << 28) | length. */
#define THJ_CIG_MATCH     1u
#define THJ_CIG_INS       3u
#define THJ_CIG_DEL       5u
#define THJ_CIG_REF_SKIP 11u
#define THJ_CIG_SOFT_CLIP 13u
/* --fusion-search: the lower-case (running down the genome) forms and the four fusion ops, whose length is the 0-based
 * position on the second contig (bwt_map.h:36-55; print_bamhit writes them as m / i / d / n and <pos + 1>F) */
#define THJ_CIG_mATCH     2u
#define THJ_CIG_iNS       4u
#define THJ_CIG_dEL       6u
#define THJ_CIG_FUSION_FF 7u
#define THJ_CIG_FUSION_FR 8u
#define THJ_CIG_FUSION_RF 9u
#define THJ_CIG_FUSION_RR 10u
#define THJ_CIG_rEF_SKIP 12u

/* A segment alignment with its CIGAR: BowtieHit as produced by BAMHitFactory /
 * SplicedBAMHitFactory (bwt_map.cpp:1101-1452, :1469-1770). 32 bytes. */
typedef struct {
    uint32_t ref_id;
    int32_t  left;
    uint8_t  flags;        /* THJ_HIT_ANTISENSE | THJ_HIT_END | THJ_HIT_ANTISENSE_SPLICE */
    uint8_t  mismatches;
    uint8_t  edit_dist;
    uint8_t  n_cigar;      /* 1..5 */
    uint32_t cigar[5];
} thj_span_hit;
#define THJ_HIT_ANTISENSE_SPLICE 4u
/* A segment hit on a fusion contig of the junction database (SplicedBAMHitFactory, bwt_map.cpp:1655-1760) has a fusion op in
 * its cigar, at most 4 ops, and its second contig (ref_id2) in cigar[4].  On rf / rr contigs antisense_align is the opposite
 * of the record's strand flag (:1744-1745): THJ_HIT_STRAND_FLIPPED says so, i.e. the record's SEQ is the read piece
 * reverse-complemented iff THJ_HIT_ANTISENSE xor THJ_HIT_STRAND_FLIPPED. */
#define THJ_HIT_STRAND_FLIPPED 8u
#define THJ_HIT_FUSED 16u          /* the hit's cigar holds a fusion op (set by whoever builds the record; the stitch kernels route on it) */

/* Per-read segment hit lists of JoinSegmentsWorker (long_spanning_reads.cpp:2669-2845):
 * for every read with a hit in the first segment map, segment s holds the contig
 * hits then the spliced hits (:2706-2765, :87-163).  DEVICE pointers. */
typedef struct {
    int32_t n_reads;
    int32_t nseg;
    int32_t words_per_plane;
    int32_t qual_stride;         /* bytes between consecutive reads' quality strings */
    const uint32_t*     seg_off;     /* [n_reads*nseg+1] */
    const thj_span_hit* hits;
    const uint64_t*     read_planes; /* [n_reads*3*W] (thj_reads_pack) */
    const uint16_t*     read_len;
    const uint8_t*      quals;       /* phred+33, qual_stride bytes per read */
    const void*         hit_heads;   /* the first 16 bytes of every hit record, densely (contig, position, flags, first cigar op: all
                                      * the first stitch tier reads of a hit, so it streams 16 B per hit).  Every batch this library
                                      * makes has one -- thj_span_batch_upload derives it once at upload, the device-side ingest
                                      * writes it beside the records; NULL is accepted from other producers (the tier then reads
                                      * the 32-byte records) */
} thj_span_batch;

/* One output record: the fields print_bamhit + bowtie_sam_extra write
 * (bwt_map.cpp:1888-2093, :2467-2648). 128 bytes. */
typedef struct {
    uint32_t read_idx;     /* index of the read in its batch */
    uint32_t ref_id;
    int32_t  left;         /* POS - 1 */
    uint8_t  flags;        /* THJ_HIT_ANTISENSE (-> FLAG 0x10) | THJ_HIT_ANTISENSE_SPLICE (-> XS:A:-) */
    uint8_t  mismatches;   /* NM = mismatches + indel lengths */
    uint8_t  edit_dist;
    uint8_t  n_cigar;
    int16_t  AS;
    uint8_t  XM, XO, XG, md_len;
    uint16_t order;        /* rank among the read's records (BowtieHit::operator<, bwt_map.h:180-207) */
    uint32_t cigar[16];    /* a fusion alignment (one of its ops is THJ_CIG_FUSION_*) has at most 15 ops and its second contig,
                              ref_id2, in cigar[15] */
    char     md[40];       /* MD:Z value, md_len characters; md_len == THJ_MD_ON_HOST: longer than the record holds -- thj_md_string */
} thj_aln;
#define THJ_MD_ON_HOST 255
/* MD:Z of one alignment as bowtie_sam_extra writes it (bwt_map.cpp:2467-2648), on the host: ref = the contig's bases (any
 * case, non-ACGT = N), seq = the read bases in the orientation of the alignment, cigar = op << 28 | length.  Writes a
 * NUL-terminated string, returns its length (< 0: error).  For the records the device flags with THJ_MD_ON_HOST. */
int thj_md_string(const char* ref, int64_t ref_len, const char* seq, int32_t seq_len, int32_t left, const uint32_t* cigar, int32_t n_cigar,
                  char* out, int32_t out_cap);
/* The same for an alignment that may run down the genome (lower-case ops) and change contigs at a fusion op: ref2 = the
 * second contig (= ref for a plain alignment). */
int thj_md_string2(const char* ref, int64_t ref_len, const char* ref2, int64_t ref2_len, const char* seq, int32_t seq_len, int32_t left,
                   const uint32_t* cigar, int32_t n_cigar, char* out, int32_t out_cap);

/* The junction (+deletion) and insertion sets long_spanning_reads loads from its list
 * files (long_spanning_reads.cpp:2897-2980).  juncs: sorted unique in Junction::operator<
 * order, deletions already merged in as Junction(left, right, '+'); insertions: n_ins rows
 * of 4 uint32 {ref_id, left, length, bases 3 bits each (A,C,G,T,N = 0..4)} sorted by
 * (ref_id, left, length).  HOST pointers. */
int thj_span_sets_upload(thj_ctx* ctx, const thj_junction* juncs, int64_t n_juncs, const uint32_t* insertions, int64_t n_ins);
/* Same, taken device-to-device from the context's own finished segment_juncs tables. */
int thj_span_sets_from_segjuncs(thj_ctx* ctx);
/* The same for --fusion-search: the fusion list thj_fusion_finish left in this context (segment_juncs' .fusions output, what
 * long_spanning_reads reads back with --fusion-search) becomes the spanning stage's list without leaving the device.  Replaces
 * thj_fusion_download + thj_span_fusions_upload when both stages run in one process. */
int thj_span_fusions_from_segjuncs(thj_ctx* ctx);

/* --fusion-search: the .fusions list long_spanning_reads loads (long_spanning_reads.cpp:2998-3040), sorted unique in
 * Fusion::operator< order (fusions.h:44-71); dir = THJ_CIG_FUSION_*.  HOST pointer.  thj_span_run_async consults it when
 * thj_params.fusion_search is set; the reads that are not plain runs of abutting single hits then go through the fusion
 * branches of dfs_seg_hits / merge_chain (:2222-2610, :805-2038) in thj_k_stitch_fusion. */
typedef struct { uint32_t ref_id1, ref_id2, left, right, dir; } thj_span_fusion;
int thj_span_fusions_upload(thj_ctx* ctx, const thj_span_fusion* fusions, int64_t n);

int thj_span_batch_upload(thj_ctx* ctx, const thj_span_batch* host, int64_t n_hits, thj_span_batch** out);
int thj_span_batch_free(thj_ctx* ctx, thj_span_batch* dev);
/* d_heads[i] = first 16 bytes of d_hits[i] (device buffers, n_hits * 16 bytes out): the optional hit_heads array of a batch */
int thj_span_hit_heads_async(thj_ctx* ctx, const thj_span_hit* d_hits, int64_t n_hits, void* d_heads);

int thj_span_reset_async(thj_ctx* ctx);
/* join_segments_for_read + sort/unique + filters + bowtie_sam_extra for every read of the
 * batch (long_spanning_reads.cpp:2612-2667, :2767-2831); records accumulate in HBM. */
int thj_span_run_async(thj_ctx* ctx, const thj_params* p, const thj_span_batch* dev_batch);
/* Two batches -- the left and the right reads of a pass, or two shards of one side -- as one call: the same records in the same
 * slots as thj_span_run_async(batch0) followed by thj_span_run_async(batch1), with the kernels of the two batches running beside
 * each other on streams of the context's own (the latency-bound kernels of one batch overlap the bandwidth-bound ones of the
 * other).  Joined on the context's stream before it returns. */
int thj_span_run_pair_async(thj_ctx* ctx, const thj_params* p, const thj_span_batch* dev_batch0, const thj_span_batch* dev_batch1);
/* Optional, before thj_span_run_pair_async on the same two batches (after thj_span_reset_async): the part of the pair's work
 * that needs the batches and nothing else -- the reads whose segment hits abut end to end, two thirds of all -- is enqueued
 * now; the run then only does what is behind it.  Stage 2's junction / indel sets need not exist yet: a caller that holds both
 * stages resident calls this before thj_segjuncs_finish, and the GPU has that work while the host waits for stage 1's counts
 * and the event lists are sorted.  Same records as without the call.  Does nothing (the run does everything) for an empty
 * batch, under --fusion-search or THJ_SPAN_SERIAL.  THJ_ESTATE when anything but that run follows. */
int thj_span_tier0_pair_async(thj_ctx* ctx, const thj_params* p, const thj_span_batch* dev_batch0, const thj_span_batch* dev_batch1);
/* Synchronises and returns the record count.  THJ_EOVERFLOW when a device limit was hit (message says which); THJ_ERETRY when
 * a pool or workspace had to be enlarged (extra records of multihit reads; reads with more joined alignments than a thread keeps):
 * run the pass again.
 * On the device the records sit in BAM order already: one slot per read of the pass (run order, then read order)
 * holding the read's first record, a per-read count, and the few extra records of multihit reads keyed by
 * (slot << 16 | rank) -- rank = BowtieHit::operator< order inside the read (bwt_map.h:180-207). */
int thj_span_finish(thj_ctx* ctx, int64_t* n_alns);
/* Compact ordered host copy of the n_alns records. */
int thj_span_download(thj_ctx* ctx, thj_aln* out);
/* A record as the stitch kernels leave it in HBM: the same 128 bytes as thj_aln in two 64-byte lines, ordered so that an
 * ordinary alignment needs only the first.  The lead line: the 24-byte header of thj_aln, cigar ops 0..3, MD characters 0..23.
 * The tail line: cigar ops 4..15, MD characters 24..39 -- written, and THJ_SLOT_TAIL set in `flags`, only for a record with more
 * than four cigar ops, an MD string of more than 24 characters, or a fusion alignment; otherwise its bytes are undefined
 * (left over from an earlier pass) and the fields it would hold are zero.  thj_span_download converts to thj_aln. */
typedef struct {
    uint32_t read_idx, ref_id; int32_t left;
    uint8_t  flags, mismatches, edit_dist, n_cigar;
    int16_t  AS; uint8_t XM, XO, XG, md_len; uint16_t order;
    uint32_t cigar_lo[4];  char md_lo[24];     /* lead line ends here (64 bytes) */
    uint32_t cigar_hi[12]; char md_hi[16];     /* tail line: valid only with THJ_SLOT_TAIL */
} thj_aln_slot;
#define THJ_SLOT_TAIL 0x80u
/* The device-resident layout described above (DEVICE pointers), for consumers that stay on the GPU. */
int thj_span_device_records(thj_ctx* ctx, const thj_aln_slot** d_slots, const uint8_t** d_counts, int64_t* n_reads,
                            const thj_aln_slot** d_extra, const uint64_t** d_extra_keys, int64_t* n_extra);
/* Of the batch launched last: counts[0] = reads sent to the closure kernels (as chain entries to thj_k_join / thj_k_finish, or to
 * thj_k_stitch), counts[1] = to the multihit kernel thj_k_stitch_pack, counts[2] = on to the general kernel thj_k_stitch_generic,
 * counts[3] = those of counts[0] that travelled as chain entries, counts[4] = those of counts[1] whose chains thj_k_chains turned into
 * chain entries (joined and finished by thj_k_join / thj_k_finish instead of the packed tier); the rest were finished by
 * thj_k_stitch_contig. */
int thj_span_tier_counts(thj_ctx* ctx, int64_t* counts /*[5]*/);
/* Average durations (ms) of the stitch kernels since the last call -- avg_ms[0] thj_k_stitch_contig, [1] thj_k_chains, [2] thj_k_join,
 * [3] thj_k_join_closure, [4] thj_k_finish, [5] thj_k_stitch, [6] thj_k_stitch_pack, [7] thj_k_stitch_generic or, with --fusion-search,
 * thj_k_stitch_fusion -- from HIP events on the stream each runs on (kernels of batches that run beside each other share the GPU:
 * their durations overlap). */
int thj_profile_span(thj_ctx* ctx, int enable, double* avg_ms /*[8]*/, int64_t* launches);

/* ---- coverage search of segment_juncs (segment_juncs.cpp:4268-4543 capture_island_ends and what it calls: the
 * coverage map of build_coverage_map :4140-4176, the extension table of index_read_mers :548-571,
 * juncs_from_ref_segs<RecordExtendableJuncs> :2052-2377 with RecordExtendableJuncs::record :1568-1626).
 * Call order inside one segment_juncs pass: thj_covsearch_reset_async; thj_covsearch_add_hits_async for every uploaded
 * batch of both sides (the segment maps `all_segmap_fnames`, :4929-4935); thj_covsearch_add_reads for the initially
 * unmapped reads (--ium-reads; planes in the thj_reads_pack layout and lengths, host buffers or -- on_device != 0 -- device
 * buffers; only the first 32 bases of a read are used, :425); thj_covsearch_run_async finds the junctions; thj_covsearch_finish adds them to
 * the pass's junction set -- all of them, or, when there are more than max_cov_juncs (:56), the max_cov_juncs smallest by
 * (skip count, junction) as the reference's capped set keeps them (:1611-1621) -- and returns how many; it goes before
 * thj_segjuncs_finish. */
int thj_covsearch_reset_async(thj_ctx* ctx);
int thj_covsearch_add_hits_async(thj_ctx* ctx, const thj_seg_batch* device_batch);
int thj_covsearch_add_reads(thj_ctx* ctx, int64_t n_reads, int32_t words_per_plane, const uint64_t* planes, const uint16_t* lens,
                            int32_t on_device);
/* The same for a piece (whole BGZF members; thj_bam_piece below) of an UNALIGNED BAM of reads -- what tophat.py passes as --ium-reads --:
 * inflated and parsed on the device (thj_bgzf_inflate's kernels, the record walk of the ingest), every record a read
 * (ReadStream::get_direct).  *n_reads = records taken.  THJ_EFALLBACK as for thj_ingest_*: read the piece on the host.  Synchronous. */
/* Room for n_reads more unmapped reads in the extension table now (a caller that knows roughly how many are coming: the table otherwise
 * grows by half whenever it is full, each time a copy of what it holds). */
int thj_covsearch_reserve_reads(thj_ctx* ctx, int64_t n_reads);
struct thj_bam_piece;
int thj_covsearch_add_reads_bam(thj_ctx* ctx, const struct thj_bam_piece* reads, int64_t* n_reads);
int thj_covsearch_run_async(thj_ctx* ctx, int32_t min_cov_length, int32_t min_coverage_intron, int32_t max_coverage_intron);
int thj_covsearch_finish(thj_ctx* ctx, int64_t max_cov_juncs, int64_t* n_found);
/* Butterfly search (--butterfly-search; replaces prune_extension_table + compact_extension_table + pair_covered_sites with
 * juncs_from_ref_segs<RecordButterflyJuncs>, segment_juncs.cpp:466-501, :4178-4249, :1698-2049): works from the same state as the
 * coverage search -- the coverage map of thj_covsearch_add_hits_async and the extension table of thj_covsearch_add_reads (after
 * thj_covsearch_allgather when ranks share the reads); neither is changed.  Every island of the coverage map, widened by 45 bases,
 * is searched for GT / CT and AG / AC; a donor and an acceptor min..max_coverage_intron apart make a junction when an unmapped read's
 * seed next to the one and a seed next to the other show the same twelve bases across the gap.  The junctions -- all, or the
 * max_cov_juncs with the shortest introns (the capped set of :2031-2041) -- are added to the pass's junction set; *n_found = their
 * number.  Synchronous.  Run it after thj_covsearch_finish (when the coverage search runs at all: they share buffers) and before
 * thj_segjuncs_finish. */
int thj_butterfly_run(thj_ctx* ctx, int32_t min_coverage_intron, int32_t max_coverage_intron, int64_t max_cov_juncs, int64_t* n_found);
/* Microexon search (segment_juncs.cpp:3737-3941; replaces align_microexon_segs and the window registration inside look_for_hit_group).
 * A read whose first segment has no hit while every other segment has some may begin in a microexon: every hit of its second segment
 * registers a window of 2000 bases beside it with the read's first segment_length bases.  thj_microexon_collect finds those
 * candidates on the device for an uploaded / ingested batch (read_side 1 = left, 2 = right; ordinals = the batch's ordinal_base + row);
 * thj_microexon_candidates hands all of them to the host (malloc'd: free()), which merges overlapping windows in the reference's
 * visiting order -- sequential std::map logic, csrc/host/thj_mx_host.h; thj_microexon_run then builds every window's own extension
 * table, pairs the GT / CT sites of a window with its AG / AC sites (one wave per window), applies the max_cov_juncs cut
 * (segment_juncs.cpp:5021-5024) and adds the junctions to the pass's set.  Run it after thj_covsearch_finish (they share buffers).
 * str = the 2-bit string, first base most significant; strings of a window in the order the reference pooled them (the table is a set:
 * the order does not matter).  segment_length 10..32. */
typedef struct { uint32_t ordinal; uint16_t rank; uint8_t side, len; uint32_t ref_id; int32_t left, right; uint32_t reserved; uint64_t str; } thj_mx_cand;
typedef struct { uint32_t ref_id; int32_t left, right; int32_t side; } thj_mx_window;
int thj_microexon_reset_async(thj_ctx* ctx);
int thj_microexon_collect(thj_ctx* ctx, const thj_params* p, const thj_seg_batch* device_batch, int32_t read_side);
int thj_microexon_candidates(thj_ctx* ctx, thj_mx_cand** out, int64_t* n);
int thj_microexon_run(thj_ctx* ctx, const thj_mx_window* windows, int64_t n_windows, const uint64_t* strs, const uint8_t* str_len, const uint32_t* str_window,
                      int64_t n_strs, int32_t min_coverage_intron, int32_t library_type, int64_t max_cov_juncs, int64_t* n_found);

/* Reads sharded over GPUs: the coverage map of the whole run is the OR of the ranks' maps and the
 * extension table that of all their unmapped reads.  thj_covsearch_device_state exposes a rank's state (device
 * pointers: n_words coverage words, one size per contig, n_ext read records -- d_ext_keys[i] = min(length, 32) of read i,
 * d_ext_vals[i] = its first 32 bases as a 2-bit string, first base most significant: the table's entries are made from the
 * records when a pass builds the table) for the caller to all-gather;
 * thj_covsearch_merge_async folds another rank's state in; then every rank runs thj_covsearch_run_async. */
int thj_covsearch_device_state(thj_ctx* ctx, const uint64_t** d_cov_bits, int64_t* n_words, const int32_t** d_cov_size,
                               const uint32_t** d_ext_keys, const uint64_t** d_ext_vals, int64_t* n_ext);
int thj_covsearch_merge_async(thj_ctx* ctx, const uint64_t* d_other_bits, const int32_t* d_other_size,
                              const uint32_t* d_other_keys, const uint64_t* d_other_vals, int64_t n_other_ext);

/* ---------------------------------------------------------------- device-side ingest
 * BGZF members (samtools-0.1.18 bgzf.c) are independent DEFLATE streams of at most 64 KiB; thj_bgzf_inflate inflates many of
 * them at once on the GPU (one workgroup per member, tables and window in LDS).  in_off / in_len address a member's DEFLATE
 * payload (after its 18-byte header, before the 8-byte CRC32 / ISIZE trailer) inside `comp`.  Member b's bytes land at
 * out + b * 65536, out_len[b] = their number (0xFFFFFFFF: corrupt stream).  on_device == 0: host arrays, synchronous;
 * != 0: device arrays, the call only enqueues on the context stream. */
typedef struct { uint64_t in_off; uint32_t in_len; uint32_t reserved; } thj_bgzf_block;
int thj_bgzf_inflate(thj_ctx* ctx, const uint8_t* comp, int64_t comp_bytes, const thj_bgzf_block* blocks, int64_t n_blocks,
                     uint8_t* out, uint32_t* out_len, int32_t on_device);

/* One input file's share of a read-id shard, still compressed: `comp` = comp_bytes bytes of the BAM file starting at a BGZF member
 * (HOST memory, e.g. a mapping of the file) and ending at a member boundary; first_skip = bytes of the first member that come
 * before the shard's first record (the low 16 bits of the .index offset, or the end of the BAM header); tid2ref[t] = ref_id of
 * the file's target t (0: a contig the run does not know). */
typedef struct thj_bam_piece { const uint8_t* comp; int64_t comp_bytes; uint32_t first_skip; int32_t n_tid; const uint32_t* tid2ref; } thj_bam_piece;
/* The whole ingest of one shard of one side of segment_juncs on the device: inflates the pieces, parses their records
 * (BAMHitFactory::get_hit_from_buf, bwt_map.cpp:1101-1452; reads: SEQ -> bit planes), keeps read ids in [begin_id, end_id),
 * merges the nseg segment maps by read id into the visiting set of look_for_hit_group (segment_juncs.cpp:3823-4123; reads whose
 * highest mapped segment is the first are left out unless include_top0), joins the mate's hits (whole-read map, else last
 * segment map: find_gaps :3321-3348) and the reads, and returns a device-resident batch for thj_segjuncs_run_async /
 * thj_fusion_run_async / thj_covsearch_add_hits_async (thj_batch_free releases it).  *out == NULL with THJ_OK: the shard is
 * empty.  THJ_EFALLBACK: see above.  mate_full / mate_last may be NULL.  Synchronous. */
/* THJ_INGEST_TIMING=1 in the environment: the ingest calls time their phases (with a stream synchronisation at every mark: a
 * diagnosis mode), this prints the sums to stderr. */
void thj_ingest_timing_report(void);
int thj_ingest_seg_batch(thj_ctx* ctx, const thj_params* p, int32_t nseg, const thj_bam_piece* segs, const thj_bam_piece* mate_full,
                         const thj_bam_piece* mate_last, const thj_bam_piece* reads, uint32_t begin_id, uint32_t end_id, int32_t include_top0,
                         uint32_t ordinal_base, thj_seg_batch** out, int64_t* n_reads);

/* long_spanning_reads: the contig segment maps of one shard on the device -> a batch whose per-(read, segment) hit lists are
 * set, for the reads with a hit in the FIRST segment map (the groups JoinSegmentsWorker iterates over,
 * long_spanning_reads.cpp:2706-2765).  *row_ids (HOST, malloc'd: free() it) = their read ids in row order; the caller fetches
 * those reads -- it needs names, bases and qualities for the BAM records anyway -- and completes the batch with
 * thj_span_batch_attach_reads (HOST arrays in the layouts of thj_reads_pack / thj_span_batch) before thj_span_run_async.
 * thj_span_batch_free releases the batch.  *out == NULL with THJ_OK: nothing to do in this shard. */
int thj_ingest_span_hits(thj_ctx* ctx, const thj_params* p, int32_t nseg, const thj_bam_piece* segs, uint32_t begin_id, uint32_t end_id,
                         thj_span_batch** out, uint32_t** row_ids, int64_t* n_rows);
/* The same with the shard's piece of the READS file (unaligned BAM, id-sorted) riding along: its members are inflated and its
 * records located with the maps', the batch leaves complete (read planes, lengths and quality strings written on the device:
 * no thj_span_batch_attach_reads), and the inflated read records come back for the BAM output -- *reads_infl (HOST, page-locked,
 * from thj_pinned_alloc: thj_pinned_free() it): reads_infl_bytes bytes, BGZF member m of the piece at m << 16; row_loc[r] (HOST, malloc'd) = (m << 16 | offset)
 * of the block_size field of row r's record.  Replaces ReadStream::getRead per row (reads.cpp:528-630). */
/* Page-locked host buffers from a pool of the process (locking pages is slow, copies from and to them are plain DMA): for what a
 * caller hands to thj_ingest_* (thj_bam_piece.comp may point into one) and for what thj_ingest_span_batch hands back. */
void* thj_pinned_alloc(size_t bytes);
void thj_pinned_free(void* p);
/* From now on buffers are unlocked and released when they are handed back (and the idle ones at once) instead of kept for the next
 * caller: a process calls this when it has started its last piece of work, so that taking its page-locked memory apart (0.11 s per
 * GB on this driver) runs beside that work and not after the process's last instruction. */
void thj_pinned_drain(void);
int thj_ingest_span_batch(thj_ctx* ctx, const thj_params* p, int32_t nseg, const thj_bam_piece* segs, const thj_bam_piece* reads,
                          uint32_t begin_id, uint32_t end_id, thj_span_batch** out, uint32_t** row_ids, int64_t* n_rows,
                          uint8_t** reads_infl, int64_t* reads_infl_bytes, uint32_t** row_loc);
int thj_span_batch_attach_reads(thj_ctx* ctx, thj_span_batch* batch, int32_t words_per_plane, int32_t qual_stride, const uint64_t* planes,
                                const uint16_t* lens, const uint8_t* quals);
/* reads_infl, reads_infl_bytes and row_loc may all three be NULL: the read records then stay on the device with the batch only
 * (thj_span_bam_encode reads them there).  A caller that needs the host copy after all asks for it here (same buffers as above). */
int thj_span_batch_reads_host(thj_ctx* ctx, const thj_span_batch* batch, uint8_t** reads_infl, int64_t* reads_infl_bytes, uint32_t** row_loc);

/* ---------------------------------------------------------------- the BAM writer's device side (the output format)
 * long_spanning_reads prints every alignment with print_bamhit (bwt_map.cpp:1888-2093: GBamRecord with the mate fields "*", 0, 0,
 * MAPQ 255 and the tags AS XM XO XG MD NM [XS], add_aux common.cpp:1092-1173) through bam_write1 -> bgzf_write -> deflate_block
 * (samtools-0.1.18 bam.c:207-236, bgzf.c:287-349, :587-623).  Here the records are built and deflated where the alignments already
 * are; the host decides where BGZF members end (bam_write1's bgzf_flush_try rule needs record sizes only), wraps the deflated
 * members in their 18 + 8 bytes and writes them.
 *
 * thj_span_bam_encode: after thj_span_finish.  The pass's alignments (thj_span_finish's count n, thj_span_download's order) as BAM
 *   records, block_size fields included, back to back in a buffer the context keeps until the next call.  batch: the pass's batch
 *   from thj_ingest_span_batch (it holds the reads' own BAM records, where names, bases and qualities are copied from);
 *   tid_of_ref[ref_id - 1] = the contig's index in the output header; rec_size[i] / rec_id[i] (HOST, n entries) = byte count and
 *   read id (atol of the name) of record i; *total_bytes = the stream's length.  THJ_EFALLBACK: a record of the pass needs the host
 *   encoder (a fusion alignment: two records with XF:Z; an MD string the device record does not hold; a read whose length differs
 *   from the alignment's) or the batch has no read records -- nothing was encoded, thj_span_download still works.
 * thj_bgzf_deflate: the context's stream cut at member_end[k] (exclusive, rising; every member 1..65536 bytes).  comp (HOST,
 *   comp_cap bytes) receives the members' raw DEFLATE streams back to back, comp_len[k] bytes each; crc[k] = CRC-32 of member k's
 *   bytes.  One dynamic-Huffman block per member.  THJ_EFALLBACK: a member's DEFLATE stream exceeds 65536 - 26 bytes
 *   (incompressible data; bgzf.c then shrinks the member, which moves every later cut -- the caller replays on the host).
 * thj_bam_stream_upload / _download: set / fetch the context's stream (other producers; the host fallback; tests). */
int thj_span_bam_encode(thj_ctx* ctx, const thj_span_batch* batch, const int32_t* tid_of_ref, int32_t n_ref, uint32_t* rec_size, int64_t* rec_id,
                        int64_t* total_bytes);
int thj_bgzf_deflate(thj_ctx* ctx, int64_t n_members, const int64_t* member_end, uint8_t* comp, int64_t comp_cap, uint32_t* comp_len, uint32_t* crc,
                     int64_t* comp_bytes);
int thj_bam_stream_upload(thj_ctx* ctx, const uint8_t* bytes, int64_t n);
int thj_bam_stream_download(thj_ctx* ctx, uint8_t* bytes);

/* ---------------------------------------------------------------- junction consensus
 * What tophat_reports does with the reported alignments to get junctions.bed: every REF_SKIP is a junction observation
 * (junctions_from_spliced_hit, junctions.cpp:19-97), observations of one junction merge (support adds up, extents take the
 * maximum: JunctionStats::merge_with, junctions.h:87-101), filter_junctions (accept_if_valid + knockout_shadow_junctions,
 * junctions.cpp:192-330) judges the set, alignments on a rejected junction are dropped and the rest make the final set
 * (tophat_reports.cpp:1182-1230), minus junctions whose extents stay below 8 (:2974-2984).  Not included: tophat_reports'
 * choice of which alignments of a read to report -- every record handed in counts.
 * Call order: reset; add (any number of times); finish; download. */
typedef struct { uint32_t ref_id, left, right, antisense, left_extent, right_extent, support, reserved; } thj_juncstat;
/* Capacity (distinct junctions) of the device table; default: four times the candidate set of the spanning pass, >= 2^20. */
int thj_juncbed_configure(thj_ctx* ctx, int64_t junction_capacity);
int thj_juncbed_reset_async(thj_ctx* ctx);
/* The records of the last long_spanning_reads pass, still resident on the device (after thj_span_finish). */
int thj_juncbed_add_span_async(thj_ctx* ctx);
/* Any alignment records (HOST array, or DEVICE array when on_device != 0): ref_id, left, flags & THJ_HIT_ANTISENSE_SPLICE,
 * n_cigar and cigar are read. */
int thj_juncbed_add_records(thj_ctx* ctx, const thj_aln* recs, int64_t n, int32_t on_device);
/* Filters, second pass, final set in Junction::operator< order; synchronises.  min_anchor_len = --min-anchor (common.cpp:105). */
int thj_juncbed_finish(thj_ctx* ctx, int32_t min_anchor_len, int64_t* n_juncs);
int thj_juncbed_download(thj_ctx* ctx, thj_juncstat* out);

/* ---------------------------------------------------------------- multi-GPU exchange step
 * Reads shard over GPUs (contiguous read-id ranges, the reference's own thread partition: utils.cpp:22-170,
 * segment_juncs.cpp:4793-4810), the genome is replicated, and the per-rank event sets are united ONCE before
 * long_spanning_reads -- what segment_juncs.cpp:4911-4922 does with its per-thread sets -- by one RCCL all-gather
 * over xGMI.  A communicator binds one rank to one context (its device and stream).  Every rank's collective calls
 * must be made by its own host thread or process, in the same order on all ranks. */
typedef struct thj_comm thj_comm;
#define THJ_COMM_ID_BYTES 128
#define THJ_COMM_SELF      0   /* one rank, no transport */
#define THJ_COMM_RCCL      1   /* ncclAllGather on the context stream */
#define THJ_COMM_LOOPBACK  2   /* ranks of one process sharing a device: stream-ordered device copies (functional tests on a 1-GPU box) */
/* ncclGetUniqueId: made by one rank, handed to the others by the caller's own means (file, pipe, MPI, torch.distributed) */
int thj_comm_unique_id(uint8_t* id /*[THJ_COMM_ID_BYTES]*/);
/* One rank per process (or thread): ncclCommInitRank.  id == NULL with n_ranks == 1 gives a transport-less communicator. */
int thj_comm_create(thj_ctx* ctx, const uint8_t* id, int32_t n_ranks, int32_t rank, thj_comm** out);
/* All ranks inside this process, rank i on ctxs[i]: RCCL when the contexts sit on distinct GPUs, loopback otherwise. */
int thj_comm_create_local(thj_ctx* const* ctxs, int32_t n, thj_comm** out /*[n]*/);
void thj_comm_destroy(thj_comm* comm);
/* stats: [0] exchange steps enqueued, [1] of them repeats with larger message sections, [2] bytes one rank sends per step,
 * [3] junction-section capacity (keys) */
int thj_comm_info(const thj_comm* comm, int32_t* n_ranks, int32_t* rank, int32_t* transport, int64_t* stats /*[4]*/);

/* After this rank's thj_segjuncs_run_async calls (and thj_covsearch_finish), before thj_segjuncs_finish: packs the distinct
 * junction / deletion / insertion events, all-gathers them and inserts the other ranks' events into this rank's tables --
 * all enqueued on the context stream, no host synchronisation.  thj_segjuncs_finish then returns the united sets on every
 * rank (and repeats the step by itself if a message section or a table turned out too small). */
int thj_events_allgather_async(thj_ctx* ctx, thj_comm* comm);
/* After thj_fusion_finish: every rank's FusionSimpleSet becomes the merge_with() of all of them (fusions.cpp:975-990:
 * counts add up, the smallest edit distance wins); *n_fusions = size of the merged set, which thj_fusion_download then
 * copies out.  Synchronous. */
int thj_fusion_allgather(thj_ctx* ctx, thj_comm* comm, int64_t* n_fusions);
/* Before thj_covsearch_run_async: the coverage map becomes the OR of the ranks' maps, the contig extents their maximum, the
 * extension table the concatenation of their entries; every rank then runs the same coverage search.  Synchronous. */
int thj_covsearch_allgather(thj_ctx* ctx, thj_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* THJ_H */
