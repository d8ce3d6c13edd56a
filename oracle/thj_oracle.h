/*
 * thj_oracle.h -- CPU oracle for the TopHat splice-junction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (tophat_amd/, the
 * C-ABI library, the drop-in binaries) may include, link or call this code.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker.
 *
 * It is an independent plain-C restatement of the reference algorithm
 * (DaehwanKimLab/tophat v2.1.2, src/segment_juncs.cpp and
 * src/long_spanning_reads.cpp); every function cites the reference lines it
 * follows.  It deliberately uses a different data representation from the
 * device path (ASCII genome and reads, byte-at-a-time loops) so that agreement
 * between the two is evidence, not tautology.
 *
 * PARITY PINNING STATUS: see oracle/README.md.  Pinned in part: the reference's
 * own regression cases (tests/regression_tests/test_cases, fixtures under
 * tests/golden_ref/) -- the junction of test_SimpleSplicing is found exactly,
 * and every recorded alignment of the three cases (spliced, insertion and
 * deletion records included) is reproduced with its strand, POS, CIGAR and NM.
 * Everything those cases do not reach (segment_juncs' indel search, the
 * mate-anchored rescue, paired-end input, fusions, the AS/XM/XO/XG/MD tags,
 * record order) has no reference fixture, and the reference cannot be built
 * in this image without stand-ins for Boost / autotools output: for those
 * parts, formally "parity unpinned".  oracle/README.md describes the informal
 * differential checks that were run against a survey-stage scratch build.
 */
#ifndef THJ_ORACLE_H
#define THJ_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* One segment (or mate) alignment, the fields of BowtieHit (bwt_map.h:36-536)
 * that the hot path reads.  `right` is BowtieHit::right() (bwt_map.h:213-243)
 * and `read_len` BowtieHit::read_len() (bwt_map.h:141-163), both evaluated by
 * the host parser from the CIGAR. Same 16-byte layout as thj_hit in include/thj.h. */
typedef struct {
    uint32_t ref_id;     /* 1-based, @SQ order (bwt_map.h:608-632) */
    int32_t  left;
    int32_t  right;
    uint8_t  flags;      /* bit0 antisense_align, bit1 end() */
    uint8_t  edit_dist;
    uint8_t  mismatches;
    uint8_t  read_len;
} orc_hit;

#define ORC_HIT_ANTISENSE 1u
#define ORC_HIT_END       2u

typedef struct {
    int32_t segment_length;        /* common.cpp:121 */
    int32_t segment_mismatches;    /* common.cpp:122 */
    int32_t min_segment_intron;    /* common.cpp:115 */
    int32_t max_segment_intron;    /* common.cpp:116 */
    int32_t max_insertion_length;  /* common.cpp:98 */
    int32_t max_deletion_length;   /* common.cpp:99 */
    int32_t max_seg_multihits;     /* common.cpp:135 */
    int32_t inner_dist_mean;       /* common.cpp:101 */
    int32_t inner_dist_std_dev;    /* common.cpp:102 */
    int32_t library_type;          /* common.h:155-167: 0 none,1 fr-unstranded,2 fr-firststrand,3 fr-secondstrand,... */
    int32_t bowtie2;               /* common.cpp:79 */
    int32_t read_side;             /* segments.h:13-18: 1 = READ_LEFT, 2 = READ_RIGHT */
} orc_params;

/* Genome as ASCII: seq[i] is contig with ref_id i+1 (upper-case ACGT, anything
 * else already folded to 'N' as SeqAn's char->Dna5 conversion does), or NULL
 * when the FASTA has no record for that @SQ entry (segment_juncs.cpp:2105-2108). */
typedef struct {
    int32_t        n_contigs;
    const char**   seq;
    const int64_t* len;
} orc_genome;

/* A batch of reads of one side with their per-segment hit lists, i.e. the
 * sequence of hits_for_read vectors that look_for_hit_group /
 * process_next_hit_group (segment_juncs.cpp:3823-4123) hand to the finders, in
 * visiting (= increasing read id) order. */
typedef struct {
    int32_t         n_reads;
    int32_t         nseg;
    const uint32_t* read_id;       /* [n_reads] */
    const int64_t*  read_off;      /* [n_reads+1] into bases */
    const char*     bases;         /* ASCII read sequences, as ReadStream returns them */
    const int64_t*  seg_off;       /* [n_reads*nseg+1] CSR into hits, index r*nseg+s */
    const orc_hit*  hits;
    const int64_t*  mate_off;      /* [n_reads+1] CSR into mate_hits (may be NULL = no mates) */
    const orc_hit*  mate_hits;     /* partner_hit_group of find_gaps (segment_juncs.cpp:3321-3348) */
} orc_batch;

typedef struct { uint32_t ref_id, left, right, antisense; } orc_junction;   /* junctions.h:27-57 */
typedef struct { uint32_t ref_id, left; char seq[16]; uint64_t prio; } orc_insertion; /* insertions.h:31-67 */

typedef struct {
    orc_junction*  juncs;       int64_t n_juncs;       /* sorted unique (junctions.h:39-57 order) */
    orc_junction*  deletions;   int64_t n_deletions;   /* sorted unique, antisense always 0 */
    orc_insertion* insertions;  int64_t n_insertions;  /* sorted by (ref,left,len); first inserted wins */
    int64_t n_windows;          /* RefSeg windows scanned (statistic) */
    int64_t n_indel_pairs;      /* hit pairs sent to detect_small_* (statistic) */
    int64_t n_rescue_pairs;     /* (hit, mate-hit) pairs scanned by map_read_to_contig (statistic) */
} orc_events;

/* Runs find_insertions_and_deletions + find_gaps over every read of the batch
 * (the per-read body of process_next_hit_group, segment_juncs.cpp:4094-4118,
 * fusion search off).  Returns 0, or <0 on allocation failure. */
int  orc_segjuncs_batch(const orc_params* p, const orc_genome* g, const orc_batch* b, orc_events* out);
void orc_events_free(orc_events* e);

/* Primitive entry points, exposed so kernels can be checked one at a time. */

/* juncs_from_ref_segs<RecordSegmentJuncs>, POINT_DIR_BOTH, one window, all three
 * motif pairs (segment_juncs.cpp:2052-2377, :3618-3649).  Appends to out[],
 * returns the number appended (<= cap). */
int orc_window_scan(const orc_params* p, const orc_genome* g, uint32_t ref_id,
                    int32_t seg_left, int32_t seg_right, int antisense,
                    const char* support, int support_len,
                    orc_junction* out, int cap);

/* simpleSplitAlignment (segment_juncs.cpp:2390-2456): returns the FIRST best
 * insert position (or -1 when len<2) and the mismatch count through *min_err. */
int orc_simple_split(const char* shorter, const char* left_ref, const char* right_ref,
                     int len, int* min_err);

/* map_read_to_contig (segment_juncs.cpp:2946-2973). */
int orc_map_read_to_contig(const char* contig, int contig_len, const char* read, int read_len);


/* ===================== fusion search of segment_juncs (segjuncs_oracle.c) ===================== */
typedef struct { uint32_t ref_id1, ref_id2, left, right, dir, count, edit_dist, skip; } orc_fusion;   /* fusions.h:24-116 */
#define ORC_FUSION_FF 7u
#define ORC_FUSION_FR 8u
#define ORC_FUSION_RF 9u
#define ORC_FUSION_RR 10u
/* find_fusions + detect_fusion (segment_juncs.cpp:2976-3291, :2629-2805) over every read of the batch
 * (ALL visited reads, including those whose only mapped segment is the first).  Returns the FusionSimpleSet
 * in Fusion::operator< order (fusions.h:38-69); *out is malloc'd. */
int orc_fusions_batch(const orc_params* p, int fusion_anchor_length, int fusion_min_dist,
                      const orc_genome* g, const orc_batch* b, const uint32_t* ignore_ref_ids /* --fusion-ignore-chromosomes */,
                      int n_ignore, orc_fusion** out, int64_t* n_out);
/* The output filter of the fusion writer (segment_juncs.cpp:5096-5182): marks `skip`; juncs = the final junction
 * set (sorted).  Returns nothing; entries with skip != 0 are not written. */
void orc_fusion_filter(orc_fusion* f, int64_t n, const orc_junction* juncs, int64_t n_juncs);

/* juncs_db (juncs_db.cpp): FASTA text of the junction database; inputs already in their std::set orders.  Returns a
 * malloc'd NUL-terminated string (orc_free). */
char* orc_juncs_db(const orc_genome* g, const char* const* names, int read_len, int min_anchor_len,
                   const orc_junction* juncs, int64_t n_juncs, const orc_junction* dels, int64_t n_dels,
                   const uint32_t* ins_ref, const uint32_t* ins_left, const char* const* ins_seq, int64_t n_ins,
                   const orc_fusion* fus, int64_t n_fus);

/* Coverage search of segment_juncs (covsearch_oracle.c; segment_juncs.cpp:4268-4543 and what it calls): `hits` = every
 * record of every segment map of both sides (only ref_id / left / right are read); the initially unmapped reads as one
 * base string with n_ium + 1 offsets; -> junctions in Junction::operator< order (malloc'd, orc_free). */
int orc_coverage_search(const orc_genome* g, const orc_hit* hits, int64_t n_hits,
                        const char* ium_bases, const int64_t* ium_off, int64_t n_ium,
                        int min_cov_length, int min_intron, int max_intron, int64_t max_juncs,
                        orc_junction** out, int64_t* n_out);

/* Butterfly search of segment_juncs (covsearch_oracle.c; --butterfly-search: segment_juncs.cpp:4178-4249 pair_covered_sites,
 * :1698-2049 RecordButterflyJuncs, :466-501 the pruned and compacted extension table): same inputs as the coverage search; min_intron /
 * max_intron = min / max_coverage_intron_length, max_juncs = max_cov_juncs.  -> junctions in Junction::operator< order (malloc'd, orc_free). */
int orc_butterfly_search(const orc_genome* g, const orc_hit* hits, int64_t n_hits,
                         const char* ium_bases, const int64_t* ium_off, int64_t n_ium,
                         int min_intron, int max_intron, int64_t max_juncs,
                         orc_junction** out, int64_t* n_out);

/* Microexon search of segment_juncs (covsearch_oracle.c; segment_juncs.cpp:3880-3941 window registration, :3675-3735 merging,
 * :3737-3815 per-window pairing): the batches of both sides in the order the reference visits them (all left reads, then all right
 * reads), sides[i] = 1 (READ_LEFT) / 2 (READ_RIGHT); p->segment_length and p->library_type are read; min_intron =
 * min_coverage_intron_length, max_juncs = max_cov_juncs.  -> junctions in Junction::operator< order (malloc'd, orc_free). */
int orc_microexon_search(const orc_params* p, int min_anchor_len, int min_intron, int64_t max_juncs, const orc_genome* g,
                         const orc_batch* const* batches, const int* sides, int n_batches, orc_junction** out, int64_t* n_out, int64_t* n_windows);

/* ===================== long_spanning_reads (spanning_oracle.c) ===================== */

/* CigarOpCode values of bwt_map.h:36-55 */
enum { ORC_MATCH = 1, ORC_mATCH = 2, ORC_INS = 3, ORC_iNS = 4, ORC_DEL = 5, ORC_dEL = 6,
       ORC_REF_SKIP = 11, ORC_rEF_SKIP = 12, ORC_SOFT_CLIP = 13, ORC_HARD_CLIP = 14, ORC_PAD = 15 };
#define ORC_CIG(op, len) (((uint32_t)(op) << 28) | ((uint32_t)(len) & 0x0FFFFFFFu))
#define ORC_CIG_OP(c)  ((int)((c) >> 28))
#define ORC_CIG_LEN(c) ((uint32_t)((c) & 0x0FFFFFFFu))

/* A segment alignment with its CIGAR (BowtieHit, bwt_map.h:36-536), 32 bytes.
 * Same layout as thj_span_hit in include/thj.h. */
typedef struct {
    uint32_t ref_id;
    int32_t  left;
    uint8_t  flags;        /* bit0 antisense_align, bit1 end(), bit2 antisense_splice */
    uint8_t  mismatches;
    uint8_t  edit_dist;
    uint8_t  n_cigar;      /* 1..5 */
    uint32_t cigar[5];     /* ORC_CIG(op, len) */
} orc_span_hit;
#define ORC_HIT_ANTISENSE_SPLICE 4u
/* a segment hit on an rf / rr fusion contig of the junction database: antisense_align is the opposite of the record's own
 * strand flag (bwt_map.cpp:1744-1745), so the record's SEQ is the read piece reverse-complemented iff antisense ^ this bit.
 * A hit with a fusion op (ORC_FUSION_*, length = the position on the second contig) has at most 4 ops and keeps ref_id2
 * in cigar[4]. */
#define ORC_HIT_STRAND_FLIPPED 8u

typedef struct {
    int32_t segment_length;
    int32_t max_insertion_length, max_deletion_length;
    int32_t min_report_intron, max_report_intron;
    int32_t max_seg_multihits;
    int32_t read_mismatches, read_gap_length, read_edit_dist;
    int32_t bowtie2;
    int32_t bowtie2_max_penalty, bowtie2_min_penalty, bowtie2_penalty_for_N;
    int32_t bowtie2_read_gap_open, bowtie2_read_gap_cont, bowtie2_ref_gap_open, bowtie2_ref_gap_cont;
} orc_span_params;

typedef struct {
    int32_t         n_reads, nseg;
    const int64_t*  read_off;      /* [n_reads+1] into bases / quals */
    const char*     bases;
    const char*     quals;         /* phred+33 as in the FASTQ */
    const int64_t*  seg_off;       /* [n_reads*nseg+1] */
    const orc_span_hit* hits;      /* per (read, segment): contig hits then spliced hits
                                      (long_spanning_reads.cpp:2706-2765, :87-163) */
} orc_span_batch;

typedef struct { uint32_t ref_id, left; char seq[16]; } orc_ins_in;

typedef struct {
    int32_t  read_idx;
    uint32_t ref_id;
    int32_t  left;
    uint8_t  antisense, antisense_splice, mismatches, edit_dist;
    int32_t  n_cigar;
    uint32_t cigar[24];
    int32_t  AS, XM, XO, XG;
    char     md[96];
} orc_aln;

/* join_segments_for_read + sort/unique + filters + bowtie_sam_extra for every read of the batch
 * (JoinSegmentsWorker::operator(), long_spanning_reads.cpp:2669-2845), fusion search off.
 * juncs must be sorted by Junction::operator< (junctions.h:39-57) and hold the deletions too
 * (long_spanning_reads.cpp:2897-2944); insertions sorted by (ref,left,len).  *out is malloc'd. */
int orc_spanning_batch(const orc_span_params* p, const orc_genome* g, const orc_span_batch* b,
                       const orc_junction* juncs, int64_t n_juncs, const orc_ins_in* ins, int64_t n_ins,
                       orc_aln** out, int64_t* n_out);
void orc_free(void* p);
/* An MD string that does not fit a record's md field goes to a per-thread pool; the field then holds "\x01<offset>".
 * (Adversarial test batches only: a real alignment's MD is a few dozen characters.) */
void orc_long_md_reset(void);
int64_t orc_long_md_put(const char* s);
const char* orc_long_md(int64_t off);

/* ---- the same with the fusion branches (spanning_fusion_oracle.c): --fusion-search of long_spanning_reads.
 * fusions = the .fusions list (long_spanning_reads.cpp:2998-3040) sorted by Fusion::operator< (fusions.h:44-71), dir = ORC_FUSION_*. */
typedef struct { uint32_t ref1, ref2, left, right, dir; } orc_fusion_in;
typedef struct {
    int32_t  read_idx;
    uint32_t ref_id, ref_id2;
    int32_t  left;
    uint8_t  antisense, antisense_splice, mismatches, edit_dist;
    int32_t  n_cigar;
    uint32_t cigar[32];
    int32_t  AS, XM, XO, XG;
    char     md[128];
} orc_faln;
int orc_spanning_batch_fusion(const orc_span_params* p, int fusion_search, int fusion_min_dist, const orc_genome* g, const orc_span_batch* b,
                              const orc_junction* juncs, int64_t n_juncs, const orc_ins_in* ins, int64_t n_ins,
                              const orc_fusion_in* fusions, int64_t n_fusions, orc_faln** out, int64_t* n_out);


/* ---- junction consensus of tophat_reports (juncbed_oracle.c): the reported alignments'
 * REF_SKIPs reduced to the JunctionSet that becomes junctions.bed.  A record = the fields of an alignment the reduce
 * reads; cigar = ORC_CIG(op, len) with CigarOpCode values (bwt_map.h:36-55). */
typedef struct { uint32_t ref_id; int32_t left; uint8_t antisense_splice; uint8_t n_cigar; uint16_t reserved; uint32_t cigar[16]; } orc_jrec;
typedef struct { uint32_t ref_id, left, right, antisense, left_extent, right_extent, support, reserved; } orc_jstat;
int   orc_junction_consensus(const orc_jrec* recs, int64_t n_recs, int min_anchor_len, orc_jstat** out, int64_t* n_out);
char* orc_junctions_bed(const orc_jstat* j, int64_t n, const char* const* names);

#ifdef __cplusplus
}
#endif
#endif
