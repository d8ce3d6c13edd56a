/* BENCH INFRASTRUCTURE ONLY (see thj_cpuport.cpp): the entry points the executables reference and the CPU port does not answer.
 * The device-side ingest and the device-side BAM writer decline (THJ_EFALLBACK = -6: the executables' documented way onto their
 * host readers and host writer); the searches outside this figure report an error.  Declared without prototypes on purpose: this
 * file does not include thj.h, the callers' arguments are ignored. */
void thj_cpuport_error(const char* what, int declined);      /* thj_cpuport.cpp */
static int decline(const char* what) { thj_cpuport_error(what, 1); return -6; }
static int refuse(const char* what) { thj_cpuport_error(what, 0); return -1; }
int thj_ingest_seg_batch() { return decline("ingest"); }
int thj_ingest_span_batch() { return decline("ingest"); }
int thj_ingest_span_hits() { return decline("ingest"); }
int thj_span_bam_encode() { return decline("BAM encoding"); }
int thj_bgzf_deflate() { return decline("DEFLATE"); }
int thj_span_batch_reads_host() { return refuse("thj_span_batch_reads_host"); }
int thj_span_batch_attach_reads() { return refuse("thj_span_batch_attach_reads"); }
int thj_span_fusions_upload() { return refuse("fusion search"); }
int thj_fusion_reset_async() { return 0; }
int thj_fusion_set_ignored() { return 0; }
int thj_fusion_run_async() { return refuse("fusion search"); }
int thj_fusion_finish() { return refuse("fusion search"); }
int thj_fusion_download() { return refuse("fusion search"); }
int thj_fusion_allgather() { return refuse("fusion search"); }
int thj_covsearch_reset_async() { return 0; }
int thj_covsearch_add_reads() { return refuse("coverage search"); }
int thj_covsearch_add_hits_async() { return refuse("coverage search"); }
int thj_covsearch_run_async() { return refuse("coverage search"); }
int thj_covsearch_finish() { return refuse("coverage search"); }
int thj_covsearch_allgather() { return refuse("coverage search"); }
int thj_covsearch_add_reads_bam() { return decline("ingest"); }
int thj_covsearch_reserve_reads() { return 0; }
int thj_butterfly_run() { return refuse("butterfly search"); }
int thj_microexon_collect() { return refuse("microexon search"); }
int thj_microexon_candidates() { return refuse("microexon search"); }
int thj_microexon_run() { return refuse("microexon search"); }
int thj_comm_create_local() { return refuse("the exchange step (one context)"); }
int thj_events_allgather_async() { return refuse("the exchange step (one context)"); }
