// thj_hostio.h -- host side of the drop-in binaries: option table, reference table, FASTA/FASTQ/BAM/SAM readers,
// hit parsing, id-grouped hit streams, text + BAM writers.  Plain C++17 + zlib, no device code; the executables
// call the kernels only through include/thj.h.
//
// Behavioural sources (reference v2.1.2, src/): common.cpp:262-422,:459-720 (options); bwt_map.h:579-788
// (RefSequenceTable); segment_juncs.cpp:64-88 (get_seqs); reads.cpp:94-188,:528-630 (read fetch);
// bwt_map.cpp:1101-1452 (BAMHitFactory::get_hit_from_buf); bwt_map.h:1155-1220 (HitStream::next_read_hits);
// common.cpp:1000-1173 + common.h:401-627 (GBamRecord / GBamWriter); samtools-0.1.18 bgzf.c (BGZF framing).
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <dlfcn.h>
#include <getopt.h>
#include <stdint.h>
#include <unistd.h>
#include <zlib.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <signal.h>
#include <cerrno>
#include "thj_fastdeflate.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/thj.h"

namespace thjh {

// a file under construction that must not outlive a failed process (the packed-genome cache's `.tmp.<pid>`): die() removes it
inline char* pending_tmp_path() { static char path[4096] = {0}; return path; }
[[noreturn]] inline void die(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    if (pending_tmp_path()[0]) unlink(pending_tmp_path());
    exit(1);
}

// an output file is complete only if every write and the close succeeded (a full disk shows up here, not at fprintf)
inline void close_output(FILE* f, const char* what) {
    if (!f) return;
    const bool bad = ferror(f) != 0;
    if (fclose(f) != 0 || bad) die("Error: writing %s failed (%s)\n", what, strerror(errno));
}

// ------------------------------------------------------------------ "outputs are complete" hand-off
// A process that has used the GPU takes ~0.2 s to leave after its last instruction (the driver frees its device memory, queues and
// pinned pages; measured with the time stamps of PhaseTimer::report): three processes per run, 0.6 of 3.2 s on 8 M pairs.  Nobody
// has to wait for that: main() runs the real work in a child, the process the caller started returns the moment the child
// reports that every output file is written and closed, and the child's teardown overlaps whatever the caller does next.  A child
// that ends without reporting (usage errors, die()) is waited for and its exit code passed on.  THJ_NO_HANDOFF=1: one process
// (it still leaves with _exit: the kernel frees what the runtime's exit handlers would walk through; THJ_EXIT_HANDLERS=1 runs them, for profilers).
inline int& handoff_fd() { static int fd = -1; return fd; }
inline int run_with_handoff(int argc, char** argv, int (*body)(int, char**)) {
    int fd[2];
    // Off by default since round 3: with the BAM records made on the device the process holds little page-locked memory and leaves with
    // _exit, and one process (1.28 s for the three stages of 10 M pairs) beat parent + child (1.40 s).  THJ_HANDOFF=1 turns it on.
    if (!getenv("THJ_HANDOFF") || getenv("THJ_NO_HANDOFF") || pipe2(fd, O_CLOEXEC) != 0) return body(argc, argv);   // O_CLOEXEC: a popen'd packer must not hold the write end
    fflush(nullptr);
    const pid_t self = getpid();
    const pid_t pid = fork();
    if (pid < 0) { close(fd[0]); close(fd[1]); return body(argc, argv); }
    if (pid == 0) {
        close(fd[0]);
        prctl(PR_SET_PDEATHSIG, SIGKILL);                 // never outlive the process the caller knows about, except to finish dying
        if (getppid() != self) _exit(1);                  // the parent went away before the line above took effect
        handoff_fd() = fd[1];
        const int rc = body(argc, argv);
        fflush(nullptr);
        _exit(rc);
    }
    close(fd[1]);
    unsigned char code = 0; ssize_t n;
    do n = read(fd[0], &code, 1); while (n < 0 && errno == EINTR);
    if (n == 1) return (int)code;
    int st = 0;
    while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
    return WIFEXITED(st) ? WEXITSTATUS(st) : 1;
}
// Last call of a body whose outputs are all on disk: report, then leave without the exit handlers (the HIP runtime's would only
// add to the teardown).  The standard streams go to /dev/null first so that a caller reading our pipes sees their end now.
// THJ_EXIT_PROBE=1 (developer aid): take the process apart by hand before leaving and say what each part costs -- the file mappings,
// then everything the HIP runtime holds (hipDeviceReset) -- so that the time a caller sees between the report and the process's end
// can be put down to one of them
inline void exit_probe() {
    if (!getenv("THJ_EXIT_PROBE")) return;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    size_t unmapped = 0;
    {
        std::vector<std::pair<uintptr_t, uintptr_t>> maps;
        if (FILE* f = fopen("/proc/self/maps", "r")) {
            char line[1024];
            while (fgets(line, sizeof line, f)) {
                unsigned long a, b; char perms[8], path[768]; path[0] = 0;
                if (sscanf(line, "%lx-%lx %7s %*s %*s %*s %767s", &a, &b, perms, path) >= 3 && strstr(path, ".bam")) maps.emplace_back(a, b);
            }
            fclose(f);
        }
        for (auto& m : maps) { munmap((void*)m.first, m.second - m.first); unmapped += m.second - m.first; }
    }
    double t1 = now();
    typedef int (*reset_fn)();
    void* h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("libamdhip64.so.7", RTLD_NOW | RTLD_NOLOAD);
    reset_fn rf = h ? (reset_fn)dlsym(h, "hipDeviceReset") : nullptr;
    if (rf) rf();
    double t2 = now();
    fprintf(stderr, "[exit-probe] munmap of %.2f GB of .bam mappings %.3f s, hipDeviceReset %.3f s%s\n", unmapped / 1073741824.0, t1 - t0, t2 - t1, rf ? "" : " (not found)");
    const double nowu = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[exit-probe] unix time after the probe %.6f\n", nowu);
}
[[noreturn]] inline void finish_outputs_complete(int rc) {
    exit_probe();
    fflush(nullptr);
    if (handoff_fd() >= 0) {
        const int nul = open("/dev/null", O_RDWR);
        if (nul >= 0) { dup2(nul, 0); dup2(nul, 1); dup2(nul, 2); }
        // THJ_HANDOFF_LINGER_MS=n: sleep n ms before going, so that this process's teardown (device memory, page-locked buffers) does not
        // meet the start-up of whatever the caller runs next in the driver -- a knob for the occasional slow start of a stage run
        // right after another (1.8 s instead of 1.0, one run in four on some boxes); off by default: eight runs each way showed none.
        static const int linger_ms = getenv("THJ_HANDOFF_LINGER_MS") ? atoi(getenv("THJ_HANDOFF_LINGER_MS")) : 0;
        if (linger_ms > 0) prctl(PR_SET_PDEATHSIG, 0);            // the parent is about to leave: do not die with it just yet
        const unsigned char c = (unsigned char)rc;
        if (write(handoff_fd(), &c, 1) != 1) {}
        if (linger_ms > 0) usleep((useconds_t)linger_ms * 1000u);
    } else if (getenv("THJ_EXIT_HANDLERS")) exit(rc);             // a profiler wants the process to leave through the exit handlers (with THJ_NO_HANDOFF=1: one process)
    _exit(rc);
}

// ------------------------------------------------------------------ options (common.cpp:79-180, :262-720)
// Wall-clock per phase, printed to stderr at exit when THJ_TIMING is set (developer aid; no effect on results).
struct PhaseTimer {
    std::vector<std::pair<std::string, double>> acc;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* name) {
        auto now = std::chrono::steady_clock::now();
        double dt = std::chrono::duration<double>(now - t0).count();
        t0 = now;
        for (auto& a : acc) if (a.first == name) { a.second += dt; return; }
        acc.emplace_back(name, dt);
    }
    void report() const {
        if (!getenv("THJ_TIMING")) return;
        double tot = 0;
        for (auto& a : acc) tot += a.second;
        for (auto& a : acc) fprintf(stderr, "[timing] %-32s %8.3f s\n", a.first.c_str(), a.second);
        fprintf(stderr, "[timing] %-32s %8.3f s\n", "total", tot);
        // wall-clock stamps (the timer is a static object: constructed before main) so that a parent can see what the process spent
        // before its first and after its last instruction (tools/e2e_bench.py: loader and exit time)
        const double now = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
        fprintf(stderr, "[timing] %-32s %.6f %.6f\n", "unix time at start / report", wall0, now);
    }
    double wall0 = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
};

// Per-phase time summed over the shard workers (THJ_TIMING): where a parallel run spends its thread-seconds.
struct WorkClock {
    std::atomic<long long> ns[4] = {{0}, {0}, {0}, {0}};     // 0 whole shard, 1 waiting for the GPU's lock, 2 device calls, 3 encode / other
    static long long now() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void add(int k, long long t0) { ns[k] += now() - t0; }
    void report(const char* const names[4]) const {
        if (!getenv("THJ_TIMING")) return;
        for (int k = 0; k < 4; ++k) fprintf(stderr, "[worker-seconds] %-40s %8.3f\n", names[k], (double)ns[k].load() * 1e-9);
    }
};

struct Opts {
    thj_params p;
    bool no_coverage_search = false, no_microexon_search = false, butterfly_search = false, fusion_search = false;
    bool cov_state = false;         // segment_juncs: the coverage map and the extension table are kept (coverage or butterfly search)
    bool color = false, bowtie2 = true, fusion_do_not_resolve_conflicts = false;
    std::string fusion_ignore;
    int num_threads = 1;
    std::string sam_header, ium_reads, zpacker;
    int min_coverage_intron = 50, max_coverage_intron = 20000;      // common.cpp:112-113
};

enum {
    O_FASTA = 127, O_FASTQ, O_MIN_ANCHOR, O_SAM_HEADER, O_RG_ID, O_SPLICE_MM, O_VERBOSE, O_INNER_MEAN, O_INNER_SD, O_OUTDIR,
    O_GENE_FILTER, O_GTF, O_MAX_MULTIHITS, O_SUPPRESS, O_MAX_SEG_MULTIHITS, O_NO_CLOSURE, O_NO_COVERAGE, O_NO_MICROEXON,
    O_SEG_LEN, O_SEG_MM, O_READ_MM, O_READ_GAP, O_READ_ED, O_READ_REALIGN_ED, O_MIN_CLOSURE_EXON, O_MIN_CLOSURE_INTRON,
    O_MAX_CLOSURE_INTRON, O_MIN_COV_INTRON, O_MAX_COV_INTRON, O_MIN_SEG_INTRON, O_MAX_SEG_INTRON, O_MIN_REP_INTRON,
    O_MAX_REP_INTRON, O_MIN_ISO, O_IUM, O_BUTTERFLY, O_SOLEXA, O_PHRED64, O_QUALS, O_INT_QUALS, O_COLOR, O_LIBTYPE,
    O_MAX_DEL, O_MAX_INS, O_THREADS, O_ZPACKER, O_SAMTOOLS, O_AUX_OUT, O_STD_OUT, O_INDEX_OUT, O_GTF_JUNCS, O_FLT_READS,
    O_FLT_HITS, O_FLT_SIDE, O_SECONDARY, O_DISCORDANT, O_MIXED, O_FUSION, O_FUSION_ANCHOR, O_FUSION_MIN_DIST,
    O_FUSION_READ_MM, O_FUSION_MULTIREADS, O_FUSION_MULTIPAIRS, O_FUSION_IGNORE, O_FUSION_NO_RESOLVE, O_BOWTIE1,
    O_B2_MIN_SCORE, O_B2_MAX_PEN, O_B2_MIN_PEN, O_B2_N_PEN, O_B2_RDG_OPEN, O_B2_RDG_CONT, O_B2_RFG_OPEN, O_B2_RFG_CONT,
    O_B2_SCOREFLT
};

inline int parse_int(int lower, const char* msg) {
    char* e = nullptr;
    long v = strtol(optarg, &e, 10);
    if (e == optarg || *e != 0 || v < lower) die("%s\n", msg);        // parseIntOpt, common.cpp:186-204
    return (int)v;
}

// Accepts every option of the shared table (the driver passes its full params.cmd() to every binary,
// tophat.py:824-900); unknown option => usage + exit status 1 (common.cpp:714-716).
inline int parse_options(int argc, char** argv, Opts& o, void (*usage)()) {
    static const struct option lo[] = {
        {"fasta", 0, 0, O_FASTA}, {"fastq", 0, 0, O_FASTQ}, {"min-anchor", 1, 0, O_MIN_ANCHOR}, {"sam-header", 1, 0, O_SAM_HEADER},
        {"rg-id", 1, 0, O_RG_ID}, {"splice-mismatches", 1, 0, O_SPLICE_MM}, {"verbose", 0, 0, O_VERBOSE},
        {"inner-dist-mean", 1, 0, O_INNER_MEAN}, {"inner-dist-std-dev", 1, 0, O_INNER_SD}, {"output-dir", 1, 0, O_OUTDIR},
        {"gene-filter", 1, 0, O_GENE_FILTER}, {"gtf-annotations", 1, 0, O_GTF}, {"max-multihits", 1, 0, O_MAX_MULTIHITS},
        {"suppress-hits", 0, 0, O_SUPPRESS}, {"max-seg-multihits", 1, 0, O_MAX_SEG_MULTIHITS}, {"no-closure-search", 0, 0, O_NO_CLOSURE},
        {"no-coverage-search", 0, 0, O_NO_COVERAGE}, {"no-microexon-search", 0, 0, O_NO_MICROEXON}, {"segment-length", 1, 0, O_SEG_LEN},
        {"segment-mismatches", 1, 0, O_SEG_MM}, {"read-mismatches", 1, 0, O_READ_MM}, {"read-gap-length", 1, 0, O_READ_GAP},
        {"read-edit-dist", 1, 0, O_READ_ED}, {"read-realign-edit-dist", 1, 0, O_READ_REALIGN_ED}, {"min-closure-exon", 1, 0, O_MIN_CLOSURE_EXON},
        {"min-closure-intron", 1, 0, O_MIN_CLOSURE_INTRON}, {"max-closure-intron", 1, 0, O_MAX_CLOSURE_INTRON},
        {"min-coverage-intron", 1, 0, O_MIN_COV_INTRON}, {"max-coverage-intron", 1, 0, O_MAX_COV_INTRON},
        {"min-segment-intron", 1, 0, O_MIN_SEG_INTRON}, {"max-segment-intron", 1, 0, O_MAX_SEG_INTRON},
        {"min-report-intron", 1, 0, O_MIN_REP_INTRON}, {"max-report-intron", 1, 0, O_MAX_REP_INTRON},
        {"min-isoform-fraction", 1, 0, O_MIN_ISO}, {"ium-reads", 1, 0, O_IUM}, {"butterfly-search", 0, 0, O_BUTTERFLY},
        {"solexa-quals", 0, 0, O_SOLEXA}, {"phred64-quals", 0, 0, O_PHRED64}, {"quals", 0, 0, O_QUALS}, {"integer-quals", 0, 0, O_INT_QUALS},
        {"color", 0, 0, O_COLOR}, {"library-type", 1, 0, O_LIBTYPE}, {"max-deletion-length", 1, 0, O_MAX_DEL},
        {"max-insertion-length", 1, 0, O_MAX_INS}, {"num-threads", 1, 0, O_THREADS}, {"zpacker", 1, 0, O_ZPACKER},
        {"samtools", 1, 0, O_SAMTOOLS}, {"aux-outfile", 1, 0, O_AUX_OUT}, {"outfile", 1, 0, O_STD_OUT}, {"index-outfile", 1, 0, O_INDEX_OUT},
        {"gtf-juncs", 1, 0, O_GTF_JUNCS}, {"flt-reads", 1, 0, O_FLT_READS}, {"flt-hits", 1, 0, O_FLT_HITS}, {"flt-side", 1, 0, O_FLT_SIDE},
        {"report-secondary-alignments", 0, 0, O_SECONDARY}, {"report-discordant-pair-alignments", 0, 0, O_DISCORDANT},
        {"report-mixed-alignments", 0, 0, O_MIXED}, {"fusion-search", 0, 0, O_FUSION}, {"fusion-anchor-length", 1, 0, O_FUSION_ANCHOR},
        {"fusion-min-dist", 1, 0, O_FUSION_MIN_DIST}, {"fusion-read-mismatches", 1, 0, O_FUSION_READ_MM},
        {"fusion-multireads", 1, 0, O_FUSION_MULTIREADS}, {"fusion-multipairs", 1, 0, O_FUSION_MULTIPAIRS},
        {"fusion-ignore-chromosomes", 1, 0, O_FUSION_IGNORE}, {"fusion-do-not-resolve-conflicts", 0, 0, O_FUSION_NO_RESOLVE},
        {"bowtie1", 0, 0, O_BOWTIE1}, {"bowtie2-min-score", 1, 0, O_B2_MIN_SCORE}, {"bowtie2-max-penalty", 1, 0, O_B2_MAX_PEN},
        {"bowtie2-min-penalty", 1, 0, O_B2_MIN_PEN}, {"bowtie2-penalty-for-N", 1, 0, O_B2_N_PEN},
        {"bowtie2-read-gap-open", 1, 0, O_B2_RDG_OPEN}, {"bowtie2-read-gap-cont", 1, 0, O_B2_RDG_CONT},
        {"bowtie2-ref-gap-open", 1, 0, O_B2_RFG_OPEN}, {"bowtie2-ref-gap-cont", 1, 0, O_B2_RFG_CONT}, {0, 0, 0, 0}};
    thj_params_default(&o.p);
    int c, idx = 0;
    while ((c = getopt_long(argc, argv, "QCp:z:N:w:W:", lo, &idx)) != -1) {
        switch (c) {
        case O_MIN_ANCHOR: o.p.min_anchor_len = parse_int(3, "--min-anchor arg must be at least 3"); break;
        case O_SAM_HEADER: o.sam_header = optarg; break;
        case O_INNER_MEAN: o.p.inner_dist_mean = parse_int(-1024, "--inner-dist-mean arg must be at least -1024"); break;
        case O_INNER_SD: o.p.inner_dist_std_dev = parse_int(0, "--inner-dist-std-dev arg must be at least 0"); break;
        case O_MAX_SEG_MULTIHITS: o.p.max_seg_multihits = parse_int(1, "--max-seg-multihits arg must be at least 1"); break;
        case O_NO_COVERAGE: o.no_coverage_search = true; break;
        case O_NO_MICROEXON: o.no_microexon_search = true; break;
        case O_BUTTERFLY: o.butterfly_search = true; break;
        case O_SEG_LEN: o.p.segment_length = parse_int(4, "--segment-length arg must be at least 4"); break;
        case O_SEG_MM: o.p.segment_mismatches = parse_int(0, "--segment-mismatches arg must be at least 0"); break;
        case 'N': case O_READ_MM: o.p.read_mismatches = parse_int(0, "--read-mismatches arg must be at least 0"); break;
        case O_READ_GAP: o.p.read_gap_length = parse_int(0, "--read-gap-length arg must be at least 0"); break;
        case O_READ_ED: o.p.read_edit_dist = parse_int(0, "--read-edit-dist arg must be at least 0"); break;
        case O_MIN_SEG_INTRON: o.p.min_segment_intron = parse_int(1, "--min-segment-intron arg must be at least 1"); break;
        case O_MAX_SEG_INTRON: o.p.max_segment_intron = parse_int(1, "--max-segment-intron arg must be at least 1"); break;
        case O_MIN_REP_INTRON: o.p.min_report_intron = parse_int(1, "--min-report-intron arg must be at least 1"); break;
        case O_MAX_REP_INTRON: o.p.max_report_intron = parse_int(1, "--max-report-intron arg must be at least 1"); break;
        case O_IUM: o.ium_reads = optarg; break;
        case O_MIN_COV_INTRON: o.min_coverage_intron = parse_int(1, "--min-coverage-intron arg must be at least 1"); break;
        case O_MAX_COV_INTRON: o.max_coverage_intron = parse_int(1, "--max-coverage-intron arg must be at least 1"); break;
        case 'C': case O_COLOR: o.color = true; break;
        case O_LIBTYPE:
            if (!strcmp(optarg, "fr-unstranded")) o.p.library_type = 1;
            else if (!strcmp(optarg, "fr-firststrand")) o.p.library_type = 2;
            else if (!strcmp(optarg, "fr-secondstrand")) o.p.library_type = 3;
            else if (!strcmp(optarg, "ff-unstranded")) o.p.library_type = 4;
            else if (!strcmp(optarg, "ff-firststrand")) o.p.library_type = 5;
            else if (!strcmp(optarg, "ff-secondstrand")) o.p.library_type = 6;
            break;
        case O_MAX_DEL: o.p.max_deletion_length = parse_int(0, "--max-deletion-length must be at least 0"); break;
        case O_MAX_INS: o.p.max_insertion_length = parse_int(0, "--max-insertion-length must be at least 0"); break;
        case 'p': case O_THREADS: o.num_threads = parse_int(1, "-p/--num-threads must be at least 1"); break;
        case 'z': case O_ZPACKER: o.zpacker = optarg; break;
        case O_FUSION: o.fusion_search = true; break;
        case O_FUSION_ANCHOR: o.p.fusion_anchor_length = parse_int(10, "--fusion-anchor-length must be at least 10"); break;
        case O_FUSION_MIN_DIST: o.p.fusion_min_dist = parse_int(0, "--fusion-min-dist must be at least 0"); break;
        case O_FUSION_NO_RESOLVE: o.fusion_do_not_resolve_conflicts = true; break;
        case O_FUSION_IGNORE: o.fusion_ignore = optarg; break;
        case O_BOWTIE1: o.bowtie2 = false; o.p.bowtie2 = 0; break;
        case O_B2_MAX_PEN: o.p.bowtie2_max_penalty = parse_int(0, "--bowtie2-max-penalty must be at least 0"); break;
        case O_B2_MIN_PEN: o.p.bowtie2_min_penalty = parse_int(0, "--bowtie2-min-penalty must be at least 0"); break;
        case O_B2_N_PEN: o.p.bowtie2_penalty_for_N = parse_int(0, "--bowtie2-penalty-for-N must be at least 0"); break;
        case O_B2_RDG_OPEN: o.p.bowtie2_read_gap_open = parse_int(0, "--bowtie2-read-gap-open must be at least 0"); break;
        case O_B2_RDG_CONT: o.p.bowtie2_read_gap_cont = parse_int(0, "--bowtie2-read-gap-cont must be at least 0"); break;
        case O_B2_RFG_OPEN: o.p.bowtie2_ref_gap_open = parse_int(0, "--bowtie2-ref-gap-open must be at least 0"); break;
        case O_B2_RFG_CONT: o.p.bowtie2_ref_gap_cont = parse_int(0, "--bowtie2-ref-gap-cont must be at least 0"); break;
        case '?': case ':': usage(); return 1;
        default: break;      // accepted, irrelevant to this path
        }
    }
    return 0;
}

inline std::vector<std::string> split(const std::string& s, char sep) {      // tokenize.cpp
    std::vector<std::string> out;
    size_t i = 0;
    while (i <= s.size()) {
        size_t j = s.find(sep, i);
        if (j == std::string::npos) j = s.size();
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j + 1;
    }
    return out;
}
inline std::string file_ext(const std::string& f) {
    size_t d = f.rfind('.');
    if (d == std::string::npos) return "";
    std::string e = f.substr(d + 1);
    for (auto& c : e) c = (char)tolower(c);
    return e;
}

// Host threading: every input file is inflated / tokenised / parsed by its own reader thread, which hands chunks of
// finished records to the consumer through a small bounded queue; the consumer (merge by read id, batching) never
// parses.  THJ_HOST_THREADS bounds the worker count of the parallel stages (default: min(32, hardware threads)).
// CPUs this process may actually use: the hardware threads, capped by the cgroup's CPU quota (containers often show all of
// a machine's 256 hardware threads and allow 16 CPUs' worth of time; sizing thread pools by the former only adds switches)
inline int effective_cpus() {
    static const int n = [] {
        int hw = (int)std::thread::hardware_concurrency();
        if (hw < 1) hw = 1;
        long long quota = -1, period = 100000;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                       // cgroup v2: "<quota|max> <period>"
            char q[64];
            if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max")) quota = atoll(q);
            fclose(f);
        } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {  // cgroup v1
            if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
            fclose(f1);
            if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &period) != 1) period = 100000; fclose(f2); }
        }
        if (quota > 0 && period > 0) { int c = (int)((quota + period - 1) / period); if (c >= 1 && c < hw) hw = c; }
        return hw;
    }();
    return n;
}
inline int host_threads() {
    int n = getenv("THJ_HOST_THREADS") ? atoi(getenv("THJ_HOST_THREADS")) : 0;
    if (n <= 0) { n = effective_cpus(); if (n > 32) n = 32; }
    return n < 1 ? 1 : n;
}

// ------------------------------------------------------------------ reference table (bwt_map.h:579-788)
struct RefTable {
    std::vector<std::string> names;                 // id-1 -> name, @SQ order then FASTA-only names
    std::unordered_map<std::string, uint32_t> ids;
    std::vector<std::string> seqs;                  // id-1 -> folded sequence ("" = none)
    std::string header_text;                        // the '@' lines of --sam-header, verbatim
    std::vector<std::pair<std::string, uint32_t>> sq;   // @SQ (name, LN) in file order: the BAM header targets

    std::mutex mu;                                  // reader threads resolve names concurrently
    // Once the inputs' headers have been read (register_targets) the table is frozen: the genome goes to the device with
    // exactly these contigs, writers index `names` without a lock, and a record naming a contig no header knows is
    // dropped by the hit factories (id 0) instead of growing the table under the readers' feet.
    bool frozen = false;
    bool warned_unknown = false;
    void freeze() { std::lock_guard<std::mutex> lk(mu); frozen = true; }
    uint32_t get_id(const std::string& name) {
        // consecutive records mostly name the same contig: a per-thread one-entry cache keeps the lock out of the way
        static thread_local RefTable* c_rt = nullptr; static thread_local std::string c_name; static thread_local uint32_t c_id = 0;
        if (c_rt == this && c_name == name) return c_id;
        std::lock_guard<std::mutex> lk(mu);
        uint32_t id = get_id_locked(name);
        c_rt = this; c_name = name; c_id = id;
        return id;
    }
    uint32_t get_id_locked(const std::string& name) {
        auto it = ids.find(name);
        if (it != ids.end()) return it->second;
        if (frozen) {
            if (!warned_unknown) { warned_unknown = true; fprintf(stderr, "Warning: alignment on contig %s, which no header or FASTA record names; such records are skipped\n", name.c_str()); }
            return 0;
        }
        names.push_back(name);
        seqs.emplace_back();
        ids[name] = (uint32_t)names.size();
        return (uint32_t)names.size();
    }
    void load_sam_header(const std::string& fn) {
        if (fn.empty()) return;
        FILE* f = fopen(fn.c_str(), "r");
        if (!f) die("Failed to open SAM header file %s\n", fn.c_str());
        char* line = nullptr; size_t cap = 0; ssize_t n;
        while ((n = getline(&line, &cap, f)) > 0) {
            if (line[0] != '@') break;
            std::string l(line, (size_t)n);
            if (l.back() != '\n') l.push_back('\n');
            header_text += l;
            if (!strncmp(line, "@SQ", 3)) {
                std::string sn; uint32_t ln = 0;
                for (auto& tok : split(l.substr(0, l.size() - 1), '\t')) {
                    if (!tok.compare(0, 3, "SN:")) sn = tok.substr(3);
                    else if (!tok.compare(0, 3, "LN:")) ln = (uint32_t)atoll(tok.c_str() + 3);
                }
                if (!sn.empty()) { get_id(sn); sq.emplace_back(sn, ln); }
            }
        }
        free(line);
        fclose(f);
    }
    // get_seqs (segment_juncs.cpp:64-88): names cut at the first blank; sequences folded to ACGTN.
    // The file is read in one piece; record boundaries are found with memchr, every record is folded by its own
    // share of the worker threads (a 3 Gb genome is a few seconds instead of a char-at-a-time minute).
    void load_fasta(const std::string& fn) {
        FILE* f = fopen(fn.c_str(), "rb");
        if (!f) die("Error: cannot open %s for reading\n", fn.c_str());
        std::vector<char> buf;
        const char* d = nullptr; size_t n = 0;
        struct Unmap { void* p = nullptr; size_t n = 0; ~Unmap() { if (p) munmap(p, n); } } um;
        {
            fseek(f, 0, SEEK_END);
            long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            if (sz > 0) {                             // a file: mapped (the page cache's pages are read where they lie; copying 3 GB of them took a second)
                void* m = mmap(nullptr, (size_t)sz, PROT_READ, MAP_PRIVATE, fileno(f), 0);
                if (m != MAP_FAILED) { um.p = m; um.n = (size_t)sz; d = (const char*)m; n = (size_t)sz; madvise(m, (size_t)sz, MADV_WILLNEED); }
                else {
                    buf.resize((size_t)sz);
                    if (fread(buf.data(), 1, (size_t)sz, f) != (size_t)sz) die("Error: cannot read %s\n", fn.c_str());
                }
            } else {                                  // not seekable (a pipe): read to the end
                char tmp[1 << 16]; size_t k;
                while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + k);
            }
        }
        fclose(f);
        if (!d) { d = buf.data(); n = buf.size(); }
        struct Rec { uint32_t id; size_t b, e; };                 // sequence text [b, e) of one record
        std::vector<Rec> recs;
        size_t i = 0;
        while (i < n) {
            if (d[i] == '>') {
                const char* nl = (const char*)memchr(d + i, '\n', n - i);
                size_t le = nl ? (size_t)(nl - d) : n;
                std::string name(d + i + 1, le - i - 1);
                size_t e = name.find_first_of(" \t\r\n");
                if (e != std::string::npos) name.resize(e);
                uint32_t id = get_id(name);
                if (!recs.empty()) recs.back().e = i;
                recs.push_back({id, le < n ? le + 1 : n, n});
                i = le < n ? le + 1 : n;
            } else {                                              // next header: a '>' at the start of a line
                const char* p = d + i;
                for (;;) {
                    const char* nl = (const char*)memchr(p, '\n', (size_t)(d + n - p));
                    if (!nl || nl + 1 >= d + n) { i = n; break; }
                    if (nl[1] == '>') { i = (size_t)(nl + 1 - d); break; }
                    p = nl + 1;
                }
            }
        }
        static const struct Fold { char t[256]; Fold() { for (int c = 0; c < 256; ++c) t[c] = 'N'; t['a'] = t['A'] = 'A'; t['c'] = t['C'] = 'C';
                                                          t['g'] = t['G'] = 'G'; t['t'] = t['T'] = 'T'; t['\n'] = t['\r'] = t[' '] = t['\t'] = 0; } } fold;
        const int T = host_threads();
        for (auto& r : recs) {
            // pass 1: bases per slice; pass 2: fold into place
            const size_t len = r.e - r.b;
            const int parts = (int)std::min<size_t>((size_t)T, len / (1 << 20) + 1);
            std::vector<size_t> cnt((size_t)parts + 1, 0);
            auto slice = [&](int k, size_t& a, size_t& b) { a = r.b + len * (size_t)k / (size_t)parts; b = r.b + len * (size_t)(k + 1) / (size_t)parts; };
            auto run = [&](const std::function<void(int)>& fn) {
                if (parts == 1) { fn(0); return; }
                std::vector<std::thread> th;
                for (int k = 0; k < parts; ++k) th.emplace_back(fn, k);
                for (auto& x : th) x.join();
            };
            run([&](int k) { size_t a, b, c = 0; slice(k, a, b); for (size_t q = a; q < b; ++q) c += fold.t[(unsigned char)d[q]] != 0; cnt[(size_t)k + 1] = c; });
            for (int k = 0; k < parts; ++k) cnt[(size_t)k + 1] += cnt[(size_t)k];
            std::string& out = seqs[r.id - 1];
            out.assign(cnt[(size_t)parts], 'N');
            run([&](int k) { size_t a, b; slice(k, a, b); char* o = &out[0] + cnt[(size_t)k];
                             for (size_t q = a; q < b; ++q) { char c = fold.t[(unsigned char)d[q]]; if (c) *o++ = c; } });
        }
    }
    // pack (once) + upload (to every GPU's context) through the C ABI
    std::once_flag packed_once;
    std::unique_ptr<uint64_t[]> packed_own; std::vector<uint32_t> packed_blk; std::vector<int64_t> packed_lens; int64_t packed_nb = 0;
    const uint64_t* packed_ptr = nullptr;           // the blocks: packed_own, or the mapped cache file

    // ---- the packed-genome cache.  tophat.py starts three processes on one reference (segment_juncs, long_spanning_reads per side), and
    // each of them parses and packs the FASTA (as the reference's do, segment_juncs.cpp:64-88, long_spanning_reads.cpp:2891): seconds
    // each for a 3 Gb genome, against a budget of seconds for the whole run at north_star's rate.  The first process that packs a
    // genome writes the blocks beside its outputs (`<cache>`: names, lengths, block offsets, the 32-byte blocks); a later process
    // whose FASTA has the same size and modification time, and whose @SQ names are a prefix of the cached name table, maps the file
    // instead -- the page cache serves it -- and never sees the text.  THJ_GENOME_CACHE=0 turns it off, =<dir> puts it elsewhere.
    std::string fasta_path, cache_file;             // cache_file empty: no cache
    void* mapped = nullptr; size_t mapped_bytes = 0;
    bool from_cache = false;
    size_t cached_names = 0;                        // names in the table when the cache was adopted
    std::future<void> cache_writer;
    struct CacheHeader { char magic[8]; uint64_t fasta_size; int64_t fasta_mtime_ns; uint64_t n_names, n_blocks, names_bytes, header_bytes; };
    static constexpr const char* CACHE_MAGIC = "THJ2BIT\2";
    static std::string cache_path_for(const std::string& fasta, const std::string& out_file) {
        const char* e = getenv("THJ_GENOME_CACHE");
        if (e && !strcmp(e, "0")) return "";
        std::string dir = e && *e ? std::string(e) : (out_file.find('/') == std::string::npos ? std::string(".") : out_file.substr(0, out_file.rfind('/')));
        char* rp = realpath(fasta.c_str(), nullptr);
        const std::string key = rp ? rp : fasta;
        free(rp);
        uint64_t h = 1469598103934665603ull;
        for (unsigned char ch : key) { h ^= ch; h *= 1099511628211ull; }
        char buf[64]; snprintf(buf, sizeof buf, "/.thj2bit.%016llx", (unsigned long long)h);
        return dir + buf;
    }
    static bool fasta_stat(const std::string& fn, uint64_t& size, int64_t& mtime_ns) {
        struct stat st;
        if (stat(fn.c_str(), &st) || !S_ISREG(st.st_mode)) return false;
        size = (uint64_t)st.st_size; mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
        return true;
    }
    // the reference: from the cache when it is there and fits, else from the FASTA (and the cache is written once the genome is packed)
    void load_reference(const std::string& fasta, const std::string& out_file) {
        fasta_path = fasta;
        cache_file = cache_path_for(fasta, out_file);
        if (!cache_file.empty() && load_cache()) return;
        load_fasta(fasta);
    }
    bool load_cache() {
        uint64_t fsz; int64_t fmt;
        if (!fasta_stat(fasta_path, fsz, fmt)) return false;
        const int fd = open(cache_file.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        CacheHeader h;
        bool ok = !fstat(fd, &st) && (size_t)st.st_size >= sizeof h && pread(fd, &h, sizeof h, 0) == (ssize_t)sizeof h && !memcmp(h.magic, CACHE_MAGIC, 8) &&
                  h.fasta_size == fsz && h.fasta_mtime_ns == fmt && h.n_names < (1ull << 31) && (uint64_t)st.st_size == h.header_bytes + h.n_blocks * 32;
        void* m = MAP_FAILED;
        if (ok) { m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0); ok = m != MAP_FAILED; }
        close(fd);
        if (!ok) return false;
        const char* base = (const char*)m;
        const char* nm = base + sizeof h;
        // the header's own layout must hold what it says it holds (a torn or foreign file must not send the name scan past the mapping)
        const uint64_t names_pad = (h.names_bytes + 7) & ~7ull;
        if (h.names_bytes > (1ull << 32) || h.header_bytes > (uint64_t)st.st_size ||
            h.header_bytes < sizeof h + names_pad + 8 * h.n_names + 4 * (h.n_names + 1)) { munmap(m, (size_t)st.st_size); return false; }
        const int64_t* lens = (const int64_t*)(nm + names_pad);
        const uint32_t* blk = (const uint32_t*)(lens + h.n_names);
        // the cached table: the names this process knows so far (--sam-header's @SQ lines, in order) must lead it
        std::vector<std::string> cn;
        {
            const char* q = nm; const char* const qe = nm + h.names_bytes;
            bool names_ok = true;
            for (uint64_t i = 0; i < h.n_names && names_ok; ++i) {
                const char* z = (const char*)memchr(q, 0, (size_t)(qe - q));
                if (!z) { names_ok = false; break; }
                cn.emplace_back(q, (size_t)(z - q)); q = z + 1;
            }
            if (!names_ok || blk[h.n_names] > h.n_blocks) { munmap(m, (size_t)st.st_size); return false; }
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            ok = names.size() <= cn.size();
            for (size_t i = 0; ok && i < names.size(); ++i) ok = names[i] == cn[i];
            if (ok) for (size_t i = names.size(); i < cn.size(); ++i) get_id_locked(cn[i]);
        }
        if (!ok) { munmap(m, (size_t)st.st_size); return false; }
        packed_lens.assign(lens, lens + h.n_names);
        packed_blk.assign(blk, blk + h.n_names + 1);
        packed_nb = (int64_t)h.n_blocks;
        mapped = m; mapped_bytes = (size_t)st.st_size;
        packed_ptr = (const uint64_t*)(base + h.header_bytes);
        madvise(m, mapped_bytes, MADV_WILLNEED);
        from_cache = true; cached_names = cn.size();
        if (getenv("THJ_TIMING")) fprintf(stderr, "[timing] reference taken from the packed-genome cache %s\n", cache_file.c_str());
        return true;
    }
    void write_cache() {             // after pack(), on a thread of its own; the file appears under its name only when complete
        uint64_t fsz; int64_t fmt;
        if (cache_file.empty() || from_cache || !fasta_stat(fasta_path, fsz, fmt)) return;
        std::string nm;
        for (auto& n : names) { nm += n; nm.push_back('\0'); }
        CacheHeader h;
        memcpy(h.magic, CACHE_MAGIC, 8);
        h.fasta_size = fsz; h.fasta_mtime_ns = fmt; h.n_names = names.size(); h.n_blocks = (uint64_t)packed_nb; h.names_bytes = nm.size();
        const size_t names_pad = (nm.size() + 7) & ~(size_t)7;
        size_t hb = sizeof h + names_pad + names.size() * 8 + (names.size() + 1) * 4;
        hb = (hb + 4095) & ~(size_t)4095;
        h.header_bytes = hb;
        const std::string tmp = cache_file + ".tmp." + std::to_string((long)getpid());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) return;                          // (a directory we may not write to: no cache)
        if (tmp.size() < 4096) strcpy(pending_tmp_path(), tmp.c_str());
        std::vector<char> head(hb, 0);
        memcpy(head.data(), &h, sizeof h);
        memcpy(head.data() + sizeof h, nm.data(), nm.size());
        memcpy(head.data() + sizeof h + names_pad, packed_lens.data(), names.size() * 8);
        memcpy(head.data() + sizeof h + names_pad + names.size() * 8, packed_blk.data(), (names.size() + 1) * 4);
        auto put = [&](const char* p, size_t n) { while (n) { const ssize_t w = ::write(fd, p, n > (1u << 30) ? (1u << 30) : n); if (w <= 0) return false; p += w; n -= (size_t)w; } return true; };
        const bool ok = put(head.data(), hb) && put((const char*)packed_own.get(), (size_t)packed_nb * 32);
        close(fd);
        if (!ok || rename(tmp.c_str(), cache_file.c_str())) unlink(tmp.c_str());
        pending_tmp_path()[0] = 0;
    }
    void finish_cache() { if (cache_writer.valid()) cache_writer.get(); }      // before the process leaves
    ~RefTable() { finish_cache(); if (mapped) munmap(mapped, mapped_bytes); }
    // a contig's text for the few host-side uses (MD strings longer than a device record holds): decoded from the blocks when the
    // reference came from the cache
    const std::string& text(uint32_t ref_id) {
        std::string& s = seqs[ref_id - 1];
        if (!from_cache || ref_id > packed_lens.size() || packed_lens[ref_id - 1] == 0) return s;
        { std::lock_guard<std::mutex> lk(text_mu); if (!s.empty()) return s; }      // (checked under the lock: another thread may be putting the text in place)
        // decoded outside the lock (a chromosome is a quarter of a second); the first thread done puts its copy in place
        const int64_t n = packed_lens[ref_id - 1];
        std::string t((size_t)n, 'N');
        const uint64_t* b = packed_ptr + (size_t)packed_blk[ref_id - 1] * 4;
        for (int64_t i = 0; i < n; ++i) {
            const uint64_t* w = b + (size_t)(i >> 6) * 4; const int k = (int)(i & 63);
            if (!((w[2] >> k) & 1ull)) t[(size_t)i] = "ACGT"[((w[0] >> k) & 1ull) | (((w[1] >> k) & 1ull) << 1)];
        }
        std::lock_guard<std::mutex> lk(text_mu);
        if (s.empty()) s.swap(t);
        return s;
    }
    std::mutex text_mu;
    void pack() {
        std::call_once(packed_once, [this] {
            {
                std::lock_guard<std::mutex> lk(mu);
                frozen = true;                           // the device genome has exactly the contigs known now
            }
            if (from_cache && names.size() != cached_names) {
                // a map's header named a contig the cached table does not have: the cache does not describe this run.  Parse the FASTA after all
                from_cache = false; packed_ptr = nullptr;
                { std::lock_guard<std::mutex> lk(mu); frozen = false; }
                load_fasta(fasta_path);
                { std::lock_guard<std::mutex> lk(mu); frozen = true; }
            }
            if (from_cache) return;
            std::lock_guard<std::mutex> lk(mu);
            int32_t n = (int32_t)names.size();
            packed_lens.resize(n); std::vector<const char*> ptrs(n);
            for (int32_t i = 0; i < n; ++i) { packed_lens[i] = (int64_t)seqs[i].size(); ptrs[i] = seqs[i].empty() ? nullptr : seqs[i].data(); }
            packed_blk.resize(n + 1);
            if (thj_genome_layout(n, packed_lens.data(), packed_blk.data(), &packed_nb)) die("Error: %s\n", thj_last_error());
            packed_own.reset(new uint64_t[(size_t)packed_nb * 4]);           // (uninitialised: thj_genome_pack writes every block)
            if (thj_genome_pack(n, ptrs.data(), packed_lens.data(), packed_blk.data(), packed_own.get(), packed_nb)) die("Error: %s\n", thj_last_error());
            packed_ptr = packed_own.get();
            if (!cache_file.empty()) cache_writer = std::async(std::launch::async, [this] { write_cache(); });
        });
    }
    void upload(thj_ctx* ctx) {
        pack();
        if (thj_genome_upload(ctx, packed_ptr, packed_nb, packed_blk.data(), packed_lens.data(), (int32_t)packed_lens.size())) die("Error: %s\n", thj_last_error());
    }
};

// ------------------------------------------------------------------ BAM / SAM record reader
struct AlnRec {
    std::string qname, rname, rnext;     // names ("*" when unset)
    int32_t pos = -1;                    // 0-based
    uint32_t flag = 0;
    std::vector<std::pair<char, uint32_t>> cigar;
    std::string seq, qual;               // qual as phred+33 text
    int nm = 0; bool has_nm = false;
    char xs = 0;
    bool has_xf = false;
    std::string md; bool has_md = false;
};

class AlnReader {
    gzFile gz_ = nullptr;       // BGZF is a multi-member gzip stream: zlib walks the members for us
    FILE* txt_ = nullptr;
    bool bam_ = false;
    std::vector<std::string> targets_;
    char* line_ = nullptr; size_t cap_ = 0;
    std::vector<uint8_t> buf_;
    void rd(void* dst, int n, const char* fn) { if (gzread(gz_, dst, (unsigned)n) != n) die("Error: truncated BAM file %s\n", fn); }
public:
    std::string fname;
    bool want_seq = true;       // false: SEQ / QUAL are not decoded, r.seq only gets its length (the hit factories' need)
    const std::vector<std::string>& targets() const { return targets_; }
    // offset: where to start reading records -- a BGZF virtual offset (compressed block address << 16 | offset inside the
    // inflated block: what GBamWriter's .index holds and bgzf_seek takes, common.h:277-283) or a byte offset for SAM text
    bool open(const std::string& fn, int64_t offset = 0) {
        fname = fn;
        if (file_ext(fn) == "sam") {                 // bwt_map.cpp:170-175
            txt_ = fopen(fn.c_str(), "r");
            if (txt_ && offset > 0) fseek(txt_, (long)offset, SEEK_SET);
            return txt_ != nullptr;
        }
        bam_ = true;
        gz_ = gzopen(fn.c_str(), "rb");
        if (!gz_) return false;
        gzbuffer(gz_, 1 << 20);
        char magic[4];
        if (gzread(gz_, magic, 4) != 4 || memcmp(magic, "BAM\1", 4)) die("Error: %s is not a BAM file\n", fn.c_str());
        int32_t l_text, n_ref;
        rd(&l_text, 4, fn.c_str());
        std::vector<char> t((size_t)l_text + 1);
        if (l_text) rd(t.data(), l_text, fn.c_str());
        rd(&n_ref, 4, fn.c_str());
        for (int i = 0; i < n_ref; ++i) {
            int32_t l_name, l_ref;
            rd(&l_name, 4, fn.c_str());
            std::vector<char> nm((size_t)l_name + 1);
            rd(nm.data(), l_name, fn.c_str());
            rd(&l_ref, 4, fn.c_str());
            targets_.emplace_back(nm.data());
        }
        if (offset > 0) {
            // a BGZF block is a complete gzip member: start zlib at the block's file address, then skip inside it
            gzclose(gz_);
            int fd = ::open(fn.c_str(), O_RDONLY);
            if (fd < 0 || lseek(fd, (off_t)(offset >> 16), SEEK_SET) < 0) die("Error: cannot seek in %s\n", fn.c_str());
            gz_ = gzdopen(fd, "rb");
            if (!gz_) die("Error: cannot reopen %s\n", fn.c_str());
            gzbuffer(gz_, 1 << 20);
            char skip[65536];
            int within = (int)(offset & 0xFFFF);
            if (within && gzread(gz_, skip, (unsigned)within) != within) die("Error: bad index offset for %s\n", fn.c_str());
        }
        return true;
    }
    void close() { if (gz_) gzclose(gz_); if (txt_) fclose(txt_); gz_ = nullptr; txt_ = nullptr; free(line_); line_ = nullptr; }
    ~AlnReader() { close(); }

    bool is_bam() const { return bam_; }
    // the next BAM record as raw bytes (after its block_size field); nullptr at the end of the file
    const uint8_t* next_raw(int32_t& bs) {
        int got = gzread(gz_, &bs, 4);
        if (got != 4) return nullptr;
        buf_.resize((size_t)bs);
        rd(buf_.data(), bs, fname.c_str());
        return buf_.data();
    }
    bool next(AlnRec& r) {
        // reset in place: the strings keep their capacity from record to record
        r.qname.clear(); r.rname.clear(); r.rnext.clear(); r.pos = -1; r.flag = 0; r.cigar.clear(); r.seq.clear(); r.qual.clear();
        r.nm = 0; r.has_nm = false; r.xs = 0; r.has_xf = false; r.md.clear(); r.has_md = false;
        if (bam_) {
            int32_t bs;
            int got = gzread(gz_, &bs, 4);
            if (got != 4) return false;
            buf_.resize((size_t)bs);
            rd(buf_.data(), bs, fname.c_str());
            const uint8_t* d = buf_.data();
            int32_t tid, pos, mtid; uint32_t bin_mq_nl, flag_nc; int32_t l_seq;
            memcpy(&tid, d, 4); memcpy(&pos, d + 4, 4); memcpy(&bin_mq_nl, d + 8, 4); memcpy(&flag_nc, d + 12, 4);
            memcpy(&l_seq, d + 16, 4); memcpy(&mtid, d + 20, 4);
            uint32_t l_rn = bin_mq_nl & 0xFF, n_cig = flag_nc & 0xFFFF;
            r.flag = flag_nc >> 16; r.pos = pos;
            r.rname = tid >= 0 && tid < (int)targets_.size() ? targets_[tid] : "*";
            r.rnext = mtid < 0 ? "*" : (mtid == tid ? "=" : (mtid < (int)targets_.size() ? targets_[mtid] : "*"));
            size_t p = 32;
            r.qname.assign((const char*)d + p, l_rn ? l_rn - 1 : 0); p += l_rn;
            static const char CIG[] = "MIDNSHP=X";
            for (uint32_t i = 0; i < n_cig; ++i) { uint32_t c; memcpy(&c, d + p, 4); p += 4; r.cigar.emplace_back(CIG[c & 0xF], c >> 4); }
            static const char SEQ[] = "=ACMGRSVTWYHKDBN";
            r.seq.resize((size_t)l_seq);
            if (want_seq) {
                r.qual.resize((size_t)l_seq);
                for (int i = 0; i < l_seq; ++i) r.seq[i] = SEQ[(d[p + (i >> 1)] >> ((i & 1) ? 0 : 4)) & 0xF];
                for (int i = 0; i < l_seq; ++i) r.qual[i] = (char)(d[p + (size_t)(l_seq + 1) / 2 + i] + 33);
            }
            p += (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
            while (p + 3 <= (size_t)bs) {                       // bam_aux_get for NM / XS / XF
                char t0 = (char)d[p], t1 = (char)d[p + 1], ty = (char)d[p + 2];
                p += 3;
                long long iv = 0; bool isint = false;
                switch (ty) {
                case 'A': if (t0 == 'X' && t1 == 'S') r.xs = (char)d[p]; p += 1; break;
                case 'c': iv = (int8_t)d[p]; isint = true; p += 1; break;
                case 'C': iv = d[p]; isint = true; p += 1; break;
                case 's': { int16_t v; memcpy(&v, d + p, 2); iv = v; isint = true; p += 2; break; }
                case 'S': { uint16_t v; memcpy(&v, d + p, 2); iv = v; isint = true; p += 2; break; }
                case 'i': { int32_t v; memcpy(&v, d + p, 4); iv = v; isint = true; p += 4; break; }
                case 'I': { uint32_t v; memcpy(&v, d + p, 4); iv = v; isint = true; p += 4; break; }
                case 'f': p += 4; break;
                case 'd': p += 8; break;
                case 'Z': case 'H': { if (t0 == 'X' && t1 == 'F') r.has_xf = true; size_t z0 = p; while (p < (size_t)bs && d[p]) ++p;
                            if (t0 == 'M' && t1 == 'D') { r.md.assign((const char*)d + z0, p - z0); r.has_md = true; } ++p; break; }
                case 'B': { char st = (char)d[p]; int32_t cnt; memcpy(&cnt, d + p + 1, 4); int sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                            p += 5 + (size_t)cnt * sz; break; }
                default: p = (size_t)bs; break;
                }
                if (isint && t0 == 'N' && t1 == 'M') { r.nm = (int)iv; r.has_nm = true; }
            }
            return true;
        }
        ssize_t n;
        while ((n = getline(&line_, &cap_, txt_)) > 0) {
            if (line_[0] == '@') continue;
            while (n > 0 && (line_[n - 1] == '\n' || line_[n - 1] == '\r')) --n;
            if (n == 0) continue;
            line_[n] = 0;
            // fields in place: f[k] .. f[k] + flen[k]
            const char* f[64]; size_t flen[64]; int nf = 0;
            {
                const char* s0 = line_; const char* end = line_ + n;
                while (nf < 64) {
                    const char* t = (const char*)memchr(s0, '\t', (size_t)(end - s0));
                    f[nf] = s0; flen[nf] = (size_t)((t ? t : end) - s0); ++nf;
                    if (!t) break;
                    s0 = t + 1;
                }
            }
            if (nf < 11) die("Error: malformed SAM line in %s\n", fname.c_str());
            r.qname.assign(f[0], flen[0]); r.flag = (uint32_t)atoi(f[1]); r.rname.assign(f[2], flen[2]); r.pos = atoi(f[3]) - 1;
            r.rnext.assign(f[6], flen[6]);
            if (want_seq) { r.seq.assign(f[9], flen[9]); r.qual.assign(f[10], flen[10]); }
            else r.seq.resize(flen[9]);
            if (!(flen[5] == 1 && f[5][0] == '*')) {
                const char* s = f[5]; const char* e5 = f[5] + flen[5];
                while (s < e5) { char* e; long len = strtol(s, &e, 10); if (e >= e5) break; r.cigar.emplace_back(*e, (uint32_t)len); s = e + 1; }
            }
            for (int k = 11; k < nf; ++k) {
                if (flen[k] < 5) continue;
                if (!memcmp(f[k], "NM:i:", 5)) { r.nm = atoi(f[k] + 5); r.has_nm = true; }
                else if (!memcmp(f[k], "XS:A:", 5)) r.xs = flen[k] > 5 ? f[k][5] : 0;
                else if (!memcmp(f[k], "XF:Z:", 5)) r.has_xf = true;
                else if (!memcmp(f[k], "MD:Z:", 5)) { r.md.assign(f[k] + 5, flen[k] - 5); r.has_md = true; }
            }
            return true;
        }
        return false;
    }
};

// ------------------------------------------------------------------ BAMHitFactory::get_hit_from_buf (bwt_map.cpp:1101-1452)
struct Hit {
    uint32_t insert_id = 0;
    thj_hit h16{};           // segment_juncs form
    thj_span_hit h32{};      // long_spanning_reads form
};

// returns false when the reference's factory returns false (record dropped) or the record is unmapped
inline bool parse_hit(const AlnRec& r, RefTable& rt, const thj_params& p, Hit& out) {
    bool end = true;
    const std::string& q = r.qname;
    size_t pipe = q.rfind('|');
    if (pipe != std::string::npos) {
        const char* tag = q.c_str() + pipe + 1;
        if (strchr(tag, ':')) {                       // "<offset>:<segment>:<segments>" (tophat.py:2948)
            char* e;
            unsigned long b = 0, c = 0;
            strtoul(tag, &e, 10);
            if (*e == ':') { b = strtoul(e + 1, &e, 10); if (*e == ':') c = strtoul(e + 1, &e, 10); }
            end = (b + 1 == c);
        }
    }
    out.insert_id = (uint32_t)atoi(q.c_str());          // stops at the '|'

    if (r.rname == "*" || (r.flag & 4)) return false;    // unmapped: the maps this path is fed hold mapped records only
    if (r.has_xf) die("Error: fusion (XF) alignments in %s are not supported by this build\n", r.qname.c_str());
    unsigned char mism = (unsigned char)r.nm;
    int right = r.pos, read_len = 0, gap = 0;
    bool spliced = false;
    int n32 = 0;
    for (auto& c : r.cigar) {
        uint32_t len = c.second;
        if (len == 0) return false;                                            // :1322-1326
        uint32_t op;
        switch (c.first) {
        case 'M': case '=': case 'X': op = THJ_CIG_MATCH; right += (int)len; read_len += (int)len; break;
        case 'I': op = THJ_CIG_INS; read_len += (int)len; gap += (int)len; mism = (unsigned char)(mism - len); break;
        case 'D': op = THJ_CIG_DEL; right += (int)len; gap += (int)len; mism = (unsigned char)(mism - len); break;
        case 'S': op = THJ_CIG_SOFT_CLIP; read_len += (int)len; break;
        case 'H': continue;
        case 'P': op = 15; break;
        case 'N': op = THJ_CIG_REF_SKIP; spliced = true; if ((int)len > p.max_report_intron) return false; right += (int)len; break;   // :1337-1345
        default: return false;
        }
        if (n32 < 5) out.h32.cigar[n32] = (op << 28) | (len & 0x0FFFFFFFu);
        ++n32;
    }
    if (r.rnext != "*" && r.rnext != "=" && r.rnext != r.rname) return false;  // :1409-1415
    uint32_t ref_id = rt.get_id(r.rname);
    if (ref_id == 0) return false;                                             // no header names this contig (frozen table)
    bool anti = (r.flag & 0x10) != 0;
    unsigned char ed = (unsigned char)(mism + gap);
    out.h16.ref_id = ref_id; out.h16.left = r.pos; out.h16.right = right;
    out.h16.flags = (uint8_t)((anti ? THJ_HIT_ANTISENSE : 0) | (end ? THJ_HIT_END : 0));
    out.h16.edit_dist = ed; out.h16.mismatches = mism; out.h16.read_len = (uint8_t)(read_len > 255 ? 255 : read_len);
    if (n32 > 5) die("Error: segment alignment %s has %d CIGAR operations (this build supports 5)\n", r.qname.c_str(), n32);
    out.h32.ref_id = ref_id; out.h32.left = r.pos;
    out.h32.flags = (uint8_t)(out.h16.flags | ((spliced && r.xs == '-') ? THJ_HIT_ANTISENSE_SPLICE : 0));
    out.h32.mismatches = mism; out.h32.edit_dist = ed; out.h32.n_cigar = (uint8_t)n32;
    return true;
}

// The same factory straight from a BAM record's bytes (no intermediate strings; `tid2ref` = the file's targets resolved to
// reference-table ids once, 0 = unknown contig): what the segment / read maps of a real run go through, record by record.
// A BAM record's own header must describe something that fits its block_size (name, CIGAR, bases, qualities; the name NUL-terminated
// inside it) before anybody walks it: a truncated or damaged file is an error, not a read past the buffer.
inline bool bam_record_shape_ok(const uint8_t* d, int32_t bs) {
    if (bs < 32) return false;
    uint32_t bin_mq_nl, flag_nc; int32_t l_seq;
    memcpy(&bin_mq_nl, d + 8, 4); memcpy(&flag_nc, d + 12, 4); memcpy(&l_seq, d + 16, 4);
    const uint64_t l_rn = bin_mq_nl & 0xFF, n_cig = flag_nc & 0xFFFF;
    if (l_seq < 0 || l_rn == 0) return false;
    if (32 + l_rn + 4 * n_cig + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq > (uint64_t)bs) return false;
    return d[32 + l_rn - 1] == 0;
}
inline size_t bam_aux_fixed_size(char ty) {                   // bytes a tag of this type has at least after its three-byte key
    switch (ty) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; case 'd': return 8; case 'B': return 5; default: return 0; }
}
inline bool parse_hit_bam(const uint8_t* d, int32_t bs, const std::vector<uint32_t>& tid2ref, const thj_params& p, Hit& out) {
    if (!bam_record_shape_ok(d, bs)) die("Error: malformed BAM record (its header does not fit its %d bytes)\n", (int)bs);
    int32_t tid, pos, mtid; uint32_t bin_mq_nl, flag_nc; int32_t l_seq;
    memcpy(&tid, d, 4); memcpy(&pos, d + 4, 4); memcpy(&bin_mq_nl, d + 8, 4); memcpy(&flag_nc, d + 12, 4);
    memcpy(&l_seq, d + 16, 4); memcpy(&mtid, d + 20, 4);
    const uint32_t l_rn = bin_mq_nl & 0xFF, n_cig = flag_nc & 0xFFFF, flag = flag_nc >> 16;
    const char* q = (const char*)d + 32;                       // NUL-terminated
    bool end = true;
    const char* pipe = strrchr(q, '|');
    if (pipe && strchr(pipe + 1, ':')) {
        char* e;
        unsigned long b = 0, c = 0;
        strtoul(pipe + 1, &e, 10);
        if (*e == ':') { b = strtoul(e + 1, &e, 10); if (*e == ':') c = strtoul(e + 1, &e, 10); }
        end = (b + 1 == c);
    }
    out.insert_id = (uint32_t)atoi(q);
    if (tid < 0 || (flag & 4)) return false;
    size_t pp = 32 + l_rn;
    int right = pos, read_len = 0, gap = 0, n32 = 0, ind = 0;
    bool spliced = false;
    for (uint32_t i = 0; i < n_cig; ++i) {
        uint32_t c; memcpy(&c, d + pp, 4); pp += 4;
        const uint32_t len = c >> 4, bop = c & 0xF;
        if (len == 0) return false;
        uint32_t op;
        switch (bop) {                                         // "MIDNSHP=X"
        case 0: case 7: case 8: op = THJ_CIG_MATCH; right += (int)len; read_len += (int)len; break;
        case 1: op = THJ_CIG_INS; read_len += (int)len; gap += (int)len; ind += (int)len; break;
        case 2: op = THJ_CIG_DEL; right += (int)len; gap += (int)len; ind += (int)len; break;
        case 4: op = THJ_CIG_SOFT_CLIP; read_len += (int)len; break;
        case 5: continue;
        case 6: op = 15; break;
        case 3: op = THJ_CIG_REF_SKIP; spliced = true; if ((int)len > p.max_report_intron) return false; right += (int)len; break;
        default: return false;
        }
        if (n32 < 5) out.h32.cigar[n32] = (op << 28) | (len & 0x0FFFFFFFu);
        ++n32;
    }
    if (mtid >= 0 && mtid != tid) return false;                // the mate maps to another contig (:1409-1415)
    pp += (size_t)(l_seq + 1) / 2 + (size_t)l_seq;
    int nm = 0; char xs = 0;
    while (pp + 3 <= (size_t)bs) {                             // bam_aux_get for NM / XS / XF
        const char t0 = (char)d[pp], t1 = (char)d[pp + 1], ty = (char)d[pp + 2];
        pp += 3;
        if (bam_aux_fixed_size(ty) > (size_t)bs - pp) die("Error: malformed BAM record (a tag runs past its end)\n");
        long long iv = 0; bool isint = false;
        switch (ty) {
        case 'A': if (t0 == 'X' && t1 == 'S') xs = (char)d[pp]; pp += 1; break;
        case 'c': iv = (int8_t)d[pp]; isint = true; pp += 1; break;
        case 'C': iv = d[pp]; isint = true; pp += 1; break;
        case 's': { int16_t v; memcpy(&v, d + pp, 2); iv = v; isint = true; pp += 2; break; }
        case 'S': { uint16_t v; memcpy(&v, d + pp, 2); iv = v; isint = true; pp += 2; break; }
        case 'i': { int32_t v; memcpy(&v, d + pp, 4); iv = v; isint = true; pp += 4; break; }
        case 'I': { uint32_t v; memcpy(&v, d + pp, 4); iv = v; isint = true; pp += 4; break; }
        case 'f': pp += 4; break;
        case 'd': pp += 8; break;
        case 'Z': case 'H':
            if (t0 == 'X' && t1 == 'F') die("Error: fusion (XF) alignments in %s are not supported by this build\n", q);
            while (pp < (size_t)bs && d[pp]) ++pp;
            ++pp;
            break;
        case 'B': { char st = (char)d[pp]; int32_t cnt; memcpy(&cnt, d + pp + 1, 4); int sz = (st == 'c' || st == 'C') ? 1 : (st == 's' || st == 'S') ? 2 : 4;
                    if (cnt < 0 || (uint64_t)cnt * (uint64_t)sz > (uint64_t)bs) die("Error: malformed BAM record (an array tag runs past its end)\n");
                    pp += 5 + (size_t)cnt * sz; break; }
        default: pp = (size_t)bs; break;
        }
        if (isint && t0 == 'N' && t1 == 'M') nm = (int)iv;
    }
    const uint32_t ref_id = (size_t)tid < tid2ref.size() ? tid2ref[(size_t)tid] : 0;
    if (ref_id == 0) return false;
    const unsigned char mism = (unsigned char)((unsigned char)nm - (unsigned char)ind);
    const bool anti = (flag & 0x10) != 0;
    const unsigned char ed = (unsigned char)(mism + gap);
    out.h16.ref_id = ref_id; out.h16.left = pos; out.h16.right = right;
    out.h16.flags = (uint8_t)((anti ? THJ_HIT_ANTISENSE : 0) | (end ? THJ_HIT_END : 0));
    out.h16.edit_dist = ed; out.h16.mismatches = mism; out.h16.read_len = (uint8_t)(read_len > 255 ? 255 : read_len);
    if (n32 > 5) die("Error: segment alignment %s has %d CIGAR operations (this build supports 5)\n", q, n32);
    out.h32.ref_id = ref_id; out.h32.left = pos;
    out.h32.flags = (uint8_t)(out.h16.flags | ((spliced && xs == '-') ? THJ_HIT_ANTISENSE_SPLICE : 0));
    out.h32.mismatches = mism; out.h32.edit_dist = ed; out.h32.n_cigar = (uint8_t)n32;
    return true;
}

// ---- SplicedBAMHitFactory::get_hit_from_buf + spliceCigar + getBAMmismatches (bwt_map.cpp:1469-1770, :681-883,
// :410-475): a segment mapped against a junction-db contig `name|left|l-r|right|type|strand` (juncs_db.cpp:99,:143)
// becomes a genomic hit with the REF_SKIP / DEL / INS operation spliced into its CIGAR.
typedef std::vector<std::pair<int, int>> CigVec;      // (CigarOpCode, length)
inline void cigar_add(CigVec& c, std::pair<int, int> op) {          // bwt_map.cpp:672-678, quirk included:
    if (op.second <= 0) return;                                      // an op equal to the previous one extends it
    if (!c.empty() && c.back().first == op.first) c.back().second += op.second;   // AND is appended again
    c.push_back(op);
}
inline bool splice_cigar(CigVec& out, const CigVec& cigar, const std::vector<bool>& mism, int& left, int spl_start, int spl_len,
                         int spl_code, int& spl_mm, int min_anchor_len) {
    const int INS = 3, DEL = 5, REF_SKIP = 11, MATCH = 1, PAD = 15, SOFT = 13;
    // fusion ops (7 ff, 8 fr, 9 rf, 10 rr): the piece before an rf / rr break and the piece after an fr / rr break run down the
    // genome and come out as the lower-case ops (MATCH -> mATCH ..., bwt_map.cpp:753-790, :822-853)
    const bool fus = spl_code >= 7 && spl_code <= 10;
    const bool low_before = spl_code == 9 || spl_code == 10, low_after = spl_code == 8 || spl_code == 10;
    auto lower = [](std::pair<int, int> op) { if (op.first == 1 || op.first == 3 || op.first == 5 || op.first == 11) ++op.first; return op; };
    int spl_ofs = spl_start - left;
    if (fus) spl_ofs = abs(spl_ofs);
    int spl_ofs_end = spl_ofs;
    std::pair<int, int> gapop(spl_code, spl_len);
    if (spl_code == INS) spl_ofs_end += spl_len;
    int ref_ofs = 0, read_ofs = 0;
    bool xfound = false;
    if (spl_ofs_end > 0) {
        for (size_t c = 0; c < cigar.size(); ++c) {
            int prev_read_ofs = read_ofs, cur_op_ofs = ref_ofs;
            int cur_opcode = cigar[c].first, cur_oplen = cigar[c].second;
            if (cur_opcode == MATCH) {
                ref_ofs += cur_oplen; read_ofs += cur_oplen;
                for (int o = cur_op_ofs; o < ref_ofs; ++o) {
                    int rofs = prev_read_ofs + (o - cur_op_ofs);
                    bool mmb = rofs >= 0 && rofs < (int)mism.size() && mism[(size_t)rofs];
                    if (spl_code == INS) { if (o >= spl_ofs && o < spl_ofs_end && mmb) ++spl_mm; }
                    else if (abs(spl_ofs - o) < min_anchor_len && mmb) ++spl_mm;
                }
            } else if (cur_opcode == DEL || cur_opcode == REF_SKIP || cur_opcode == PAD) ref_ofs += cur_oplen;
            else if (cur_opcode == SOFT || cur_opcode == INS) read_ofs += cur_oplen;
            if (cur_op_ofs >= spl_ofs_end || ref_ofs <= spl_ofs) {
                if (cur_op_ofs == spl_ofs_end && spl_code != INS && cur_opcode != INS) { xfound = true; cigar_add(out, gapop); }
                cigar_add(out, ((xfound && low_after) || (!xfound && low_before)) ? lower(cigar[c]) : cigar[c]);
            } else {
                xfound = true;
                if (spl_code == INS) {
                    std::pair<int, int> op = cigar[c];
                    op.second = spl_ofs - cur_op_ofs;
                    if (spl_ofs > cur_op_ofs) cigar_add(out, op);
                    if (spl_ofs < 0) { std::pair<int, int> t = gapop; t.second += spl_ofs; if (t.second > 0) cigar_add(out, t); }
                    else cigar_add(out, gapop);
                    op.second = ref_ofs - spl_ofs_end;
                    if (ref_ofs > spl_ofs_end) cigar_add(out, op);
                } else {
                    std::pair<int, int> op = cigar[c];
                    op.second = spl_ofs - cur_op_ofs;
                    cigar_add(out, low_before ? lower(op) : op);
                    cigar_add(out, gapop);
                    op.second = ref_ofs - spl_ofs;
                    cigar_add(out, low_after ? lower(op) : op);
                }
            }
        }
    }
    (void)xfound;
    if (spl_ofs_end <= 0) {           // alignment starts after the splice event
        if (spl_code == INS) left -= spl_len; else left += spl_len;
        out = cigar;
    }
    if (out.size() < cigar.size() + 2) return false;
    if ((out.front().first != MATCH && out.front().first != 2) || (out.back().first != MATCH && out.back().first != 2)) return false;
    return true;
}

// from_bam: SplicedBAMHitFactory understands the fusion contigs of the junction database (strand ff / fr / rf / rr), the SAM
// factory drops every record whose strand field is not fwd / rev (bwt_map.cpp:972-975)
inline bool parse_spliced_hit(const AlnRec& r, RefTable& rt, const thj_params& p, Hit& out, bool from_bam = true) {
    bool end = true;
    std::string q = r.qname;
    size_t pipe = q.rfind('|');
    if (pipe != std::string::npos) {
        const char* tag = q.c_str() + pipe + 1;
        if (strchr(tag, ':')) { unsigned a = 0, b = 0, c = 0; sscanf(tag, "%u:%u:%u", &a, &b, &c); end = (b + 1 == c); }
        q.resize(pipe);
    }
    out.insert_id = (uint32_t)atoi(q.c_str());
    if (r.rname == "*" || (r.flag & 4)) return false;
    CigVec samcigar;
    for (auto& c : r.cigar) {
        if (c.second == 0) return false;
        int op;
        switch (c.first) {
        case 'M': op = 1; break; case 'I': op = 3; break; case 'D': op = 5; break; case 'S': op = 13; break;
        case 'H': continue; case 'P': op = 15; break;
        case 'N': op = 11; if ((int)c.second > p.max_report_intron) return false; break;
        default: return false;
        }
        samcigar.emplace_back(op, (int)c.second);
    }
    if (r.rnext != "*" && r.rnext != "=" && r.rnext != r.rname) return false;
    // getBAMmismatches: mismatch positions from MD
    std::vector<bool> mism(r.seq.size(), false);
    int num_mm = 0;
    if (r.has_md) {
        const char* s = r.md.c_str();
        size_t bi = 0;
        while (*s) {
            if (isdigit((unsigned char)*s)) { bi += (size_t)atoi(s); while (isdigit((unsigned char)*s)) ++s; }
            while (isalpha((unsigned char)*s)) { ++s; ++num_mm; if (bi < mism.size()) mism[bi] = true; ++bi; }
            if (*s == '^') { ++s; while (isalpha((unsigned char)*s)) { ++s; ++bi; } }
            if (*s && !isdigit((unsigned char)*s) && !isalpha((unsigned char)*s) && *s != '^') ++s;
        }
    }
    // tokenize_strict(text_name, "|")
    std::vector<std::string> toks;
    {
        const std::string& s = r.rname;
        size_t last = s.find_first_not_of('|', 0), pos = s.find_first_of('|', last);
        while (last < s.size() || pos < s.size()) {
            toks.push_back(s.substr(last, pos - last));
            if (pos == std::string::npos) break;
            last = pos + 1; pos = s.find_first_of('|', last);
        }
    }
    int ne = (int)toks.size() - 6;
    if (ne < 0) { fprintf(stderr, "Warning: found malformed splice record, skipping\n"); return false; }
    std::string contig = toks[0];
    for (int t = 1; t <= ne; ++t) contig += "|" + toks[(size_t)t];
    std::vector<std::string> st = split(toks[(size_t)ne + 2], '-');
    if (st.size() != 2) { fprintf(stderr, "Warning: found malformed splice record, skipping:\n"); return false; }
    const std::string& jtype = toks[(size_t)ne + 4];
    const std::string& jstrand = toks[(size_t)ne + 5];
    if (!from_bam && jstrand != "rev" && jstrand != "fwd") { fprintf(stderr, "Malformed insertion record\n"); return false; }
    int left = atoi(toks[(size_t)ne + 1].c_str()) + r.pos;
    int lsp = atoi(st[0].c_str());
    CigVec spl;
    int spl_mm = 0;
    bool anti = (r.flag & 0x10) != 0, flipped = false;
    uint32_t ref_id2 = 0;
    if (jtype == "ins") {
        if (left > lsp) return false;
        if (!splice_cigar(spl, samcigar, mism, left, lsp + 1, (int)st[1].size(), 3, spl_mm, p.min_anchor_len)) return false;
        if (spl_mm < 0) return false;
        num_mm -= spl_mm;
    } else {
        if (!(jstrand == "ff" || jstrand == "fr" || jstrand == "rf" || jstrand == "rr" || jstrand == "rev" || jstrand == "fwd")) {
            fprintf(stderr, "Warning: found malformed splice record, skipping\n"); return false;
        }
        const bool fus = jtype == "fus";
        // :1672-1677: on rf / rr fusion contigs the first piece runs down the genome from the contig's left edge
        if (fus && (jstrand == "rf" || jstrand == "rr")) left = atoi(toks[(size_t)ne + 1].c_str()) - r.pos;
        int opcode = jtype == "del" ? 5 : 11;
        if (fus) opcode = jstrand == "ff" ? 7 : (jstrand == "fr" ? 8 : (jstrand == "rf" ? 9 : 10));
        int gap_len = fus ? atoi(st[1].c_str()) : atoi(st[1].c_str()) - lsp - 1;
        if (opcode == 9 || opcode == 10) { lsp -= 1; if (left <= lsp) return false; }
        else { lsp += 1; if (left >= lsp) return false; }
        if (!splice_cigar(spl, samcigar, mism, left, lsp, gap_len, opcode, spl_mm, p.min_anchor_len)) return false;
        if (fus) {
            std::vector<std::string> cs = split(contig, '-');
            if (cs.size() != 2) return false;
            contig = cs[0];
            ref_id2 = rt.get_id(cs[1]);
            if (ref_id2 == 0) return false;
            if (jstrand == "rf" || jstrand == "rr") { anti = !anti; flipped = true; }
        }
    }
    if (spl.size() > (ref_id2 ? 4u : 5u))
        die("Error: spliced segment alignment %s has %d CIGAR operations (this build supports 5, 4 on a fusion contig)\n", r.qname.c_str(), (int)spl.size());
    int gap = 0, right = left, read_len = 0;
    for (int k = 0; k < 5; ++k) out.h32.cigar[k] = 0;
    for (size_t k = 0; k < spl.size(); ++k) {
        int op = spl[k].first, len = spl[k].second;
        if (op >= 3 && op <= 6) gap += len;
        if (op == 1 || op == 5 || op == 11) right += len;
        else if (op == 2 || op == 6 || op == 12) right -= len;
        else if (op >= 7 && op <= 10) right = len;
        if (op == 1 || op == 2 || op == 3 || op == 4 || op == 13) read_len += len;
        out.h32.cigar[k] = ((uint32_t)op << 28) | ((uint32_t)len & 0x0FFFFFFFu);
    }
    if (ref_id2) out.h32.cigar[4] = ref_id2;          // a fused hit keeps its second contig in the last cigar slot (include/thj.h)
    uint32_t ref_id = rt.get_id(contig);
    if (ref_id == 0) return false;
    unsigned char mm8 = (unsigned char)num_mm, ed = (unsigned char)(num_mm + gap);
    out.h16.ref_id = ref_id; out.h16.left = left; out.h16.right = right;
    out.h16.flags = (uint8_t)((anti ? THJ_HIT_ANTISENSE : 0) | (end ? THJ_HIT_END : 0));
    out.h16.edit_dist = ed; out.h16.mismatches = mm8; out.h16.read_len = (uint8_t)(read_len > 255 ? 255 : read_len);
    out.h32.ref_id = ref_id; out.h32.left = left;
    out.h32.flags = (uint8_t)(out.h16.flags | (jstrand == "rev" ? THJ_HIT_ANTISENSE_SPLICE : 0) | (flipped ? THJ_HIT_STRAND_FLIPPED : 0) |
                              (ref_id2 ? THJ_HIT_FUSED : 0));
    out.h32.mismatches = mm8; out.h32.edit_dist = ed; out.h32.n_cigar = (uint8_t)spl.size();
    return true;
}

template <class T>
class ChunkQueue {
    std::mutex mu_; std::condition_variable cv_;
    std::vector<std::vector<T>> q_;
    size_t cap_ = 4;
    bool closed_ = false, aborted_ = false;
public:
    bool push(std::vector<T>&& c) {            // false: the consumer is gone
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return q_.size() < cap_ || aborted_; });
        if (aborted_) return false;
        q_.push_back(std::move(c));
        cv_.notify_all();
        return true;
    }
    void close() { std::lock_guard<std::mutex> lk(mu_); closed_ = true; cv_.notify_all(); }
    void abort() { std::lock_guard<std::mutex> lk(mu_); aborted_ = true; cv_.notify_all(); }
    bool pop(std::vector<T>& out) {            // false: end of stream
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.erase(q_.begin());
        cv_.notify_all();
        return true;
    }
};

// HitStream (bwt_map.h:1040-1227): groups of consecutive records with equal insert_id, one-record look-ahead.
// Records the factory drops never enter the stream, as in the reference.
class HitStream {
    AlnReader rd_;
    RefTable* rt_ = nullptr;
    const thj_params* p_ = nullptr;
    bool spliced_ = false, done_ = true;
    uint32_t begin_id_ = 0, end_id_ = 0xFFFFFFFFu;      // a shard: records with begin_id <= insert_id < end_id (segment_juncs.cpp:4005)
    std::thread th_;
    ChunkQueue<Hit> q_;
    std::vector<Hit> cur_; size_t pos_ = 0;
    void producer() {
        std::vector<Hit> chunk;
        chunk.reserve(8192);
        AlnRec r; Hit h;
        const bool lean = rd_.is_bam() && !spliced_;
        std::vector<uint32_t> tid2ref;
        if (lean) for (auto& t : rd_.targets()) tid2ref.push_back(rt_->get_id(t));
        for (;;) {
            h = Hit();
            if (lean) {
                int32_t bs = 0;
                const uint8_t* d = rd_.next_raw(bs);
                if (!d) break;
                if (!parse_hit_bam(d, bs, tid2ref, *p_, h)) continue;
            } else {
                if (!rd_.next(r)) break;
                if (!(spliced_ ? parse_spliced_hit(r, *rt_, *p_, h, rd_.is_bam()) : parse_hit(r, *rt_, *p_, h))) continue;
            }
            if (h.insert_id < begin_id_) continue;       // the index entry the shard starts from lies at or before begin_id
            if (h.insert_id >= end_id_) break;           // id-sorted file: the shard is over
            chunk.push_back(h);
            if (chunk.size() >= 8192) { if (!q_.push(std::move(chunk))) return; chunk = std::vector<Hit>(); chunk.reserve(8192); }
        }
        if (!chunk.empty()) q_.push(std::move(chunk));
        q_.close();
    }
    void advance() { while (!done_ && pos_ >= cur_.size()) { pos_ = 0; if (!q_.pop(cur_)) { cur_.clear(); done_ = true; } } }
public:
    HitStream() = default;
    HitStream(const HitStream&) = delete;
    HitStream& operator=(const HitStream&) = delete;
    bool open(const std::string& fn, RefTable& rt, const thj_params& p, bool spliced = false, int64_t offset = 0,
              uint64_t begin_id = 0, uint64_t end_id = ~0ull) {
        rt_ = &rt; p_ = &p; spliced_ = spliced;
        begin_id_ = begin_id > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)begin_id;
        end_id_ = end_id > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)end_id;
        rd_.want_seq = false;                    // neither hit factory looks at the bases, only at their number
        if (fn.empty() || !rd_.open(fn, offset)) return false;
        done_ = false;
        th_ = std::thread([this] { producer(); });
        advance();
        return true;
    }
    ~HitStream() { if (th_.joinable()) { q_.abort(); th_.join(); } }
    uint32_t next_group_id() const { return done_ ? 0 : cur_[pos_].insert_id; }
    // appends the next group to `out`; returns its id (0 at end)
    uint32_t next_group(std::vector<Hit>& out) {
        if (done_) return 0;
        uint32_t id = cur_[pos_].insert_id;
        while (!done_ && cur_[pos_].insert_id == id) { out.push_back(cur_[pos_]); ++pos_; advance(); }
        return id;
    }
    void skip_group() {
        if (done_) return;
        uint32_t id = cur_[pos_].insert_id;
        while (!done_ && cur_[pos_].insert_id == id) { ++pos_; advance(); }
    }
};

// ------------------------------------------------------------------ reads (reads.cpp:94-188, :528-630)
struct Read { uint32_t id = 0; std::string name, seq, qual; const uint8_t* raw = nullptr; /* the read's own BAM record (after block_size) when it came inflated from the device */
              bool qc_fail = false; /* a BAM record with BAM_FQCFAIL: ReadStream::get_direct (reads.cpp:546-560) reads on past it -- get() does, and next_direct()'s callers look at the flag */ };

class ReadStream {
    FILE* f_ = nullptr; bool pipe_ = false;
    AlnReader bam_; bool is_bam_ = false;
    char* line_ = nullptr; size_t cap_ = 0;
    std::string pending_;     // pushed-back header line
    std::map<uint32_t, Read> ahead_;
    bool eof_ = false;
    std::thread th_;
    ChunkQueue<Read> q_;
    std::vector<Read> cur_; size_t pos_ = 0;
    void producer() {
        std::vector<Read> chunk;
        Read r;
        while (next(r)) {
            chunk.push_back(std::move(r));
            r = Read();
            if (chunk.size() >= 4096) { if (!q_.push(std::move(chunk))) return; chunk = std::vector<Read>(); }
        }
        if (!chunk.empty()) q_.push(std::move(chunk));
        q_.close();
    }
    bool next_async(Read& r) {
        while (pos_ >= cur_.size()) { pos_ = 0; if (!q_.pop(cur_)) { cur_.clear(); return false; } }
        r = std::move(cur_[pos_++]);
        return true;
    }
    bool getl(std::string& s) {
        if (!pending_.empty()) { s.swap(pending_); pending_.clear(); return true; }
        ssize_t n;
        while ((n = getline(&line_, &cap_, f_)) > 0) {
            s.assign(line_, (size_t)n);
            while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
            return true;
        }
        return false;
    }
    bool next(Read& r) {
        if (is_bam_) {
            AlnRec a;
            if (!bam_.next(a)) return false;
            r.id = (uint32_t)atoi(a.qname.c_str()); r.name = a.qname; r.seq = a.seq; r.qual = a.qual; r.qc_fail = (a.flag & 0x200u) != 0;
            return true;
        }
        std::string l;
        do { if (!getl(l)) return false; } while (l.empty());
        if (l[0] != '@' && l[0] != '>') die("Error: unrecognised reads file format\n");
        bool fq = l[0] == '@';
        std::string name = l.substr(1);
        size_t sp = name.find_first_of(" \t");
        if (sp != std::string::npos) name.resize(sp);
        r.id = (uint32_t)atoi(name.c_str());
        r.name = name;
        r.seq.clear(); r.qual.clear();
        while (getl(l)) {
            if (l.empty()) continue;
            if ((fq && l[0] == '+') || (!fq && l[0] == '>')) { if (!fq) pending_ = l; break; }
            r.seq += l;
        }
        std::replace(r.seq.begin(), r.seq.end(), '.', 'N');          // reads.cpp:119
        if (fq) {
            while (r.qual.size() < r.seq.size() && getl(l)) r.qual += l;
            if (r.qual.size() != r.seq.size()) die("Error: qual length (%d) differs from seq length (%d) for fastq record %s!\n", (int)r.qual.size(), (int)r.seq.size(), name.c_str());
        } else r.qual.assign(r.seq.size(), 'I');
        return !r.seq.empty();
    }
public:
    // offset: where the shard's reads start (ReadStream::seek, reads.h:431-434): a BGZF virtual offset for an unaligned
    // BAM, a byte offset at a record start for FASTA / FASTQ text
    bool open(const std::string& fn, const std::string& zpacker, int64_t offset = 0) {
        std::string e = file_ext(fn);
        if (e == "bam") { is_bam_ = true; if (!bam_.open(fn, offset)) return false; }
        else {
            if (e == "z" && !zpacker.empty()) {                      // FZPipe, common.cpp:899-922
                std::string cmd = zpacker + " -cd '" + fn + "'";
                f_ = popen(cmd.c_str(), "r"); pipe_ = true;
            } else f_ = fopen(fn.c_str(), "r");
            if (!f_) return false;
            if (offset > 0 && !pipe_) fseek(f_, (long)offset, SEEK_SET);
        }
        th_ = std::thread([this] { producer(); });
        return true;
    }
    ReadStream() = default;
    ReadStream(const ReadStream&) = delete;
    ReadStream& operator=(const ReadStream&) = delete;
    ~ReadStream() {
        if (th_.joinable()) { q_.abort(); th_.join(); }
        if (f_) { if (pipe_) pclose(f_); else fclose(f_); }
        free(line_);
    }
    // the next read of the file, whatever its id (ReadStream::get_direct, reads.cpp:600-630)
    bool next_direct(Read& out) { return next_async(out); }
    // monotone fetch: requests arrive in increasing id order (the visiting order of both stages)
    bool get(uint32_t id, Read& out) {
        auto it = ahead_.find(id);
        if (it != ahead_.end()) { out = std::move(it->second); ahead_.erase(ahead_.begin(), std::next(it)); return true; }
        Read r;
        while (!eof_) {
            if (!next_async(r)) { eof_ = true; break; }
            if (r.qc_fail) continue;                  // reads.cpp:556: a QC-failed record is not a read
            if (r.id == id) { out = std::move(r); while (!ahead_.empty() && ahead_.begin()->first < id) ahead_.erase(ahead_.begin()); return true; }
            if (r.id > id) ahead_[r.id] = r;          // slightly out-of-order files (reads.h:142 keeps a 500k-entry heap)
        }
        return false;
    }
};

// ------------------------------------------------------------------ read-id shards (utils.cpp:22-170)
// The reference's worker threads (-p N) each take a contiguous read-id range [begin_id, end_id) and start reading every
// input at an offset taken from its `.index` side file (`read_id \t offset` lines, written every >= 1000 records at a
// read-id change: GBamWriter, common.h:562-606).  The same plan shards the reads over host workers and GPUs here; the
// results do not depend on the number of shards (events are sets, first-inserted-wins priorities follow the read id).
typedef std::vector<std::pair<uint64_t, int64_t>> IndexList;

inline bool read_index_file(const std::string& fn, IndexList& out) {
    FILE* f = fopen(fn.c_str(), "r");
    if (!f) return false;
    unsigned long long id; long long off;
    while (fscanf(f, "%llu %lld", &id, &off) == 2) out.emplace_back((uint64_t)id, (int64_t)off);
    fclose(f);
    return true;
}

// Plain-text inputs (FASTA / FASTQ reads, SAM maps: test fixtures and small runs) have no `.index`.  They are seekable, so
// an index is made by probing: at evenly spaced byte offsets, the next place where a record GROUP starts and its id.
inline void probe_text_index(const std::string& fn, int want, IndexList& out) {
    const std::string ext = file_ext(fn);
    const bool sam = ext == "sam";
    FILE* f = fopen(fn.c_str(), "rb");
    if (!f) return;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    if (size < (long)want * 4096) { fclose(f); return; }            // small file: not worth cutting
    std::vector<char> buf((size_t)1 << 18);
    for (int k = 1; k < want; ++k) {
        const long at = (long)((double)size * k / want);
        fseek(f, at, SEEK_SET);
        const size_t n = fread(buf.data(), 1, buf.size(), f);
        // line starts inside the buffer
        std::vector<size_t> ls;
        for (size_t i = 0; i + 1 < n; ++i) if (buf[i] == '\n') ls.push_back(i + 1);
        auto line_id = [&](size_t b) { return (uint64_t)strtoull(buf.data() + b + (sam ? 0 : 1), nullptr, 10); };
        bool found = false;
        if (sam) {
            // first line whose id differs from the line before it: a group start
            for (size_t j = 1; j < ls.size() && !found; ++j) {
                if (buf[ls[j - 1]] == '@' || buf[ls[j]] == '@') continue;
                if (line_id(ls[j]) != line_id(ls[j - 1])) { out.emplace_back(line_id(ls[j]), (int64_t)(at + (long)ls[j])); found = true; }
            }
        } else {
            for (size_t j = 0; j + 2 < ls.size() && !found; ++j) {
                const char c = buf[ls[j]];
                if (c == '>') { out.emplace_back(line_id(ls[j]), (int64_t)(at + (long)ls[j])); found = true; }
                // four-line FASTQ: a header line is an '@' line whose line after next starts with '+' (a quality line that
                // happens to start with '@' is followed by a header and a sequence line instead)
                else if (c == '@' && buf[ls[j + 2]] == '+' && buf[ls[j + 1]] != '@' && buf[ls[j + 1]] != '+') {
                    out.emplace_back(line_id(ls[j]), (int64_t)(at + (long)ls[j])); found = true;
                }
            }
        }
        if (!found) { out.clear(); break; }                          // unusual layout (wrapped FASTQ ...): do not shard this file
    }
    // ids must increase along the file for the plan to mean anything
    for (size_t i = 1; i < out.size(); ++i) if (out[i].first <= out[i - 1].first) { out.clear(); break; }
    fclose(f);
}

inline void load_index(const std::string& fname, int want, IndexList& out) {
    // utils.cpp:33-71: "<name>.bam.index", or "<name><j>.bam.index" for j = 0, 1, ... when <name> is not a .bam
    if (fname.size() >= 4 && fname.substr(fname.size() - 4) == ".bam") { read_index_file(fname + ".index", out); return; }
    const std::string ext = file_ext(fname);
    if (ext == "sam" || ext == "fq" || ext == "fastq" || ext == "fa" || ext == "fasta") { probe_text_index(fname, want, out); return; }
    for (size_t j = 0;; ++j) if (!read_index_file(fname + std::to_string(j) + ".bam.index", out)) break;
}

// calculate_offsets (utils.cpp:22-127): n - 1 boundary read ids from the LAST file's index, and for every file the offset of
// its last index entry at or before the boundary (walking the files from the last to the first, each one bounded by the
// entry chosen in the file after it).  lists[i] = index of file i.  false: "too small for blocking".
inline bool calculate_offsets(const std::vector<IndexList>& lists, int n, std::vector<uint64_t>& ids, std::vector<std::vector<int64_t>>& offsets) {
    ids.clear(); offsets.clear();
    if (n < 2 || lists.empty()) return false;
    for (auto& l : lists) if (l.size() < (size_t)n) return false;
    offsets.resize((size_t)n - 1);
    for (int i = 1; i < n; ++i) {
        const IndexList& last = lists.back();
        const size_t index = last.size() / (size_t)n * (size_t)i;
        uint64_t id = last[index].first;
        ids.push_back(id);
        std::vector<int64_t>& off = offsets[(size_t)i - 1];
        off.push_back(last[index].second);
        for (int j = (int)lists.size() - 2; j >= 0; --j) {
            const IndexList& l = lists[(size_t)j];
            size_t oi = l.size() / (size_t)n * (size_t)i;
            uint64_t oid = l[oi].first;
            while (oid > id && oi > 0) { --oi; oid = l[oi].first; }
            while (oi + 1 < l.size() && l[oi + 1].first < id) { ++oi; oid = l[oi].first; }
            int64_t ooff = l[oi].second;
            if (oid > id) { oid = 0; ooff = 0; }
            id = oid;
            off.push_back(ooff);
        }
        std::reverse(off.begin(), off.end());
    }
    // equal boundaries (very uneven indexes) would make empty shards; they are harmless but pointless
    return true;
}

// calculate_offsets_from_ids (utils.cpp:129-170): for a file that is only looked up by id (the mate's maps), the offset of
// its last index entry with id <= the boundary.  An empty result means "read the file from the start".
inline void calculate_offsets_from_ids(const IndexList& l, const std::vector<uint64_t>& ids, std::vector<int64_t>& offsets) {
    offsets.clear();
    size_t k = 0;
    uint64_t last_id = 0; int64_t last_off = 0;
    for (size_t i = 0; i < ids.size(); ++i) {
        const uint64_t ref = ids[i];
        bool pushed = false;
        while (k < l.size()) {
            const uint64_t rid = l[k].first; const int64_t off = l[k].second;
            ++k;
            if (rid > ref) { offsets.push_back(last_id <= ref ? last_off : 0); pushed = true; }
            last_id = rid; last_off = off;
            if (last_id > ref) break;
        }
        if (!pushed) break;
    }
    if (ids.size() != offsets.size()) offsets.clear();
}

// Where a shard's share of a file ends, for readers that take a whole piece of the file at once (the device-side ingest): the
// offset of the first index entry whose read id is >= end_id -- every record of a smaller id lies before it.  -1: up to the end.
inline int64_t shard_end_offset(const IndexList& l, uint64_t end_id) {
    size_t lo = 0, hi = l.size();
    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (l[mid].first < end_id) lo = mid + 1; else hi = mid; }
    return lo < l.size() ? l[lo].second : -1;
}

// every contig an input's header names enters the reference table before it is frozen (and before any record is parsed)
inline void register_targets(const std::string& fn, RefTable& rt) {
    if (fn.empty()) return;
    if (file_ext(fn) == "sam") {
        FILE* f = fopen(fn.c_str(), "r");
        if (!f) return;
        char* line = nullptr; size_t cap = 0; ssize_t n;
        while ((n = getline(&line, &cap, f)) > 0 && line[0] == '@') {
            if (strncmp(line, "@SQ", 3)) continue;
            std::string l(line, (size_t)n);
            while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
            for (auto& tok : split(l, '\t')) if (!tok.compare(0, 3, "SN:")) rt.get_id(tok.substr(3));
        }
        free(line);
        fclose(f);
        return;
    }
    if (file_ext(fn) != "bam") return;
    AlnReader r;
    if (!r.open(fn)) return;
    for (auto& t : r.targets()) {
        // junction-db contigs "name|left|l-r|right|type|strand" are resolved to their genomic contig by the spliced hit factory
        if (t.find('|') != std::string::npos) continue;
        rt.get_id(t);
    }
}

// ------------------------------------------------------------------ BAM files for the device-side ingest
// The host's whole part in the device path: map the file, know where the header ends and what its targets are called,
// and hand out [member-aligned piece of the mapping, bytes to skip in its first member] for a shard.
struct BamFile {
    const uint8_t* data = nullptr; size_t size = 0;
    int fd = -1;                                      // kept open: shard pieces are pread() into page-locked staging buffers (stage_pieces)
    std::vector<std::string> targets;
    std::vector<uint32_t> tid2ref;
    int64_t first_rec_voff = 0;                       // virtual offset of the first alignment record
    static uint32_t member_size(const uint8_t* d, size_t n) {            // BSIZE + 1 of the member at d, 0 if d is not one
        if (n < 28 || d[0] != 31 || d[1] != 139 || d[2] != 8 || !(d[3] & 4)) return 0;
        const uint32_t xlen = d[10] | (d[11] << 8);
        size_t x = 12; const size_t xe = 12 + xlen;
        while (x + 4 <= xe && xe <= n) {
            const uint32_t slen = d[x + 2] | (d[x + 3] << 8);
            if (d[x] == 'B' && d[x + 1] == 'C' && slen == 2) return (uint32_t)(d[x + 4] | (d[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        return 0;
    }
    bool open(const std::string& fn, RefTable& rt) {
        if (file_ext(fn) != "bam") return false;
        const int fd = ::open(fn.c_str(), O_RDONLY);
        if (fd < 0) return false;
        const off_t sz = lseek(fd, 0, SEEK_END);
        if (sz <= 0) { ::close(fd); return false; }
        void* m = mmap(nullptr, (size_t)sz, PROT_READ, MAP_SHARED, fd, 0);
        if (m == MAP_FAILED) { ::close(fd); return false; }
        this->fd = fd;
        data = (const uint8_t*)m; size = (size_t)sz;
        // header: inflate members from the start until magic, text, n_ref and the n_ref (name, length) entries are in hand
        std::vector<uint8_t> h;
        std::vector<std::pair<int64_t, size_t>> mem;              // (file offset, inflated size) of the members read so far
        size_t off = 0, need = 12;
        auto have = [&](size_t n) {
            while (h.size() < n && off < size) {
                const uint32_t bs = member_size(data + off, size - off);
                if (!bs || off + bs > size) return false;
                const uint32_t xlen = data[off + 10] | (data[off + 11] << 8);
                uint32_t isz; memcpy(&isz, data + off + bs - 4, 4);
                const size_t at = h.size();
                h.resize(at + isz);
                z_stream zs; memset(&zs, 0, sizeof zs);
                inflateInit2(&zs, -15);
                zs.next_in = const_cast<uint8_t*>(data + off + 12 + xlen); zs.avail_in = bs - 12 - xlen - 8;
                zs.next_out = h.data() + at; zs.avail_out = isz;
                const int st = inflate(&zs, Z_FINISH);
                inflateEnd(&zs);
                if (st != Z_STREAM_END && isz) return false;
                mem.emplace_back((int64_t)off, (size_t)isz);
                off += bs;
            }
            return h.size() >= n;
        };
        if (!have(need) || memcmp(h.data(), "BAM\1", 4)) return false;
        int32_t l_text; memcpy(&l_text, h.data() + 4, 4);
        need = 8 + (size_t)l_text + 4;
        if (!have(need)) return false;
        int32_t n_ref; memcpy(&n_ref, h.data() + 8 + l_text, 4);
        size_t p = need;
        for (int32_t i = 0; i < n_ref; ++i) {
            if (!have(p + 4)) return false;
            int32_t l_name; memcpy(&l_name, h.data() + p, 4);
            if (!have(p + 4 + (size_t)l_name + 4)) return false;
            targets.emplace_back((const char*)h.data() + p + 4);
            p += 4 + (size_t)l_name + 4;
        }
        // p = inflated offset of the first record -> (member, offset inside it)
        size_t acc = 0;
        first_rec_voff = (int64_t)off << 16;                  // right after the members read, unless it falls inside one
        for (auto& me : mem) { if (p < acc + me.second) { first_rec_voff = (me.first << 16) | (int64_t)(p - acc); break; } acc += me.second; }
        for (auto& t : targets) tid2ref.push_back(t.find('|') == std::string::npos ? rt.get_id(t) : 0u);
        return true;
    }
    // the piece of the file that holds the records from virtual offset `from` up to (not including) those at `to` (< 0: the end)
    thj_bam_piece piece(int64_t from, int64_t to) const {
        if (from < first_rec_voff) from = first_rec_voff;
        thj_bam_piece pc;
        const size_t a = (size_t)(from >> 16);
        size_t e = size;
        if (to >= 0) {
            e = (size_t)(to >> 16);
            if ((to & 0xFFFF) && e < size) e += member_size(data + e, size - e);      // the member the boundary falls into belongs to both sides
            if (e < a) e = a;
            if (e > size) e = size;
        }
        pc.comp = data + (a < size ? a : size); pc.comp_bytes = (int64_t)(e > a ? e - a : 0); pc.first_skip = (uint32_t)(from & 0xFFFF);
        pc.n_tid = (int32_t)tid2ref.size(); pc.tid2ref = tid2ref.data();
        return pc;
    }
    ~BamFile() { if (data) munmap((void*)data, size); if (fd >= 0) ::close(fd); }
    BamFile() = default;
    BamFile(const BamFile&) = delete;
    BamFile& operator=(const BamFile&) = delete;
};

// A shard's compressed pieces, read into ONE page-locked buffer (thj_pinned_alloc; the caller thj_pinned_free()s it once the device
// has them) and the pieces pointed at it.  pread() from the page cache, not a copy out of the mapping: the bulk of the files then
// never gets page-table entries in this process (a fault per page on the way in, and 0.06 s per GB to take apart when the process
// leaves -- more when the runtime had to lock the mapping's pages for its own DMA: 0.16 s of segment_juncs' 0.27 s between its report
// and its end on 10 M pairs), and the copy up is plain DMA inside the GPU's lock.  Runs outside that lock, beside the other workers.
// Returns null (pieces untouched: they still point into the mappings) when there is no buffer or a read fails.
inline uint8_t* stage_pieces(const std::vector<std::pair<const BamFile*, thj_bam_piece*>>& pcs) {
    if (getenv("THJ_NO_STAGING")) return nullptr;
    size_t total = 0;
    for (auto& q : pcs) if (q.first && q.second) total += (size_t)q.second->comp_bytes;
    uint8_t* stage = (uint8_t*)thj_pinned_alloc(total + 64);
    if (!stage) return nullptr;
    size_t at = 0;
    std::vector<size_t> where;
    for (auto& q : pcs) {
        where.push_back(at);
        if (!q.first || !q.second) continue;
        const thj_bam_piece& pc = *q.second;
        size_t done = 0; const size_t n = (size_t)pc.comp_bytes;
        const off_t off0 = (off_t)(pc.comp - q.first->data);
        while (done < n) {
            const ssize_t r = q.first->fd >= 0 ? pread(q.first->fd, stage + at + done, n - done, off0 + (off_t)done) : -1;
            if (r <= 0) { if (r < 0 && errno == EINTR) continue; break; }
            done += (size_t)r;
        }
        if (done < n) memcpy(stage + at + done, pc.comp + done, n - done);       // (a short read: the rest from the mapping)
        at += n;
    }
    for (size_t i = 0; i < pcs.size(); ++i) if (pcs[i].first && pcs[i].second) pcs[i].second->comp = stage + where[i];
    return stage;
}

// ------------------------------------------------------------------ BGZF + BAM writer (samtools-0.1.18 bgzf.c, common.cpp:1000-1173)
inline int reg2bin(int beg, int end) {                // bam.h bam_reg2bin
    --end;
    if (beg >> 14 == end >> 14) return 4681 + (beg >> 14);
    if (beg >> 17 == end >> 17) return 585 + (beg >> 17);
    if (beg >> 20 == end >> 20) return 73 + (beg >> 20);
    if (beg >> 23 == end >> 23) return 9 + (beg >> 23);
    if (beg >> 26 == end >> 26) return 1 + (beg >> 26);
    return 0;
}

// GBamWriter with the read-id -> BGZF-offset side file (common.h:562-606).
// The uncompressed stream is cut into BGZF members exactly as samtools-0.1.18 does: the header in members of its own, and a
// record that does not fit the rest of a 64 KiB member starts the next one (bgzf_flush_try) -- so no record straddles two
// members, which is also what lets the device-side ingest parse members independently; records of a batch are
// encoded by worker threads, the blocks of the batch are deflated by worker threads (blocks are independent
// members), and one thread writes them in order and replays the `.index` rule on the now-known block addresses.
#ifndef THJ_BGZF_DEFAULT_LEVEL
#define THJ_BGZF_DEFAULT_LEVEL 1
#endif
class BamWriter {
    FILE* f_ = nullptr;
    FILE* idx_ = nullptr;
    std::unordered_map<std::string, int32_t> tid_;
    std::vector<uint8_t> carry_;                    // bytes after the last full block (always starts a block)
    int64_t file_addr_ = 0;                         // compressed bytes written = address of the block `carry_` opens
    uint64_t idxcount_ = 0; long last_id_ = 0;
    static const size_t BLOCK = 0x10000;
    static void put32(std::vector<uint8_t>& v, uint32_t x) { uint8_t b[4]; memcpy(b, &x, 4); v.insert(v.end(), b, b + 4); }

    // one BGZF member from `take` input bytes; false when they do not fit (bgzf.c deflate_block then shrinks its input)
    static bool deflate_member(const uint8_t* in, size_t take, std::vector<uint8_t>& out) {
        out.resize(BLOCK + 1024);
        // The default compressor is thj_fastdeflate.h (greedy LZ77 + one dynamic-Huffman block per member: zlib level 1's ratio on BAM
        // records at several times its speed); setting THJ_BGZF_LEVEL selects zlib at that level (-1: zlib's default, what bgzf.c
        // uses).  Members the fast path declines (tiny ones -- the 28-byte EOF marker must be zlib's --, output that does not fit)
        // go to zlib as well.
        static const bool use_zlib = getenv("THJ_BGZF_LEVEL") != nullptr;
        if (!use_zlib && take >= 64) {
            size_t clen = 0;
            if (fdz::deflate_fast(in, take, out.data() + 18, BLOCK - 18 - 8, &clen)) { finish_member(in, take, clen, out); return true; }
        }
        z_stream zs; memset(&zs, 0, sizeof zs);
        // bgzf.c compresses at zlib's default level (6); this writer defaults to level 1 -- the BAM stream inside is the same, the
        // file 9 % larger, the deflate 3x cheaper (measured: 0.6 s of a 2.7 s long_spanning_reads run); THJ_BGZF_LEVEL=-1 restores zlib's default
        static const int level = getenv("THJ_BGZF_LEVEL") ? atoi(getenv("THJ_BGZF_LEVEL")) : THJ_BGZF_DEFAULT_LEVEL;
        deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
        zs.next_in = const_cast<uint8_t*>(in); zs.avail_in = (uInt)take;
        zs.next_out = out.data() + 18; zs.avail_out = (uInt)(BLOCK - 18 - 8);
        int st = deflate(&zs, Z_FINISH);
        size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (st != Z_STREAM_END) return false;
        finish_member(in, take, clen, out);
        return true;
    }
    // BGZF envelope round the clen bytes of DEFLATE data already at out[18..)
    static void finish_member(const uint8_t* in, size_t take, size_t clen, std::vector<uint8_t>& out) {
        static const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
        memcpy(out.data(), hdr, 12);
        out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
        uint16_t bs = (uint16_t)(clen + 18 + 8 - 1);
        memcpy(out.data() + 16, &bs, 2);
        uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), in, (uInt)take), isize = (uint32_t)take;
        memcpy(out.data() + 18 + clen, &crc, 4);
        memcpy(out.data() + 18 + clen + 4, &isize, 4);
        out.resize(clen + 26);
    }

    struct Blk { size_t ustart, ulen; int64_t addr; };
    // Where BGZF members end in `stream` (= carry_ + the new records' bytes): bam_write1 first calls
    // bgzf_flush_try(4 + block_len) -- a record that does not fit what is left of the 64 KiB block starts a new one
    // (bam.c:225, bgzf.c:587-592) -- then bgzf_write fills the block and flushes whenever it is full (bgzf.c:594-623), which only
    // happens inside records larger than a block.  `off` = fill of the open block when the first new record arrives.
    static void plan_cuts(size_t off, size_t pos, const std::vector<uint32_t>& size, size_t first, std::vector<size_t>& cuts, size_t* end_off) {
        for (size_t i = first; i < size.size(); ++i) {
            size_t s = size[i];
            if (off + s > BLOCK && off > 0) { cuts.push_back(pos); off = 0; }
            while (s > 0) {
                const size_t c = std::min(BLOCK - off, s);
                off += c; pos += c; s -= c;
                if (off == BLOCK) { cuts.push_back(pos); off = 0; }
            }
        }
        *end_off = off;
    }
    void write_member(const std::vector<uint8_t>& m) { fwrite(m.data(), 1, m.size(), f_); file_addr_ += (int64_t)m.size(); }
    // Compresses and writes the closed members of `stream`; returns the block table including the still-open last block, whose
    // bytes become the new carry.  Members are deflated in parallel; should one not fit its 64 KiB envelope (incompressible
    // data: bgzf.c deflate_block then shrinks the input by 1 KiB steps and the rest opens the next block) the remainder of the
    // call is replayed sequentially with that rule, since every later cut moves.
    std::vector<Blk> flush_blocks(const std::vector<uint8_t>& stream, const std::vector<uint32_t>& size, bool final_flush) {
        std::vector<size_t> cuts;
        size_t end_off = 0;
        plan_cuts(carry_.size(), carry_.size(), size, 0, cuts, &end_off);
        if (final_flush && (cuts.empty() || cuts.back() != stream.size()) && !stream.empty()) cuts.push_back(stream.size());
        const size_t nb = cuts.size();
        std::vector<std::vector<uint8_t>> out(nb);
        std::vector<char> ok(nb, 1);
        const int T = host_threads();
        auto work = [&](int t) { for (size_t k = (size_t)t; k < nb; k += (size_t)T) { const size_t a = k ? cuts[k - 1] : 0; ok[k] = deflate_member(stream.data() + a, cuts[k] - a, out[k]); } };
        if (nb > 1 && T > 1) {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back(work, t);
            for (auto& x : th) x.join();
        } else for (int t = 0; t < T; ++t) work(t);
        std::vector<Blk> tab;
        size_t upos = 0, k = 0;
        for (; k < nb && ok[k]; ++k) {
            tab.push_back({upos, cuts[k] - upos, file_addr_});
            write_member(out[k]);
            upos = cuts[k];
        }
        if (k < nb) {
            // sequential replay from `upos` with the shrink rule: record ends tell where bgzf_flush_try looks
            std::vector<size_t> rec_end;
            { size_t x = carry_.size(); for (auto sz : size) { x += sz; rec_end.push_back(x); } }
            size_t ri = 0;
            while (ri < rec_end.size() && rec_end[ri] <= upos) ++ri;
            size_t blk_start = upos, pos = upos;
            std::vector<uint8_t> o;
            auto flush = [&](size_t upto) {                       // bgzf_flush (bgzf.c:568-585): members until the block is empty; what
                while (blk_start < upto) {                        // deflate_block could not fit stays in the block and goes next
                    size_t take = upto - blk_start;
                    while (!deflate_member(stream.data() + blk_start, take, o)) take -= 1024;
                    tab.push_back({blk_start, take, file_addr_});
                    write_member(o);
                    blk_start += take;
                }
            };
            while (pos < stream.size()) {
                const size_t rend = ri < rec_end.size() ? rec_end[ri] : stream.size();
                const size_t rsize = rend - pos;
                if ((pos - blk_start) + rsize > BLOCK && pos > blk_start) flush(pos);
                size_t left = rsize;
                while (left > 0) {
                    const size_t c = std::min(BLOCK - (pos - blk_start), left);
                    pos += c; left -= c;
                    if (pos - blk_start == BLOCK) flush(pos);
                }
                ++ri;
            }
            if (final_flush) flush(stream.size());
            upos = blk_start;
        }
        tab.push_back({upos, stream.size() - upos, file_addr_});       // the open block
        return tab;
    }
    static int64_t tell_at(const std::vector<Blk>& tab, size_t x) {       // bgzf_tell of stream offset x
        size_t lo = 0, hi = tab.size();
        while (hi - lo > 1) { size_t mid = (lo + hi) / 2; if (tab[mid].ustart <= x) lo = mid; else hi = mid; }
        return (tab[lo].addr << 16) | (int64_t)(x - tab[lo].ustart);
    }
public:
    bool open(const std::string& fn, const RefTable& rt, const std::string& idx_fn) {
        f_ = fopen(fn.c_str(), "wb");
        if (!f_) return false;
        if (!idx_fn.empty()) { idx_ = fopen(idx_fn.c_str(), "w"); if (!idx_) return false; }
        std::vector<uint8_t>& h = carry_;
        h.insert(h.end(), {'B', 'A', 'M', 1});
        put32(h, (uint32_t)rt.header_text.size());
        h.insert(h.end(), rt.header_text.begin(), rt.header_text.end());
        put32(h, (uint32_t)rt.sq.size());
        for (size_t i = 0; i < rt.sq.size(); ++i) {
            put32(h, (uint32_t)rt.sq[i].first.size() + 1);
            h.insert(h.end(), rt.sq[i].first.begin(), rt.sq[i].first.end());
            h.push_back(0);
            put32(h, rt.sq[i].second);
            tid_[rt.sq[i].first] = (int32_t)i;
        }
        // bam_header_write ends with bgzf_flush (bam.c:144): the header has its members to itself, records start a fresh one
        std::vector<uint8_t> hb; hb.swap(carry_);
        std::vector<uint8_t> o;
        for (size_t at = 0; at < hb.size();) {
            size_t take = std::min(BLOCK, hb.size() - at);
            while (!deflate_member(hb.data() + at, take, o)) take -= 1024;
            write_member(o);
            at += take;
        }
        return true;
    }
    // Appends block_size + one record exactly as GBamRecord builds it: mate fields "*", 0, 0; MAPQ 255.  Thread-safe.
    int32_t tid_of(const std::string& rname) const { auto it = tid_.find(rname); return it == tid_.end() ? -1 : it->second; }
    void encode(std::vector<uint8_t>& d, const std::string& qname, uint32_t flag, const std::string& rname, int pos1,
                const uint32_t* cigar /*op<<28|len*/, int n_cigar, const std::string& seq, const std::string& qual,
                const std::vector<std::string>& aux) const {
        static const uint8_t nt16[256] = {
            15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
            15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 1,2,4,8,15,15,15,15,15,15,15,15,15,0,15,15,
            15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15,
            15,1,14,2,13,15,15,4,11,15,15,12,15,3,15,15, 15,15,5,6,8,15,7,9,15,10,15,15,15,15,15,15,
            15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
            15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
            15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,
            15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15, 15,15,15,15,15,15,15,15,15,15,15,15,15,15,15,15};
        static const uint8_t bamop[16] = {0, 0, 0, 1, 1, 2, 2, 0, 0, 0, 0, 3, 3, 4, 5, 6};   // upper-cased letters (set_cigar :1044-1053)
        const size_t at = d.size();
        put32(d, 0);                                                           // block_size, patched below
        auto it = tid_.find(rname);
        int32_t tid = it == tid_.end() ? -1 : it->second;
        int32_t pos = pos1 <= 0 ? -1 : pos1 - 1;
        int end = pos;
        for (int i = 0; i < n_cigar; ++i) { uint32_t c = cigar[i], op = bamop[c >> 28]; if (op == 0 || op == 2 || op == 3) end += (int)(c & 0x0FFFFFFF); }
        uint32_t bin = (uint32_t)reg2bin(pos, n_cigar == 0 ? pos + 1 : end);
        uint32_t l_rn = (uint32_t)qname.size() + 1;
        put32(d, (uint32_t)tid); put32(d, (uint32_t)pos);
        put32(d, (bin << 16) | (255u << 8) | l_rn);
        put32(d, (flag << 16) | (uint32_t)n_cigar);
        put32(d, (uint32_t)seq.size());
        put32(d, (uint32_t)-1); put32(d, (uint32_t)-1); put32(d, 0);          // mtid, mpos (0-1), isize
        d.insert(d.end(), qname.begin(), qname.end()); d.push_back(0);
        for (int i = 0; i < n_cigar; ++i) put32(d, ((cigar[i] & 0x0FFFFFFF) << 4) | bamop[cigar[i] >> 28]);
        size_t so = d.size();
        d.resize(so + (seq.size() + 1) / 2, 0);
        for (size_t i = 0; i < seq.size(); ++i) d[so + i / 2] |= (uint8_t)(nt16[(uint8_t)seq[i]] << (4 * (1 - i % 2)));
        for (size_t i = 0; i < seq.size(); ++i) d.push_back((uint8_t)(qual[i] - 33));
        for (auto& a : aux) {                                                  // add_aux, common.cpp:1092-1173
            d.push_back((uint8_t)a[0]); d.push_back((uint8_t)a[1]);
            char ty = a[3];
            if (ty == 'A' || ty == 'a' || ty == 'c' || ty == 'C') { d.push_back('A'); d.push_back((uint8_t)a[5]); }
            else if (ty == 'i' || ty == 'I') {
                long long x = atoll(a.c_str() + 5);
                if (x < 0) {
                    if (x >= -127) { d.push_back('c'); d.push_back((uint8_t)(int8_t)x); }
                    else if (x >= -32767) { d.push_back('s'); int16_t v = (int16_t)x; uint8_t b[2]; memcpy(b, &v, 2); d.insert(d.end(), b, b + 2); }
                    else { d.push_back('i'); put32(d, (uint32_t)(int32_t)x); }
                } else {
                    if (x <= 255) { d.push_back('C'); d.push_back((uint8_t)x); }
                    else if (x <= 65535) { d.push_back('S'); uint16_t v = (uint16_t)x; uint8_t b[2]; memcpy(b, &v, 2); d.insert(d.end(), b, b + 2); }
                    else { d.push_back('I'); put32(d, (uint32_t)x); }
                }
            } else if (ty == 'Z' || ty == 'H') { d.push_back((uint8_t)ty); d.insert(d.end(), a.begin() + 5, a.end()); d.push_back(0); }
        }
        uint32_t bs = (uint32_t)(d.size() - at - 4);
        memcpy(d.data() + at, &bs, 4);
    }
    // Records already encoded (by encode()) elsewhere: bytes = the records back to back, size[i] / rid[i] = byte count and
    // read id (atol(qname)) of record i.  Shard workers encode in parallel; one thread appends in read order.
    struct Encoded { std::vector<uint8_t> bytes; std::vector<uint32_t> size; std::vector<long> rid; };
    void write_encoded(const Encoded& e) {
        std::vector<uint8_t> stream;
        stream.reserve(carry_.size() + e.bytes.size());
        stream.insert(stream.end(), carry_.begin(), carry_.end());
        stream.insert(stream.end(), e.bytes.begin(), e.bytes.end());
        std::vector<Blk> tab = flush_blocks(stream, e.size, false);
        index_lines(tab, carry_.size(), e.size, e.rid);
        const Blk& open = tab.back();
        carry_.assign(stream.begin() + (ptrdiff_t)open.ustart, stream.end());
    }
    // The same in three steps, so that shard workers deflate their own members and the ordered writer only appends:
    //   plan      where the members end depends on the bytes still open from the batch before (`carry`) and on record sizes alone, so
    //             batches are planned one after the other as soon as they are ENCODED (cheap, the caller serialises it), not written;
    //   compress  the batch's members, by whoever holds the batch, in parallel with other batches;
    //   commit    in order: members to the file, `.index` lines on the now-known addresses.
    // Should a member not fit its envelope (incompressible data: bgzf.c then shrinks the block and every later cut moves) the
    // writer drops the plans from that batch on and goes back to write_encoded.
    struct Prepared {
        std::vector<uint8_t> stream;                 // carry + the batch's records
        size_t carry_in = 0;
        std::vector<uint32_t> size; std::vector<long> rid;
        std::vector<size_t> cuts;                    // member ends in `stream`
        std::vector<std::vector<uint8_t>> members;
        bool ok = true;
        // A batch whose records were encoded AND deflated on the device (thj_span_bam_encode, thj_bgzf_deflate): `members` come
        // ready, `stream` holds at most the carry (the records' bytes never reach the host).  Such a batch starts a member and
        // closes its last one at its end, like a bgzf_flush on either side: the BAM stream inside is the same, members are cut at
        // the batch's ends in addition to where bam_write1 cuts them.
        bool device = false;
        bool carry_member = false;                   // members[0] is the carry of the batch before, closed by plan_device
    };
    // BGZF envelope round clen bytes of DEFLATE data made elsewhere (crc / isize: of the member's uncompressed bytes)
    static void wrap_member(const uint8_t* cdata, size_t clen, uint32_t crc, uint32_t isize, std::vector<uint8_t>& out) {
        static const uint8_t hdr[12] = {31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0};
        out.resize(clen + 26);
        memcpy(out.data(), hdr, 12);
        out[12] = 'B'; out[13] = 'C'; out[14] = 2; out[15] = 0;
        const uint16_t bs = (uint16_t)(clen + 18 + 8 - 1);
        memcpy(out.data() + 16, &bs, 2);
        memcpy(out.data() + 18, cdata, clen);
        memcpy(out.data() + 18 + clen, &crc, 4);
        memcpy(out.data() + 18 + clen + 4, &isize, 4);
    }
    // member ends for a batch that starts a member and ends one (the device batches): bam_write1's rule inside, a cut at the end
    static void plan_cuts_closed(const std::vector<uint32_t>& size, std::vector<size_t>& cuts) {
        size_t end_off = 0, total = 0;
        for (auto s : size) total += s;
        plan_cuts(0, 0, size, 0, cuts, &end_off);
        if (total && (cuts.empty() || cuts.back() != total)) cuts.push_back(total);
    }
    // the planner's step for a device batch (p.device, p.size / p.rid / p.cuts / p.members filled, cuts counted from the batch's
    // first byte): what is still open from the batch before becomes a member of its own, deflated by compress()
    static void plan_device(std::vector<uint8_t>& carry, Prepared& p) {
        p.carry_in = carry.size();
        if (!p.carry_in) return;
        p.stream = std::move(carry); carry.clear();
        for (auto& c : p.cuts) c += p.carry_in;
        p.cuts.insert(p.cuts.begin(), p.carry_in);
        p.members.insert(p.members.begin(), std::vector<uint8_t>());
        p.carry_member = true;
    }
    static void plan(std::vector<uint8_t>& carry, Encoded&& e, Prepared& p) {
        p.carry_in = carry.size();
        p.stream.reserve(carry.size() + e.bytes.size());
        p.stream.insert(p.stream.end(), carry.begin(), carry.end());
        p.stream.insert(p.stream.end(), e.bytes.begin(), e.bytes.end());
        std::vector<uint8_t>().swap(e.bytes);
        p.size = std::move(e.size); p.rid = std::move(e.rid);
        size_t end_off = 0;
        plan_cuts(p.carry_in, p.carry_in, p.size, 0, p.cuts, &end_off);
        const size_t last = p.cuts.empty() ? 0 : p.cuts.back();
        carry.assign(p.stream.begin() + (ptrdiff_t)last, p.stream.end());
    }
    static void compress(Prepared& p) {
        if (p.device) {
            if (p.carry_member && !deflate_member(p.stream.data(), p.carry_in, p.members[0])) p.ok = false;
            return;
        }
        p.members.resize(p.cuts.size());
        for (size_t k = 0; k < p.cuts.size() && p.ok; ++k) {
            const size_t a = k ? p.cuts[k - 1] : 0;
            if (!deflate_member(p.stream.data() + a, p.cuts[k] - a, p.members[k])) p.ok = false;
        }
    }
    void commit(Prepared& p) {
        if (!p.ok) replay_ = true;
        if (p.device) {
            // In replay mode (a member of an earlier batch did not fit its envelope, so the cuts planned since then are void) what
            // is open goes out here with bgzf.c's shrink rule and the planned carry member is dropped; the batch's own members do
            // not depend on anything before them.
            size_t first = 0, shift = 0;
            if (replay_) {
                if (!carry_.empty()) { std::vector<uint8_t> st = carry_; flush_blocks(st, std::vector<uint32_t>(), true); }
                first = p.carry_member ? 1 : 0; shift = p.carry_in;
            }
            std::vector<Blk> tab;
            size_t upos = 0;
            for (size_t k = first; k < p.cuts.size(); ++k) {
                const size_t end = p.cuts[k] - shift;
                tab.push_back({upos, end - upos, file_addr_});
                write_member(p.members[k]);
                upos = end;
            }
            tab.push_back({upos, 0, file_addr_});
            index_lines(tab, p.carry_in - shift, p.size, p.rid);
            carry_.clear();
            return;
        }
        if (replay_) {
            Encoded e;
            e.bytes.assign(p.stream.begin() + (ptrdiff_t)p.carry_in, p.stream.end());
            e.size = std::move(p.size); e.rid = std::move(p.rid);
            write_encoded(e);
            return;
        }
        std::vector<Blk> tab;
        size_t upos = 0;
        for (size_t k = 0; k < p.cuts.size(); ++k) {
            tab.push_back({upos, p.cuts[k] - upos, file_addr_});
            write_member(p.members[k]);
            upos = p.cuts[k];
        }
        tab.push_back({upos, p.stream.size() - upos, file_addr_});       // the open block
        index_lines(tab, p.carry_in, p.size, p.rid);
        carry_.assign(p.stream.begin() + (ptrdiff_t)upos, p.stream.end());
    }
private:
    bool replay_ = false;
    // GBamWriter::write(b, read_id): index line once >= INDEX_REC_COUNT (1000) records have passed and the id changes
    void index_lines(const std::vector<Blk>& tab, size_t carry_len, const std::vector<uint32_t>& size, const std::vector<long>& rid) {
        if (!idx_) return;
        size_t x = carry_len;
        for (size_t i = 0; i < size.size(); ++i) {
            const size_t s0 = x, e0 = x + size[i];
            x = e0;
            const long read_id = rid[i];
            if (!read_id) continue;
            bool widx = idxcount_ >= 1000 && read_id != last_id_;
            last_id_ = read_id; ++idxcount_;
            if (!widx) continue;
            int64_t pre_pos = tell_at(tab, s0), pre_addr = (pre_pos >> 16) & 0xFFFFFFFFFFFFLL;
            int64_t off = tell_at(tab, e0);
            int post_offs = (int)(off & 0xFFFF); int64_t post_addr = (off >> 16) & 0xFFFFFFFFFFFFLL;
            int data_len = (int)size[i] - 4;       // b->data_len + BAM_CORE_SIZE == block_size of the record
            if (post_addr != pre_addr && post_offs >= data_len) pre_pos = post_addr << 16;
            fprintf(idx_, "%ld\t%ld\n", read_id, (long)pre_pos);
            idxcount_ = 0;
        }
    }
public:
    // Writes records 0..n-1 in order.  enc(i, bytes) appends record i with encode() and returns its read id (atol(qname)).
    void write_records(size_t n, const std::function<long(size_t, std::vector<uint8_t>&)>& enc) {
        int T = host_threads();
        if ((size_t)T > n / 256 + 1) T = (int)(n / 256 + 1);
        std::vector<std::vector<uint8_t>> part((size_t)T);
        Encoded e;
        e.size.resize(n); e.rid.resize(n);
        auto work = [&](int t) {
            const size_t a = n * (size_t)t / (size_t)T, b = n * (size_t)(t + 1) / (size_t)T;
            std::vector<uint8_t>& d = part[(size_t)t];
            d.reserve((b - a) * 256);
            for (size_t i = a; i < b; ++i) { size_t before = d.size(); e.rid[i] = enc(i, d); e.size[i] = (uint32_t)(d.size() - before); }
        };
        if (T > 1) { std::vector<std::thread> th; for (int t = 0; t < T; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
        else work(0);
        size_t total = 0;
        for (auto& d : part) total += d.size();
        e.bytes.reserve(total);
        for (auto& d : part) { e.bytes.insert(e.bytes.end(), d.begin(), d.end()); std::vector<uint8_t>().swap(d); }
        write_encoded(e);
    }
    void close() {
        if (!f_) return;
        if (!carry_.empty()) { std::vector<uint8_t> s = carry_; flush_blocks(s, std::vector<uint32_t>(), true); carry_.clear(); }
        std::vector<uint8_t> eof;
        deflate_member(nullptr, 0, eof);              // an empty member: the BGZF EOF marker block
        fwrite(eof.data(), 1, eof.size(), f_);
        close_output(f_, "the BAM output"); f_ = nullptr;
        if (idx_) { close_output(idx_, "the BAM output's .index"); idx_ = nullptr; }
    }
    ~BamWriter() { close(); }
};

inline void reverse_complement(std::string& s) {       // reads.cpp:189-207
    for (auto& c : s) { switch (c) { case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break; default: c = 'N'; } }
    std::reverse(s.begin(), s.end());
}

}  // namespace thjh
