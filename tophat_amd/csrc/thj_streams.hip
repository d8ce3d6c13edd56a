// thj_streams.hip -- the context's side streams (thj_ctx.h: aux_stream), chosen by measurement.  A translation unit of its own: the
// spin kernel's code object is a few hundred bytes, and a process that never runs stage 1 (long_spanning_reads) does not load
// thj_segjuncs.hip's for it (that was 45-65 ms of a context's first stitch when the function lived there: THJ_TRACE).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../../include/thj.h"
#include "thj_ctx.h"

// HIP hands a new stream the next of GPU_MAX_HW_QUEUES (4) hardware queues, round robin over every stream the PROCESS ever made -- and
// kernels of two streams on one queue leave one after the other.  Which queue a stream is on cannot be asked, so it is measured: a
// candidate stream runs a 150 us spin kernel beside one on the context's stream and on each side stream taken so far; when they all end
// in the time of one, the candidate is on a queue of its own and is taken, otherwise it is set aside (not destroyed before the search is
// over: its queue would be the next one handed out) and another is made.  (bench.py's files-in -> files-out leg opens a context of its own
// for the inflater's figure before the resident-data steps; the streams it made moved the round robin on, the second side's stream landed
// on the context's queue, and the default line's steps were 6.5 ms where a bare run's were 5.55: profiles/r05_default_slow.txt.)
// Streams are made as they are needed -- `need` = 1 for a single batch, 2 for a pair call, 3 behind developer switches: making one costs
// ~10 ms (a hardware queue is set up), a context's first stitch in long_spanning_reads waits for it.
__global__ void thj_k_spin(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
static int spin_group_us(thj_ctx* c, const hipStream_t* st, int n, double* us) {        // a spin kernel on the context's stream and on st[0 .. n)
    double best = 1e30;
    for (int rep = 0; rep < 2; ++rep) {
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < n; ++k) HIPCHK(hipStreamSynchronize(st[k]));
        const auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(thj_k_spin, dim3(1), dim3(64), 0, c->stream, 15000ull);          // 150 us at the 100 MHz of s_memrealtime
        for (int k = 0; k < n; ++k) hipLaunchKernelGGL(thj_k_spin, dim3(1), dim3(64), 0, st[k], 15000ull);
        HIPCHK(hipStreamSynchronize(c->stream));
        for (int k = 0; k < n; ++k) HIPCHK(hipStreamSynchronize(st[k]));
        const double dt = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (dt < best) best = dt;
    }
    *us = best;
    return THJ_OK;
}
int thj_ensure_aux_streams(thj_ctx* c, int need) {
    if (need > 3) need = 3;
    int have = 0;
    while (have < 3 && c->aux_stream[have]) ++have;
    if (!c->aux_ev[0]) for (int i = 0; i < 10; ++i) HIPCHK(hipEventCreateWithFlags(&c->aux_ev[i], hipEventDisableTiming));
    if (have >= need) return THJ_OK;
    // (THJ_SJ_PRIO=1: the side streams at the highest priority the device has -- measured worse, 7.0 against 6.6 ms per step: the flat reads'
    // rescue scan then waits for them)
    int lo = 0, hi = 0;
    static const bool prio = getenv("THJ_SJ_PRIO") && atoi(getenv("THJ_SJ_PRIO")) != 0;
    static const bool no_probe = getenv("THJ_NO_QUEUE_PROBE") != nullptr;                       // developer switch: the streams as they come
    static const bool trace = getenv("THJ_TRACE") != nullptr;
    if (prio) (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    double alone = 0;
    if (!no_probe) { const int rc0 = spin_group_us(c, nullptr, 0, &alone); if (rc0) return rc0; }
    hipStream_t aside[4]; int n_aside = 0;
    int rc = THJ_OK;
    while (have < need && rc == THJ_OK) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, prio ? hi : 0) != hipSuccess) { thj_set_error("hipStreamCreate failed"); rc = THJ_EHIP; break; }
        // (four set aside: every queue is shared with something -- the next one as it is, flagged and warned about below)
        const bool forced = no_probe || n_aside == 4;
        double us = 0;
        hipStream_t grp[3]; int n = 0;
        for (int k = 0; k < have; ++k) grp[n++] = c->aux_stream[k];
        grp[n++] = cand;
        if (no_probe && alone == 0) rc = spin_group_us(c, nullptr, 0, &alone);      // THJ_NO_QUEUE_PROBE: the stream is taken as it comes, but what it shares is still measured
        if (rc == THJ_OK) rc = spin_group_us(c, grp, n, &us);
        if (rc != THJ_OK) { (void)hipStreamDestroy(cand); break; }
        const bool indep = us < 1.5 * alone;
        if (indep || forced) {
            c->aux_ratio[have] = alone > 0 ? us / alone : 0.0; c->aux_independent[have] = indep;
            c->aux_stream[have++] = cand;
            if (!indep) fprintf(stderr, "thj: warning: side stream %d shares a hardware queue with another stream of this context (a spin kernel beside the others %.0f us, alone %.0f): "
                                        "the two sides of a pass will partly run one after the other (same results, slower)%s\n", have, us, alone, no_probe ? " [THJ_NO_QUEUE_PROBE]" : "");
            if (trace) fprintf(stderr, "[streams] side stream %d: beside the others %.0f us (a spin kernel alone %.0f), %d set aside\n", have, us, alone, n_aside);
        }
        else aside[n_aside++] = cand;
    }
    for (int k = 0; k < n_aside; ++k) (void)hipStreamDestroy(aside[k]);
    return rc;
}

// What the probe found (the placement is asserted, not trusted): the side streams in use, and for each
// whether it was measured to run beside the context's stream and the side streams before it.
extern "C" int thj_ctx_stream_info(thj_ctx* c, int32_t* n_side, int32_t* independent /* [3] */, double* ratio /* [3] */) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    int have = 0;
    while (have < 3 && c->aux_stream[have]) ++have;
    if (n_side) *n_side = have;
    for (int k = 0; k < 3; ++k) {
        if (independent) independent[k] = k < have && c->aux_independent[k] ? 1 : 0;
        if (ratio) ratio[k] = k < have ? c->aux_ratio[k] : 0.0;
    }
    return THJ_OK;
}
extern "C" int thj_ctx_probe_streams(thj_ctx* c, int32_t need) {
    if (!c) { thj_set_error("null ctx"); return THJ_EINVAL; }
    if (need < 1 || need > 2) { thj_set_error("thj_ctx_probe_streams: need must be 1 or 2"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    return thj_ensure_aux_streams(c, need);
}
