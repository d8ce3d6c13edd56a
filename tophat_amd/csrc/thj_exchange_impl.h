// thj_exchange_impl.h -- the ONE exchange step of a read-sharded run, inside the C ABI:
// every rank (one GPU each) ran segment_juncs' finders over its own reads; before the sets are sorted and handed to
// long_spanning_reads they become the union over all ranks, which is what segment_juncs.cpp:4911-4922 does with its
// per-thread sets (seg_juncs.insert / deletions.insert / insertions.insert in thread order, merge_with for fusions).
//
//   pack      thj_k_xpack: the dense lists of distinct events (see set_insert) -> one fixed-size message
//             [8 header words | junction keys | deletion keys | insertion keys | insertion values]
//   move      ONE ncclAllGather (RCCL over xGMI) on the context's stream
//   merge     thj_k_xmerge: the other ranks' keys go through the same idempotent set_insert / atomicMin inserts the
//             finders use
// No host synchronisation anywhere: counts travel in the message header, message capacities are fixed per
// communicator, and thj_segjuncs_finish -- which waits for the stream anyway -- looks at the gathered headers and
// repeats the step with larger capacities in the rare case one was too small (every rank sees the same headers, so
// every rank takes the same decision; re-inserting keys is harmless).
//
// Included at the end of thj_segjuncs.hip (needs Tables, set_insert, grow_tables).
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>          // types and enums only: librccl.so.1 (573 MB) is dlopen'ed when the first communicator is made

#include <condition_variable>
#include <mutex>

namespace xch {

struct Api {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

static Api* rccl() {
    static Api api;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
#define THJ_SYM(name) api.name = (decltype(api.name))dlsym(h, "nccl" #name)
        THJ_SYM(GetUniqueId); THJ_SYM(CommInitRank); THJ_SYM(CommDestroy); THJ_SYM(AllGather); THJ_SYM(AllReduce);
        THJ_SYM(GroupStart); THJ_SYM(GroupEnd); THJ_SYM(GetErrorString);
#undef THJ_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.AllGather && api.AllReduce && api.GroupStart &&
                 api.GroupEnd && api.GetErrorString;
    });
    return api.ok ? &api : nullptr;
}

// Contexts of ONE process that share a device cannot form an RCCL communicator (one rank per GPU).  For them -- the
// single-GPU functional test of the N > 1 path -- the all-gather is done with stream-ordered device copies between the
// ranks' buffers; the host threads only meet to publish pointers and events, never to wait for the GPU.
struct LoopGroup {
    std::mutex mu; std::condition_variable cv;
    int n = 0, arrived = 0, refs = 0; uint64_t gen = 0;
    std::vector<const void*> send; std::vector<hipEvent_t> ready, done;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t g = gen;
        if (++arrived == n) { arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

}  // namespace xch

struct thj_comm {
    thj_ctx* ctx = nullptr;
    int n = 1, rank = 0;
    ncclComm_t nc = nullptr;
    xch::LoopGroup* loop = nullptr;
    // event message: capacities (keys) of the three sections -- equal on every rank by construction: they start from the
    // same defaults and only change as a function of the gathered headers
    u64 cap_j = 1ull << 19, cap_d = 1ull << 14, cap_i = 1ull << 14;
    u64 *d_send = nullptr, *d_recv = nullptr, *d_hdr = nullptr;
    u64 alloc_words = 0;
    unsigned long long* h_hdr = nullptr;        // pinned, 8 words per rank
    int64_t rounds = 0, redo_rounds = 0;        // exchange steps enqueued / of them repeats with larger capacities
};

#define NCCLCHK(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t r__ = (expr);                                                                      \
        if (r__ != ncclSuccess) {                                                                       \
            thj_set_error("%s: %s (%s:%d)", #expr, xch::rccl()->GetErrorString(r__), __FILE__, __LINE__); \
            return THJ_EHIP;                                                                            \
        }                                                                                               \
    } while (0)

static inline u64 x_words(const thj_comm* m) { return 8 + m->cap_j + m->cap_d + 2 * m->cap_i; }

static int x_alloc(thj_comm* m) {
    const u64 w = x_words(m);
    if (w <= m->alloc_words) return THJ_OK;
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    hipFree(m->d_send); hipFree(m->d_recv);
    m->d_send = m->d_recv = nullptr; m->alloc_words = 0;
    HIPCHK(hipMalloc(&m->d_send, (size_t)w * 8));
    HIPCHK(hipMalloc(&m->d_recv, (size_t)w * 8 * (size_t)m->n));
    m->alloc_words = w;
    return THJ_OK;
}

static int comm_finish_init(thj_comm* m) {
    HIPCHK(hipSetDevice(m->ctx->device));
    HIPCHK(hipMalloc(&m->d_hdr, (size_t)m->n * 8 * 8));
    HIPCHK(hipHostMalloc(&m->h_hdr, (size_t)m->n * 8 * 8));
    if (getenv("THJ_XCHG_CAPS")) {              // test knob: tiny message sections, so that the grow-and-repeat path runs
        unsigned long long a = 0, b = 0, c2 = 0;
        if (sscanf(getenv("THJ_XCHG_CAPS"), "%llu,%llu,%llu", &a, &b, &c2) == 3 && a && b && c2) { m->cap_j = a; m->cap_d = b; m->cap_i = c2; }
    }
    return THJ_OK;
}

// stream-ordered all-gather of `bytes` bytes per rank: recv[r * bytes ..] = rank r's send
static int comm_allgather(thj_comm* m, const void* send, void* recv, size_t bytes) {
    hipStream_t s = m->ctx->stream;
    if (m->nc) {
        NCCLCHK(xch::rccl()->AllGather(send, recv, bytes, ncclUint8, m->nc, s));
        return THJ_OK;
    }
    if (!m->loop) {                              // a communicator of one
        if (bytes) HIPCHK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
        return THJ_OK;
    }
    xch::LoopGroup* g = m->loop;
    HIPCHK(hipEventRecord(g->ready[(size_t)m->rank], s));
    g->send[(size_t)m->rank] = send;
    g->barrier();
    for (int p = 0; p < m->n; ++p) {
        HIPCHK(hipStreamWaitEvent(s, g->ready[(size_t)p], 0));
        if (bytes) HIPCHK(hipMemcpyAsync((char*)recv + (size_t)p * bytes, g->send[(size_t)p], bytes, hipMemcpyDefault, s));
    }
    HIPCHK(hipEventRecord(g->done[(size_t)m->rank], s));
    g->barrier();
    for (int p = 0; p < m->n; ++p) HIPCHK(hipStreamWaitEvent(s, g->done[(size_t)p], 0));    // nobody overwrites a buffer a peer still reads
    return THJ_OK;
}

extern "C" int thj_comm_unique_id(uint8_t* id /*[128]*/) {
    if (!id) { thj_set_error("thj_comm_unique_id: null argument"); return THJ_EINVAL; }
    xch::Api* a = xch::rccl();
    if (!a) { thj_set_error("librccl.so.1 could not be loaded: %s", dlerror()); return THJ_ESTATE; }
    static_assert(sizeof(ncclUniqueId) == THJ_COMM_ID_BYTES, "unique id size");
    ncclUniqueId u;
    NCCLCHK(a->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return THJ_OK;
}

extern "C" int thj_comm_create(thj_ctx* c, const uint8_t* id, int32_t n_ranks, int32_t rank, thj_comm** out) {
    if (!c || !out || n_ranks < 1 || rank < 0 || rank >= n_ranks || (n_ranks > 1 && !id)) { thj_set_error("thj_comm_create: bad argument"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    thj_comm* m = new thj_comm();
    m->ctx = c; m->n = n_ranks; m->rank = rank;
    if (id) {                                    // also at n_ranks == 1: the collective then really goes through RCCL
        xch::Api* a = xch::rccl();
        if (!a) { delete m; thj_set_error("librccl.so.1 could not be loaded: %s", dlerror()); return THJ_ESTATE; }
        ncclUniqueId u;
        memcpy(&u, id, sizeof u);
        ncclResult_t r = a->CommInitRank(&m->nc, n_ranks, u, rank);
        if (r != ncclSuccess) { thj_set_error("ncclCommInitRank: %s", a->GetErrorString(r)); delete m; return THJ_EHIP; }
    }
    int rc = comm_finish_init(m);
    if (rc) { delete m; return rc; }
    *out = m;
    return THJ_OK;
}

extern "C" int thj_comm_create_local(thj_ctx* const* ctxs, int32_t n, thj_comm** out) {
    if (!ctxs || !out || n < 1) { thj_set_error("thj_comm_create_local: bad argument"); return THJ_EINVAL; }
    bool distinct = true;
    for (int i = 0; i < n; ++i) { if (!ctxs[i]) { thj_set_error("thj_comm_create_local: null context"); return THJ_EINVAL; }
                                  for (int k = 0; k < i; ++k) if (ctxs[k]->device == ctxs[i]->device) distinct = false; }
    std::vector<thj_comm*> ms((size_t)n, nullptr);
    for (int i = 0; i < n; ++i) { ms[(size_t)i] = new thj_comm(); ms[(size_t)i]->ctx = ctxs[i]; ms[(size_t)i]->n = n; ms[(size_t)i]->rank = i; }
    auto fail = [&](int rc) { for (auto* m : ms) delete m; return rc; };
    if (n > 1 && distinct) {                    // one rank per GPU: RCCL, all ranks initialised by this thread as one group
        xch::Api* a = xch::rccl();
        if (!a) { thj_set_error("librccl.so.1 could not be loaded: %s", dlerror()); return fail(THJ_ESTATE); }
        ncclUniqueId u;
        ncclResult_t r = a->GetUniqueId(&u);
        if (r == ncclSuccess) r = a->GroupStart();
        for (int i = 0; i < n && r == ncclSuccess; ++i) {
            if (hipSetDevice(ctxs[i]->device) != hipSuccess) { thj_set_error("hipSetDevice(%d) failed", ctxs[i]->device); return fail(THJ_EHIP); }
            r = a->CommInitRank(&ms[(size_t)i]->nc, n, u, i);
        }
        if (r == ncclSuccess) r = a->GroupEnd();
        if (r != ncclSuccess) { thj_set_error("RCCL communicator over %d GPUs: %s", n, a->GetErrorString(r)); return fail(THJ_EHIP); }
    } else if (n > 1) {
        xch::LoopGroup* g = new xch::LoopGroup();
        g->n = n; g->refs = n; g->send.assign((size_t)n, nullptr); g->ready.resize((size_t)n); g->done.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            hipSetDevice(ctxs[i]->device);
            hipEventCreateWithFlags(&g->ready[(size_t)i], hipEventDisableTiming);
            hipEventCreateWithFlags(&g->done[(size_t)i], hipEventDisableTiming);
            ms[(size_t)i]->loop = g;
        }
    }
    for (int i = 0; i < n; ++i) { int rc = comm_finish_init(ms[(size_t)i]); if (rc) return fail(rc); }
    for (int i = 0; i < n; ++i) out[i] = ms[(size_t)i];
    return THJ_OK;
}

extern "C" void thj_comm_destroy(thj_comm* m) {
    if (!m) return;
    hipSetDevice(m->ctx->device);
    hipStreamSynchronize(m->ctx->stream);
    if (m->ctx->xchg == m) m->ctx->xchg = nullptr;
    if (m->nc) xch::rccl()->CommDestroy(m->nc);
    if (m->loop) {
        bool last;
        { std::lock_guard<std::mutex> lk(m->loop->mu); last = --m->loop->refs == 0; }
        if (last) { for (auto e : m->loop->ready) hipEventDestroy(e); for (auto e : m->loop->done) hipEventDestroy(e); delete m->loop; }
    }
    hipFree(m->d_send); hipFree(m->d_recv); hipFree(m->d_hdr);
    if (m->h_hdr) hipHostFree(m->h_hdr);
    delete m;
}

extern "C" int thj_comm_info(const thj_comm* m, int32_t* n_ranks, int32_t* rank, int32_t* transport, int64_t* stats /*[4]*/) {
    if (!m) { thj_set_error("null comm"); return THJ_EINVAL; }
    if (n_ranks) *n_ranks = m->n;
    if (rank) *rank = m->rank;
    if (transport) *transport = m->nc ? THJ_COMM_RCCL : (m->loop ? THJ_COMM_LOOPBACK : THJ_COMM_SELF);
    if (stats) { stats[0] = m->rounds; stats[1] = m->redo_rounds; stats[2] = (int64_t)(x_words(m) * 8); stats[3] = (int64_t)m->cap_j; }
    return THJ_OK;
}

// ------------------------------------------------------------------ event sets

__global__ __launch_bounds__(256) void thj_k_xpack(Tables t, u64* send, u64 cap_j, u64 cap_d, u64 cap_i) {
    const u64 nj = t.cnt[CNT_JUNC], nd = t.cnt[CNT_DEL], ni = t.cnt[CNT_INS];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        send[0] = nj; send[1] = nd; send[2] = ni; send[3] = (u64)(t.ovf[0] | t.ovf[1] | t.ovf[2] | t.ovf[3]);      // ([3]: a full task list -- events of this rank are missing: every rank must know)
        send[4] = cap_j; send[5] = cap_d; send[6] = cap_i; send[7] = 0;
    }
    u64* pj = send + 8; u64* pd = pj + cap_j; u64* pk = pd + cap_d; u64* pv = pk + cap_i;
    const u64 tj = nj < cap_j ? nj : cap_j, td = nd < cap_d ? nd : cap_d, ti = ni < cap_i ? ni : cap_i;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < tj + td + ti; i += (u64)gridDim.x * blockDim.x) {
        if (i < tj) pj[i] = t.junc_list[i];
        else if (i < tj + td) pd[i - tj] = t.del_list[i - tj];
        else { const u64 k = i - tj - td, h = t.ins_list[k]; pk[k] = t.ins_key[h]; pv[k] = t.ins_val[h]; }
    }
}

// blockIdx.y = source rank.  Every block of a rank's column copies nothing but inserts its share of that rank's keys.
__global__ __launch_bounds__(256) void thj_k_xmerge(Tables t, const u64* recv, u64* hdr_out, int me, u64 words, u64 cap_j, u64 cap_d, u64 cap_i) {
    const int r = (int)blockIdx.y;
    const u64* src = recv + (u64)r * words;
    if (blockIdx.x == 0 && threadIdx.x < 8) hdr_out[r * 8 + threadIdx.x] = src[threadIdx.x];
    if (r == me) return;
    const u64 nj = src[0], nd = src[1], ni = src[2];
    const u64 tj = nj < cap_j ? nj : cap_j, td = nd < cap_d ? nd : cap_d, ti = ni < cap_i ? ni : cap_i;
    const u64* pj = src + 8; const u64* pd = pj + cap_j; const u64* pk = pd + cap_d; const u64* pv = pk + cap_i;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < tj + td + ti; i += (u64)gridDim.x * blockDim.x) {
        if (i < tj) set_insert(t.junc, t.junc_mask, pj[i], &t.cnt[CNT_JUNC], &t.ovf[0], t.junc_list);
        else if (i < tj + td) set_insert(t.del, t.del_mask, pd[i - tj], &t.cnt[CNT_DEL], &t.ovf[1], t.del_list);
        else { const u64 k = i - tj - td; map_insert_min(t.ins_key, t.ins_val, t.ins_mask, pk[k], pv[k], &t.cnt[CNT_INS], &t.ovf[2], t.ins_list); }
    }
}

static inline Tables ctx_tables(thj_ctx* c) {
    return Tables{c->d_junc, (u64)c->junc_cap - 1, c->d_del, (u64)c->indel_cap - 1, c->d_ins_key, c->d_ins_val,
                  (u64)c->indel_cap - 1, junc_list(c), del_list(c), ins_list(c), c->d_ovf, c->d_cnt};
}

static int x_merge_launch(thj_ctx* c, thj_comm* m) {
    const u64 per = m->cap_j + m->cap_d + m->cap_i;
    unsigned bx = (unsigned)((per + 255) / 256); if (bx > 512) bx = 512; if (bx < 1) bx = 1;
    hipLaunchKernelGGL(thj_k_xmerge, dim3(bx, (unsigned)m->n), dim3(256), 0, c->stream, ctx_tables(c), (const u64*)m->d_recv, m->d_hdr,
                       m->rank, x_words(m), m->cap_j, m->cap_d, m->cap_i);
    HIPCHK(hipGetLastError());
    return THJ_OK;
}

static int x_enqueue(thj_ctx* c, thj_comm* m) {
    int rc = x_alloc(m);
    if (rc) return rc;
    const u64 per = m->cap_j + m->cap_d + m->cap_i;
    unsigned bx = (unsigned)((per + 255) / 256); if (bx > 1024) bx = 1024; if (bx < 1) bx = 1;
    hipLaunchKernelGGL(thj_k_xpack, dim3(bx), dim3(256), 0, c->stream, ctx_tables(c), m->d_send, m->cap_j, m->cap_d, m->cap_i);
    HIPCHK(hipGetLastError());
    if ((rc = comm_allgather(m, m->d_send, m->d_recv, (size_t)x_words(m) * 8))) return rc;
    if ((rc = x_merge_launch(c, m))) return rc;
    ++m->rounds;
    return THJ_OK;
}

extern "C" int thj_events_allgather_async(thj_ctx* c, thj_comm* m) {
    if (!c || !m || m->ctx != c) { thj_set_error("thj_events_allgather_async: the communicator does not belong to this context"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = x_enqueue(c, m);
    if (rc) return rc;
    c->xchg = m;                                   // thj_segjuncs_finish checks the gathered headers
    return THJ_OK;
}

// Called by thj_segjuncs_finish after the stream has been synchronised and the gathered headers copied to m->h_hdr.
// Returns 1 when more work was enqueued (the caller synchronises and looks again), 0 when the exchange is complete.
static int x_finish_check(thj_ctx* c, const unsigned int* ovf_now) {
    thj_comm* m = c->xchg;
    u64 mj = 0, md = 0, mi = 0; bool pre_ovf = false;
    for (int r = 0; r < m->n; ++r) {
        const unsigned long long* h = m->h_hdr + (size_t)r * 8;
        if (h[0] > mj) mj = h[0];
        if (h[1] > md) md = h[1];
        if (h[2] > mi) mi = h[2];
        if (h[3]) pre_ovf = true;
    }
    if (pre_ovf) {
        c->xchg = nullptr;
        thj_set_error("event table or task list overflow on a rank before the exchange step: call thj_segjuncs_configure with larger capacities / split the batch and re-run");
        return THJ_EOVERFLOW;
    }
    if (mj > m->cap_j || md > m->cap_d || mi > m->cap_i) {       // a message section was too small somewhere: same verdict on every rank
        auto grow = [](u64 cap, u64 need) { while (cap < 2 * need) cap <<= 1; return cap; };
        if (mj > m->cap_j) m->cap_j = grow(m->cap_j, mj);
        if (md > m->cap_d) m->cap_d = grow(m->cap_d, md);
        if (mi > m->cap_i) m->cap_i = grow(m->cap_i, mi);
        ++m->redo_rounds;
        int rc = x_enqueue(c, m);
        return rc ? rc : 1;
    }
    if (ovf_now[0] || ovf_now[1] || ovf_now[2]) {                 // this rank's table filled up while merging: nothing is lost, the
        // gathered keys are still in d_recv.  The headers say how many keys every rank sent: the tables grow in one go to where their sum
        // (an upper bound of the distinct keys) is at most 40 % of the slots, not by one factor of four per failed merge
        u64 sj = 0, sd = 0, si = 0;
        for (int r = 0; r < m->n; ++r) { const unsigned long long* h = m->h_hdr + (size_t)r * 8; sj += h[0]; sd += h[1]; si += h[2]; }
        const u64 sdi = sd > si ? sd : si;
        int rc = THJ_OK;
        const bool gj = ovf_now[0] != 0, gdi = (ovf_now[1] | ovf_now[2]) != 0;
        do { rc = grow_tables(c, gj, gdi); } while (rc == THJ_OK && ((gj && (u64)c->junc_cap * 2 < sj * 5 && c->junc_cap < (1ll << 34)) || (gdi && (u64)c->indel_cap * 2 < sdi * 5 && c->indel_cap < (1ll << 34))));
        if (rc) return rc;
        if (getenv("THJ_TIMING")) fprintf(stderr, "[exchange] a table filled up while the ranks' keys were merged (%llu junctions, %llu deletions, %llu insertions gathered): grown to %lld / %lld slots, merged again\n",
                                          (unsigned long long)sj, (unsigned long long)sd, (unsigned long long)si, (long long)c->junc_cap, (long long)c->indel_cap);
        HIPCHK(hipMemsetAsync(c->d_ovf, 0, 3 * sizeof(unsigned int), c->stream));      // (word 3, the task list's flag, is not the merge's to clear)
        if ((rc = x_merge_launch(c, m))) return rc;
        return 1;
    }
    c->xchg = nullptr;
    return 0;
}

// ------------------------------------------------------------------ fusions (merge_with(FusionSimpleSet&, ...), fusions.cpp:975-990)

extern "C" int thj_fusion_allgather(thj_ctx* c, thj_comm* m, int64_t* n_fusions) {
    if (!c || !m || m->ctx != c) { thj_set_error("thj_fusion_allgather: the communicator does not belong to this context"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    if (m->n == 1) { if (n_fusions) *n_fusions = c->h_fus_stale ? c->n_fus_out : (int64_t)c->h_fusions.size(); return THJ_OK; }
    { const int rc0 = thj_fusions_to_host(c); if (rc0) return rc0; }
    fusion_drop_device_set(c);                  // the merged set is made on the host below
    // thj_fusion_finish has reduced this rank's events; sizes first, then the padded sets
    u64 *d_n = nullptr, *d_all = nullptr;
    HIPCHK(hipMalloc(&d_n, 8)); HIPCHK(hipMalloc(&d_all, (size_t)m->n * 8));
    unsigned long long mine = c->h_fusions.size();
    std::vector<unsigned long long> sizes((size_t)m->n);
    HIPCHK(hipMemcpyAsync(d_n, &mine, 8, hipMemcpyHostToDevice, c->stream));
    int rc = comm_allgather(m, d_n, d_all, 8);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(sizes.data(), d_all, (size_t)m->n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(d_n); hipFree(d_all);
    unsigned long long mx = 0;
    for (auto s : sizes) if (s > mx) mx = s;
    if (mx == 0) return THJ_OK;
    thj_fusion *d_s = nullptr, *d_r = nullptr;
    HIPCHK(hipMalloc(&d_s, (size_t)mx * sizeof(thj_fusion))); HIPCHK(hipMalloc(&d_r, (size_t)mx * sizeof(thj_fusion) * (size_t)m->n));
    if (mine) HIPCHK(hipMemcpyAsync(d_s, c->h_fusions.data(), (size_t)mine * sizeof(thj_fusion), hipMemcpyHostToDevice, c->stream));
    if ((rc = comm_allgather(m, d_s, d_r, (size_t)mx * sizeof(thj_fusion)))) return rc;
    std::vector<thj_fusion> all((size_t)mx * (size_t)m->n);
    HIPCHK(hipMemcpyAsync(all.data(), d_r, all.size() * sizeof(thj_fusion), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(d_s); hipFree(d_r);
    std::vector<thj_fusion> ev;
    for (int r = 0; r < m->n; ++r) ev.insert(ev.end(), all.begin() + (ptrdiff_t)((size_t)r * mx), all.begin() + (ptrdiff_t)((size_t)r * mx + sizes[(size_t)r]));
    auto less = [](const thj_fusion& a, const thj_fusion& b) {
        if (a.ref_id1 != b.ref_id1) return a.ref_id1 < b.ref_id1;
        if (a.ref_id2 != b.ref_id2) return a.ref_id2 < b.ref_id2;
        if (a.left != b.left) return a.left < b.left;
        if (a.right != b.right) return a.right < b.right;
        return a.dir < b.dir;
    };
    std::stable_sort(ev.begin(), ev.end(), less);
    c->h_fusions.clear();
    for (auto& e : ev) {
        if (!c->h_fusions.empty() && !less(c->h_fusions.back(), e) && !less(e, c->h_fusions.back())) {
            c->h_fusions.back().count += e.count;
            if (e.edit_dist < c->h_fusions.back().edit_dist) c->h_fusions.back().edit_dist = e.edit_dist;
        } else c->h_fusions.push_back(e);
    }
    if (n_fusions) *n_fusions = (int64_t)c->h_fusions.size();
    return THJ_OK;
}

// ------------------------------------------------------------------ coverage search state
// The coverage map of the run is the OR of the ranks' maps, the per-contig extent their maximum, the extension table the
// concatenation of their entries (thj_covsearch_merge_async); afterwards every rank runs the same pairing pass.

extern "C" int thj_covsearch_allgather(thj_ctx* c, thj_comm* m) {
    if (!c || !m || m->ctx != c) { thj_set_error("thj_covsearch_allgather: the communicator does not belong to this context"); return THJ_EINVAL; }
    HIPCHK(hipSetDevice(c->device));
    int rc = cov_ensure(c);
    if (rc) return rc;
    if (m->n == 1) return THJ_OK;
    const int64_t nw = c->n_blocks, nc1 = c->n_contigs + 1;
    // extension-table sizes (one small round trip: this is not the per-batch hot loop)
    u64 *d_n = nullptr, *d_all = nullptr;
    HIPCHK(hipMalloc(&d_n, 8)); HIPCHK(hipMalloc(&d_all, (size_t)m->n * 8));
    unsigned long long mine = (unsigned long long)c->n_ext;
    std::vector<unsigned long long> sizes((size_t)m->n);
    HIPCHK(hipMemcpyAsync(d_n, &mine, 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = comm_allgather(m, d_n, d_all, 8))) return rc;
    HIPCHK(hipMemcpyAsync(sizes.data(), d_all, (size_t)m->n * 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(d_n); hipFree(d_all);
    unsigned long long mx = 0, total = 0;
    for (auto s : sizes) { if (s > mx) mx = s; total += s; }
    // one message per rank: coverage words | contig extents (padded to 8 bytes) | keys (u32, padded) | values
    const size_t b_bits = (size_t)nw * 8, b_size = (size_t)((nc1 * 4 + 7) / 8) * 8, b_keys = (size_t)((mx * 4 + 7) / 8) * 8, b_vals = (size_t)mx * 8;
    const size_t msg = b_bits + b_size + b_keys + b_vals;
    char *d_s = nullptr, *d_r = nullptr;
    HIPCHK(hipMalloc(&d_s, msg)); HIPCHK(hipMalloc(&d_r, msg * (size_t)m->n));
    HIPCHK(hipMemcpyAsync(d_s, c->d_cov, b_bits, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_s + b_bits, c->d_cov_size, (size_t)nc1 * 4, hipMemcpyDeviceToDevice, c->stream));
    if (mine) {
        HIPCHK(hipMemcpyAsync(d_s + b_bits + b_size, c->d_ext_key, (size_t)mine * 4, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(d_s + b_bits + b_size + b_keys, c->d_ext_val, (size_t)mine * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    if ((rc = comm_allgather(m, d_s, d_r, msg))) return rc;
    if ((rc = cov_reserve_ext(c, (int64_t)total))) return rc;
    for (int r = 0; r < m->n; ++r) {
        if (r == m->rank) continue;
        const char* p = d_r + (size_t)r * msg;
        if ((rc = thj_covsearch_merge_async(c, (const uint64_t*)p, (const int32_t*)(p + b_bits), (const uint32_t*)(p + b_bits + b_size),
                                            (const uint64_t*)(p + b_bits + b_size + b_keys), (int64_t)sizes[(size_t)r]))) return rc;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    hipFree(d_s); hipFree(d_r);
    return THJ_OK;
}
