// bamout_sim.cpp -- TEST-ONLY: the device side of the BAM writer compiled for the CPU: tophat_amd/csrc/thj_deflate_core.h (the
// deflater, one workgroup at a time under simt.h) and thj_bamenc_core.h (the record encoder, one call per record).
// tests/test_bamout_sim_cpu.py inflates what the first writes with zlib and parses what the second writes.  Never linked into
// libthj_hip.so.
#include "simt.h"
#include "../../tophat_amd/csrc/thj_deflate_core.h"
#include "../../tophat_amd/csrc/thj_bamenc_core.h"

#include <vector>

namespace {
struct SimX {
    simt::Block* b; int tid, lane, wave;
    uint64_t ballot(bool p) { const uint32_t* a = b->exchange(tid, p ? 1u : 0u); uint64_t m = 0; for (int i = 0; i < 64; ++i) m |= (uint64_t)(a[i] & 1u) << i; return m; }
    uint32_t shfl(uint32_t v, int src) { return b->exchange(tid, v)[src & 63]; }
    uint32_t bcast(uint32_t v, int src) { return b->exchange(tid, v)[src & 63]; }
    uint32_t incl_scan(uint32_t v) { const uint32_t* a = b->exchange(tid, v); uint32_t s = 0; for (int i = 0; i <= lane; ++i) s += a[i]; return s; }
    uint32_t wave_max(uint32_t v) { const uint32_t* a = b->exchange(tid, v); uint32_t s = 0; for (int i = 0; i < 64; ++i) s = a[i] > s ? a[i] : s; return s; }
    uint32_t wave_xor(uint32_t v) { const uint32_t* a = b->exchange(tid, v); uint32_t s = 0; for (int i = 0; i < 64; ++i) s ^= a[i]; return s; }
    void wsync() { b->exchange(tid, 0); }
    void sync() { b->barrier(); }
    uint32_t lds_add(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
    void lds_or(uint32_t* p, uint32_t v) { *p |= v; }
    void glb_or(uint32_t* p, uint32_t v) { *p |= v; }
    unsigned long long clock() { return 0; }
};
}  // namespace

// one member: in[0..n) -> out (65536 bytes, the raw DEFLATE stream from byte 0), result[0..2] = compressed bytes, CRC-32, status
extern "C" int deflate_sim_member(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t* result) {
    if (n < 1 || n > dfl::MAXN) return -1;
    std::vector<uint8_t> lds(dfl::L_END, 0xA5);                       // stale LDS
    std::vector<uint32_t> tokens(65536, 0xDEADBEEFu), outw(16384, 0);
    simt::run_block(dfl::NT, [&](simt::Block& b, int tid) {
        SimX x{&b, tid, tid & 63, tid >> 6};
        dfl::deflate_member(x, lds.data(), in, n, tokens.data(), outw.data(), result);
    });
    memcpy(out, outw.data(), 65536);
    return 0;
}

// thj_k_bam_shapes + thj_k_bam_write for n records: alns (API layout), the reads' inflated BAM records and where each row's starts.
// sizes / rids: n entries; out: the records back to back (the caller sizes it from a first call with out == NULL).
// Returns the stream's length, or -1 when a record needs the host encoder.
extern "C" int64_t bamenc_sim_records(const thj_aln* alns, int64_t n, const uint8_t* infl, const uint32_t* loc, const int32_t* tid_of_ref, uint32_t* sizes,
                                      int64_t* rids, uint8_t* out) {
    int64_t at = 0;
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* raw = infl + loc[alns[i].read_idx] + 4;
        const bamenc::Shape s = bamenc::record_shape(alns[i], raw);
        if (s.host_only) return -1;
        sizes[i] = s.size; rids[i] = s.rid;
        if (out) bamenc::record_write(alns[i], raw, s, tid_of_ref[alns[i].ref_id - 1], out + at);
        at += s.size;
    }
    return at;
}
