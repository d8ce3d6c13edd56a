// thj_scan.h -- exclusive prefix sum of n counts as three small kernels (tile sums, their scan by one workgroup, the tiles again with
// their offsets), for the per-shard scans of the ingest, the record compaction and the BAM writer.  Not hipcub::DeviceScan: the first call
// of a process into rocprim's device-wide algorithms costs 25-40 ms on the host (THJ_TRACE of a context's first ingest: the scan of a
// shard's million records; with that one replaced, the next rocprim call of the process paid it) -- every long_spanning_reads process
// paid it, and these scans are launch-bound anyway.  TI: the counts' type in memory, TO: the sums' (uint32_t or unsigned long long).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace thj_scan {
static constexpr int T = 256, IPT = 16, TILE = T * IPT;

template <class TI, class TO>
__global__ __launch_bounds__(T) void k_tiles(const TI* __restrict__ in, int64_t n, TO* __restrict__ tile_sum) {
    typedef hipcub::BlockReduce<TO, T> Red;
    __shared__ typename Red::TempStorage tmp;
    const int64_t base = (int64_t)blockIdx.x * TILE;
    TO v = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) { const int64_t i = base + (int64_t)k * T + threadIdx.x; if (i < n) v += (TO)in[i]; }
    const TO sum = Red(tmp).Sum(v);
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = sum;
}
template <class TO>
__global__ __launch_bounds__(1024) void k_sums(TO* __restrict__ tile_sum, int64_t n_tiles) {      // in place, exclusive; one workgroup
    typedef hipcub::BlockScan<TO, 1024> Scan;
    __shared__ typename Scan::TempStorage tmp;
    __shared__ TO carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b = 0; b < n_tiles; b += 1024) {
        const int64_t i = b + threadIdx.x;
        const TO v = i < n_tiles ? tile_sum[i] : (TO)0;
        TO ex, total;
        Scan(tmp).ExclusiveSum(v, ex, total);
        const TO c0 = carry;
        __syncthreads();
        if (i < n_tiles) tile_sum[i] = c0 + ex;
        if (threadIdx.x == 0) carry = c0 + total;
        __syncthreads();
    }
}
template <class TI, class TO>
__global__ __launch_bounds__(T) void k_apply(const TI* __restrict__ in, int64_t n, const TO* __restrict__ tile_off, TO* __restrict__ out) {
    typedef hipcub::BlockScan<TO, T> Scan;
    __shared__ typename Scan::TempStorage tmp;
    const int64_t base = (int64_t)blockIdx.x * TILE + (int64_t)threadIdx.x * IPT;      // a thread's IPT consecutive items
    TO v[IPT], sum = 0;
#pragma unroll
    for (int k = 0; k < IPT; ++k) { v[k] = base + k < n ? (TO)in[base + k] : (TO)0; sum += v[k]; }
    TO ex;
    Scan(tmp).ExclusiveSum(sum, ex);
    TO run = tile_off[blockIdx.x] + ex;
#pragma unroll
    for (int k = 0; k < IPT; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
}
inline size_t scratch_bytes(int64_t n, size_t sum_size) { return (size_t)((n + TILE - 1) / TILE) * sum_size + 256; }
// out[i] = in[0] + .. + in[i - 1]; scratch: scratch_bytes(n, sizeof(TO)) bytes of device memory, 8-byte aligned
template <class TI, class TO>
inline void exclusive_sum(hipStream_t stream, const TI* in, TO* out, int64_t n, void* scratch) {
    if (n <= 0) return;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    TO* ts = (TO*)scratch;
    hipLaunchKernelGGL((k_tiles<TI, TO>), dim3((unsigned)n_tiles), dim3(T), 0, stream, in, n, ts);
    hipLaunchKernelGGL((k_sums<TO>), dim3(1), dim3(1024), 0, stream, ts, n_tiles);
    hipLaunchKernelGGL((k_apply<TI, TO>), dim3((unsigned)n_tiles), dim3(T), 0, stream, in, n, (const TO*)ts, out);
}
}  // namespace thj_scan
