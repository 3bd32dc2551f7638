// prims.hip.h -- hand-written device-wide primitives for gfx950 (wave64): exclusive scan with fused
// input/output functors and a stable LSD radix sort of (uint64 key, payload) pairs.
//
// These replace the Thrust calls of the reference (scan / reduce_by_key / sort) and are the
// building blocks of the builder (src/bmSpMatrix.cu:167-216) and of the SpGEMM symbolic stages
// (src/bmSparse_SPGEMM.cu:839-1107).  All of them are HBM-bound integer passes: coalesced striped loads,
// wave64 ballots/shuffles for ranking, LDS only for per-workgroup counters.
#ifndef BMSP_PRIMS_HIP_H_
#define BMSP_PRIMS_HIP_H_

#include "runtime.h"
#include "bmsp_bits.h"
#include <chrono>
#include <cstring>

namespace bmsp {

constexpr int kWave = 64;
constexpr int kThreads = 256;          // 4 waves per workgroup
constexpr int kItems = 16;             // elements per thread per tile
constexpr int kTile = kThreads * kItems;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T v)
{
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T t = __shfl_up(v, d, kWave);
        if (lane_id() >= d) v += t;
    }
    return v;
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
}

// exclusive scan of one value per thread across a 256-thread workgroup; `total` is the workgroup sum.
// lds must hold 4 T's; two barriers inside, lds is reusable afterwards.
template <typename T>
__device__ __forceinline__ T block_exclusive_sum(T v, T *lds, T &total)
{
    T inc = wave_inclusive_sum(v);
    if (lane_id() == kWave - 1) lds[wave_id()] = inc;
    __syncthreads();
    T w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
    __syncthreads();
    int w = wave_id();
    T base = (w > 0 ? w0 : T(0)) + (w > 1 ? w1 : T(0)) + (w > 2 ? w2 : T(0));
    total = w0 + w1 + w2 + w3;
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan:  out(i, sum_{j<i} in(j))  for i in [0, n)
// In : T operator()(uint64_t i) const ;  Out : void operator()(uint64_t i, T exclusive) const
// ------------------------------------------------------------------------------------------------
// ITEMS elements per thread: kItems (4096 per tile) on long arrays, kItemsSmall on short ones so that they still spread
// over many workgroups and a tile is one round of loads
constexpr int kItemsSmall = 4;
template <typename T, typename In, int ITEMS>
__global__ __launch_bounds__(kThreads) void scan_tile_sums_kernel(In in, uint64_t n, T *tile_sums)
{
    __shared__ T lds[4];
    uint64_t base = (uint64_t)blockIdx.x * (ITEMS * kThreads);
    T local = 0;
#pragma unroll 4
    for (int k = 0; k < ITEMS; k++) {
        uint64_t i = base + (uint64_t)k * kThreads + threadIdx.x;
        if (i < n) local += in(i);
    }
    local = wave_sum(local);
    if (lane_id() == 0) lds[wave_id()] = local;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// Out functors may take the element's own input value as a third argument (saves re-deriving it from memory)
template <typename Out, typename T>
__device__ __forceinline__ auto scan_emit(const Out &o, uint64_t i, T ex, T v, int) -> decltype(o(i, ex, v), void())
{
    o(i, ex, v);
}
template <typename Out, typename T>
__device__ __forceinline__ void scan_emit(const Out &o, uint64_t i, T ex, T, long)
{
    o(i, ex);
}

template <typename T, typename In, typename Out, int ITEMS, bool SUM_TILES = false>
__global__ __launch_bounds__(kThreads) void scan_tile_down_kernel(In in, Out out, uint64_t n, const T *tile_excl,
                                                                  uint64_t tiles_per_block)
{
    constexpr uint64_t kTileS = (uint64_t)ITEMS * kThreads;
    __shared__ T lds[4];
    // tile_excl == nullptr: a single workgroup walks all tiles with a running carry
    uint64_t first_tile = (uint64_t)blockIdx.x * tiles_per_block;
    T carry = T(0);
    if (tile_excl && SUM_TILES) {
        // tile_excl holds the tiles' SUMS: the workgroup adds up the ones before it (few tiles: cheaper than a scan launch in between)
        T part = 0;
        for (uint64_t j = threadIdx.x; j < first_tile; j += kThreads) part += tile_excl[j];
        part = wave_sum(part);
        if (lane_id() == 0) lds[wave_id()] = part;
        __syncthreads();
        carry = lds[0] + lds[1] + lds[2] + lds[3];
        __syncthreads();
    } else if (tile_excl) {
        carry = tile_excl[first_tile];
    }
    for (uint64_t t = 0; t < tiles_per_block; t++) {
        uint64_t base = (first_tile + t) * kTileS;
        if (base >= n) break;
        // inputs are fetched kBatchLoads at a time (independent loads, one round trip per batch) before their block scans:
        // enough to hide the latency when a single workgroup walks a short array, without the register cost of a whole tile
        constexpr int kBatchLoads = 4;
#pragma unroll
        for (int k0 = 0; k0 < ITEMS; k0 += kBatchLoads) {
            T v[kBatchLoads];
#pragma unroll
            for (int k = 0; k < kBatchLoads; k++) {
                uint64_t i = base + (uint64_t)(k0 + k) * kThreads + threadIdx.x;
                v[k] = i < n ? in(i) : T(0);
            }
#pragma unroll
            for (int k = 0; k < kBatchLoads; k++) {
                uint64_t i = base + (uint64_t)(k0 + k) * kThreads + threadIdx.x;
                T total;
                T ex = block_exclusive_sum(v[k], lds, total);
                if (i < n) scan_emit<Out, T>(out, i, carry + ex, v[k], 0);
                carry += total;
            }
        }
    }
}

template <typename T>
struct PtrIn {
    const T *p;
    __device__ T operator()(uint64_t i) const { return p[i]; }
};
template <typename T>
struct PtrOut {
    T *p;
    __device__ void operator()(uint64_t i, T v) const { p[i] = v; }
};

template <typename T, typename In, typename Out, int ITEMS>
void device_exclusive_scan_impl(In in, Out out, uint64_t n, hipStream_t st);

template <typename T, typename In, typename Out>
void device_exclusive_scan(In in, Out out, uint64_t n, hipStream_t st)
{
    if (n == 0) return;
    if (n <= (1u << 20)) device_exclusive_scan_impl<T, In, Out, kItemsSmall>(in, out, n, st);
    else device_exclusive_scan_impl<T, In, Out, kItems>(in, out, n, st);
}

template <typename T, typename In, typename Out, int ITEMS>
void device_exclusive_scan_impl(In in, Out out, uint64_t n, hipStream_t st)
{
    constexpr uint64_t kTileS = (uint64_t)ITEMS * kThreads;
    uint64_t tiles = (n + kTileS - 1) / kTileS;
    if (n <= 8192) {  // a lone workgroup walks short arrays; beyond that three parallel launches beat its serial latency chain
        hipLaunchKernelGGL((scan_tile_down_kernel<T, In, Out, ITEMS>), dim3(1), dim3(kThreads), 0, st, in, out, n,
                           (const T *)nullptr, tiles);
        BMSP_CHECK_LAUNCH();
        return;
    }
    DevBuf<T> sums(tiles);
    hipLaunchKernelGGL((scan_tile_sums_kernel<T, In, ITEMS>), dim3((unsigned)tiles), dim3(kThreads), 0, st, in, n, sums.p);
    BMSP_CHECK_LAUNCH();
    if (tiles <= 1024) {
        // few tiles: every workgroup of the down pass adds the sums of the tiles before it itself -- two launches instead of three
        hipLaunchKernelGGL((scan_tile_down_kernel<T, In, Out, ITEMS, true>), dim3((unsigned)tiles), dim3(kThreads), 0, st, in, out, n,
                           (const T *)sums.p, (uint64_t)1);
        BMSP_CHECK_LAUNCH();
        return;
    }
    device_exclusive_scan<T>(PtrIn<T>{sums.p}, PtrOut<T>{sums.p}, tiles, st);
    hipLaunchKernelGGL((scan_tile_down_kernel<T, In, Out, ITEMS>), dim3((unsigned)tiles), dim3(kThreads), 0, st, in, out, n,
                       (const T *)sums.p, (uint64_t)1);
    BMSP_CHECK_LAUNCH();
    // sums is returned to the pool here; the pool never hands memory back to the driver while kernels
    // of this stream may still read it, and later users are ordered behind us on the same stream.
}

// ------------------------------------------------------------------------------------------------
// stable LSD radix sort of (uint64 key, payload) pairs, n < 2^32.  The digit width is chosen per sort (<= 9 bits): the
// bits in use are split evenly over the fewest passes, e.g. 26 bits = 9 + 9 + 8 (three passes), 28 bits = 4 x 7.
// ------------------------------------------------------------------------------------------------
constexpr int kRadixMaxBits = 9;
constexpr int kRadixMaxBins = 1 << kRadixMaxBits;  // two bins per thread

template <int ITEMS>
__global__ __launch_bounds__(kThreads) void radix_hist_kernel(const uint64_t *__restrict__ keys, uint32_t n, int shift, int bits,
                                                       uint32_t *__restrict__ hist, uint32_t num_tiles)
{
    __shared__ uint32_t h[kRadixMaxBins];
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    h[threadIdx.x] = 0;
    h[threadIdx.x + kThreads] = 0;
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * (ITEMS * kThreads);
#pragma unroll 4
    for (int k = 0; k < ITEMS; k++) {
        uint64_t i = base + (uint64_t)k * kThreads + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < bins; d += kThreads) hist[(uint64_t)d * num_tiles + blockIdx.x] = h[d];
}

// Each wave owns a contiguous 1024-key slice of the tile and ranks it 64 keys at a time: lanes holding the
// same digit find each other with one ballot per digit bit, the lowest of them bumps the wave's LDS counter for that
// digit, and every peer takes (old count + its position among the peers).  Slices, rounds and lanes are all visited
// in index order, so equal digits keep their input order (stable).
// The tile is then reordered IN LDS by digit before it leaves: consecutive threads write consecutive destinations, so a
// digit's run inside the tile (8+ elements on average) goes out as whole 64-byte segments instead of one scattered 8-byte
// store per key.  Keys and payloads take turns in the same 32 KB staging buffer.
template <typename P, int ITEMS>
__global__ __launch_bounds__(kThreads) void radix_scatter_kernel(const uint64_t *__restrict__ kin, const P *__restrict__ pin,
                                                                 uint64_t *__restrict__ kout, P *__restrict__ pout, uint32_t n,
                                                                 int shift, int bits, const uint32_t *__restrict__ hist_scanned,
                                                                 uint32_t num_tiles)
{
    __shared__ uint32_t cnt[4][kRadixMaxBins];
    __shared__ uint32_t local_base[4][kRadixMaxBins];  // first staging slot of (wave, digit)
    __shared__ uint32_t out_shift[kRadixMaxBins];      // global destination of a digit's run minus its first staging slot
    constexpr uint32_t kTileR = (uint32_t)ITEMS * kThreads;  // keys per workgroup: each wave owns ITEMS * 64 consecutive ones
    __shared__ uint64_t stage[kTileR];
    __shared__ uint32_t scan_lds[4];
    const uint32_t bins = 1u << bits, mask = bins - 1u;
    for (int w = 0; w < 4; w++) {
        cnt[w][threadIdx.x] = 0;
        cnt[w][threadIdx.x + kThreads] = 0;
    }
    __syncthreads();
    const int w = wave_id(), lane = lane_id();
    const uint64_t tile0 = (uint64_t)blockIdx.x * kTileR;
    const uint64_t slice = tile0 + (uint64_t)w * (kTileR / 4);
    const uint32_t tile_n = (uint32_t)min((uint64_t)kTileR, (uint64_t)n - tile0);
    uint64_t key[ITEMS];
    uint32_t pos[ITEMS];  // rank inside (wave, digit), then the staging slot
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint64_t i = slice + (uint64_t)r * kWave + lane;
        bool valid = i < n;
        key[r] = valid ? kin[i] : ~0ull;
        uint32_t d = (uint32_t)(key[r] >> shift) & mask;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kRadixMaxBits; b++) {
            if (b < bits) {  // wave-uniform
                bool bit = (d >> b) & 1u;
                uint64_t m = __ballot(valid && bit);
                peers &= bit ? m : ~m;
            }
        }
        uint32_t old = 0;
        int leader = valid ? (__ffsll((long long)peers) - 1) : lane;
        if (valid && lane == leader) old = atomicAdd(&cnt[w][d], (uint32_t)__popcll(peers));
        old = __shfl(old, leader, kWave);
        pos[r] = old + (uint32_t)__popcll(peers & lt);
    }
    __syncthreads();
    {
        // thread t owns digits 2t and 2t+1 (one digit when there are <= 256): digit order = thread order for the scan
        const bool two = bins > (uint32_t)kThreads;
        const uint32_t d0 = two ? 2u * threadIdx.x : threadIdx.x, d1 = d0 + 1u;
        uint32_t c0 = 0, c1 = 0;
        if (d0 < bins) c0 = cnt[0][d0] + cnt[1][d0] + cnt[2][d0] + cnt[3][d0];
        if (two) c1 = cnt[0][d1] + cnt[1][d1] + cnt[2][d1] + cnt[3][d1];
        uint32_t total;
        const uint32_t ex = block_exclusive_sum(c0 + c1, scan_lds, total);
        if (d0 < bins) {
            uint32_t g = ex;
            out_shift[d0] = hist_scanned[(uint64_t)d0 * num_tiles + blockIdx.x] - g;
#pragma unroll
            for (int ww = 0; ww < 4; ww++) { local_base[ww][d0] = g; g += cnt[ww][d0]; }
        }
        if (two) {
            uint32_t g = ex + c0;
            out_shift[d1] = hist_scanned[(uint64_t)d1 * num_tiles + blockIdx.x] - g;
#pragma unroll
            for (int ww = 0; ww < 4; ww++) { local_base[ww][d1] = g; g += cnt[ww][d1]; }
        }
    }
    __syncthreads();
    // keys through the staging buffer
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint64_t i = slice + (uint64_t)r * kWave + lane;
        if (i < n) {
            uint32_t d = (uint32_t)(key[r] >> shift) & mask;
            pos[r] += local_base[w][d];
            stage[pos[r]] = key[r];
        }
    }
    __syncthreads();
    uint32_t dst[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = (uint32_t)k * kThreads + threadIdx.x;
        dst[k] = 0;
        if (i < tile_n) {
            const uint64_t kk = stage[i];
            dst[k] = out_shift[(uint32_t)(kk >> shift) & mask] + i;
            kout[dst[k]] = kk;
        }
    }
    __syncthreads();
    // payloads through the same buffer
    P *pstage = reinterpret_cast<P *>(stage);
    P pv[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint64_t i = slice + (uint64_t)r * kWave + lane;
        if (i < n) pv[r] = pin[i];
    }
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
        uint64_t i = slice + (uint64_t)r * kWave + lane;
        if (i < n) pstage[pos[r]] = pv[r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint32_t i = (uint32_t)k * kThreads + threadIdx.x;
        if (i < tile_n) pout[dst[k]] = pstage[i];
    }
}

// bitonic network on P (power of two) machine words held in LDS; every participating thread calls it with the same P
template <typename W, int THREADS, bool BLOCK_SYNC>
__device__ __forceinline__ void bitonic_words(W *a, uint32_t P, uint32_t tid)
{
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t p = tid; p < P / 2; p += THREADS) {
                const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const uint32_t q = i | j;
                const W x = a[i], y = a[q];
                const bool up = (i & k) == 0;
                if ((x > y) == up) { a[i] = y; a[q] = x; }
            }
            if (BLOCK_SYNC) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
        }
    }
}

template <typename T>
struct PingPong {
    T *cur;
    T *alt;
    void flip()
    {
        T *t = cur;
        cur = alt;
        alt = t;
    }
};

// Sorts by key bits [begin_bit, end_bit).  On return keys.cur / vals.cur hold the sorted data (they may be the
// buffers that were passed as .alt).
template <typename P, int ITEMS>
void device_radix_sort_pairs_impl(PingPong<uint64_t> &keys, PingPong<P> &vals, uint32_t n, int begin_bit, int end_bit, hipStream_t st);

template <typename P>
void device_radix_sort_pairs(PingPong<uint64_t> &keys, PingPong<P> &vals, uint64_t n64, int begin_bit, int end_bit,
                             hipStream_t st)
{
    if (n64 == 0 || end_bit <= begin_bit) return;
    if (n64 >= (1ull << 32)) fail(BMSP_ERR_LIMIT, "radix sort: %llu elements exceed the 32-bit position range",
                                  (unsigned long long)n64);
    // short arrays: 1024-key tiles keep every pass spread over many workgroups
    if (n64 <= (1u << 20)) device_radix_sort_pairs_impl<P, kItemsSmall>(keys, vals, (uint32_t)n64, begin_bit, end_bit, st);
    else device_radix_sort_pairs_impl<P, kItems>(keys, vals, (uint32_t)n64, begin_bit, end_bit, st);
}

template <typename P, int ITEMS>
void device_radix_sort_pairs_impl(PingPong<uint64_t> &keys, PingPong<P> &vals, uint32_t n, int begin_bit, int end_bit, hipStream_t st)
{
    constexpr uint32_t kTileR = (uint32_t)ITEMS * kThreads;
    uint32_t tiles = (uint32_t)(((uint64_t)n + kTileR - 1) / kTileR);
    const int total_bits = end_bit - begin_bit;
    const int passes = (total_bits + kRadixMaxBits - 1) / kRadixMaxBits;
    const int width = (total_bits + passes - 1) / passes;
    DevBuf<uint32_t> hist((size_t)(1u << width) * tiles);
    for (int shift = begin_bit; shift < end_bit; shift += width) {
        const int bits = std::min(width, end_bit - shift);
        hipLaunchKernelGGL((radix_hist_kernel<ITEMS>), dim3(tiles), dim3(kThreads), 0, st, keys.cur, n, shift, bits, hist.p, tiles);
        BMSP_CHECK_LAUNCH();
        device_exclusive_scan<uint32_t>(PtrIn<uint32_t>{hist.p}, PtrOut<uint32_t>{hist.p}, (uint64_t)(1u << bits) * tiles, st);
        hipLaunchKernelGGL((radix_scatter_kernel<P, ITEMS>), dim3(tiles), dim3(kThreads), 0, st, keys.cur, vals.cur, keys.alt,
                           vals.alt, n, shift, bits, hist.p, tiles);
        BMSP_CHECK_LAUNCH();
        keys.flip();
        vals.flip();
    }
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
inline dim3 grid_for(uint64_t n, int threads = kThreads)
{
    uint64_t g = (n + threads - 1) / threads;
    if (g == 0) g = 1;
    if (g > 0x7fffffffull) fail(BMSP_ERR_LIMIT, "grid too large");
    return dim3((unsigned)g);
}

template <typename F>
__global__ __launch_bounds__(kThreads) void for_each_kernel(F f, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f(i);
}

template <typename F>
void device_for_each(F f, uint64_t n, hipStream_t st)
{
    if (n == 0) return;
    hipLaunchKernelGGL((for_each_kernel<F>), grid_for(n), dim3(kThreads), 0, st, f, n);
    BMSP_CHECK_LAUNCH();
}

// max and sum of in(i) over [0, n): grid-stride partial results, ONE pair of atomics per workgroup (same-address device atomics from
// every wave serialise at the memory side: ~30 ns each on MI355X, 4 ms for 131 K rows).  *out_max / *out_sum must be zeroed by the caller;
// either may be null.  In: uint64_t operator()(uint64_t i) const
template <typename In>
__global__ __launch_bounds__(kThreads) void max_sum_kernel(In in, uint64_t n, unsigned long long *out_max, unsigned long long *out_sum)
{
    __shared__ unsigned long long lmax[4], lsum[4];
    unsigned long long mx = 0, sm = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        const unsigned long long v = in(i);
        mx = v > mx ? v : mx;
        sm += v;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o = __shfl_xor(mx, d, kWave);
        mx = o > mx ? o : mx;
    }
    sm = wave_sum(sm);
    if (lane_id() == 0) { lmax[wave_id()] = mx; lsum[wave_id()] = sm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mx = lmax[w] > mx ? lmax[w] : mx; sm += lsum[w]; }
        if (out_max) atomicMax(out_max, mx);
        if (out_sum) atomicAdd(out_sum, sm);
    }
}
template <typename In>
void device_max_sum(In in, uint64_t n, unsigned long long *out_max, unsigned long long *out_sum, hipStream_t st)
{
    if (n == 0) return;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + kThreads - 1) / kThreads, 256);
    hipLaunchKernelGGL((max_sum_kernel<In>), dim3(grid), dim3(kThreads), 0, st, in, n, out_max, out_sum);
    BMSP_CHECK_LAUNCH();
}

// A scalar a kernel hands to the host: the slot is pinned host memory the device writes straight into, so fetching it is a
// stream synchronise and a host load -- no copy kernel, no staging (read_back below costs a blit launch per scalar).
void *host_slot_acquire();
void host_slot_release(void *p);
template <typename T>
struct HostScalar {
    T *p;
    mutable bool done = false;  // wait() has returned: the producing kernel's store has landed
    // all-ones = "not written yet": a genuine all-ones value only costs the 200 us poll before the stream synchronise below confirms it
    HostScalar() : p(static_cast<T *>(host_slot_acquire())) { *(volatile T *)p = ~T(0); }
    HostScalar(const HostScalar &) = delete;
    HostScalar &operator=(const HostScalar &) = delete;
    ~HostScalar()
    {
        // unwinding (an exception between the launch and wait()) with the producer possibly still in flight: the slot must not go
        // back to the pool while a kernel may still write it -- the next user would read that stale store as its own result
        if (!done) (void)hipDeviceSynchronize();
        host_slot_release(p);
    }
    T *dev() const { return p; }  // device-accessible address
    // The producing kernel stores the scalar with one aligned store into coherent host memory, so the host can pick it up as
    // soon as it lands instead of waiting for the kernel to drain and the queue to signal (~19 us of idle GPU per read-back on
    // this platform); after 200 us without a value it falls back to a stream synchronise.
    T wait(hipStream_t st) const
    {
        const T unset = ~T(0);
        const auto t0 = std::chrono::steady_clock::now();
        while (*(volatile T *)p == unset) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                BMSP_HIP(hipStreamSynchronize(st));
                break;
            }
        }
        done = true;
        return *(volatile T *)p;
    }
};
// PtrOut that also publishes the element at index `last` (the scan total) to a host scalar
template <typename T>
struct PtrOutTotal {
    T *p;
    uint64_t last;
    T *total;
    __device__ void operator()(uint64_t i, T v) const
    {
        p[i] = v;
        if (i == last) *total = v;
    }
};

// one scalar from device memory, through a PINNED host slot: a device-to-host copy into pageable memory makes the runtime set up its staging
// buffers the first time a process does one -- 21 ms inside the first product of the drop-in executable (measured: T_1's row statistics)
template <typename T>
T read_back(const T *dptr, hipStream_t st)
{
    static_assert(sizeof(T) <= 64, "a host slot holds 64 bytes");
    T *slot = static_cast<T *>(host_slot_acquire());
    hipError_t e = hipMemcpyAsync(slot, dptr, sizeof(T), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    const T v = *slot;
    host_slot_release(slot);
    if (e != hipSuccess) fail(BMSP_ERR_HIP, "read_back failed: %s", hipGetErrorString(e));
    return v;
}

// up to 64 bytes from device memory through a pinned host slot (see read_back)
inline void read_back_bytes(void *host, const void *dptr, size_t bytes, hipStream_t st)
{
    if (bytes > 64) fail(BMSP_ERR_INVALID, "read_back_bytes: a host slot holds 64 bytes");
    void *slot = host_slot_acquire();
    hipError_t e = hipMemcpyAsync(slot, dptr, bytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    memcpy(host, slot, bytes);
    host_slot_release(slot);
    if (e != hipSuccess) fail(BMSP_ERR_HIP, "read_back failed: %s", hipGetErrorString(e));
}

}  // namespace bmsp
#endif
