// segsort.hip -- stable segmented sort of (uint64 key, payload) pairs.
//
// Reference: bb_segsort<K,T>(keys, vals, n, segs, length), include/bb_segsort-master/bb_segsort.h:35-192, called
// by bmSparse_mult with K = uint64_t C keys, T = 16-byte tasks and one segment per block-row of A
// (src/bmSparse_SPGEMM.cu:973-1010).  bb_segsort is unstable; this one is stable.
//
// Segments are binned by length like bb_segsort does (bb_segsort.h:63-171), with wave64 shapes:
//   length <= 1            nothing to do
//   2 .. 256               one WAVE per segment: the segment lives in LDS (4 elements per lane) and is sorted by a
//                          bitonic network on the composite (key, position) -- the position makes the order stable
//   257 .. 4096            one 512-thread workgroup per segment, same network with workgroup barriers
//   longer                 (hub segments) fallback: two stable LSD radix sorts of a permutation -- by the key bits that
//                          vary, then by segment number -- over the whole array
// Every path produces a permutation; values move once, through one gather (as bb_segsort does, bb_comput_s.h:88).
#include "matrix.h"
#include "prims.hip.h"
#include <cstdlib>

namespace bmsp {
namespace {

struct SegOfElement {
    const int *segs;
    uint32_t nseg;
    uint32_t *seg_of;  // 0 = in front of the first segment, s+1 = segment s
    __device__ void operator()(uint64_t i) const
    {
        uint32_t lo = 0, hi = nseg;  // first segment whose start is > i
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if ((uint64_t)(int64_t)segs[mid] <= i) lo = mid + 1;
            else hi = mid;
        }
        seg_of[i] = lo;
    }
};

struct VaryingBits {
    const uint64_t *keys;
    uint64_t n;
    unsigned long long *acc;
    __device__ void operator()(uint64_t i) const
    {
        uint64_t x = keys[i] ^ keys[0];
        // one atomic per wave
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x |= __shfl_xor(x, d, kWave);
        if (lane_id() == 0 && x) atomicOr(acc, (unsigned long long)x);
    }
};

struct VaryingBitsClamped {
    VaryingBits f;
    __device__ void operator()(uint64_t i) const { f(i < f.n ? i : f.n - 1); }
};

struct Iota32Seg {
    uint32_t *p;
    __device__ void operator()(uint64_t i) const { p[i] = (uint32_t)i; }
};

struct CopyKeysIota {
    const uint64_t *in;
    uint64_t *out;
    uint32_t *idx;
    __device__ void operator()(uint64_t i) const
    {
        out[i] = in[i];
        idx[i] = (uint32_t)i;
    }
};

struct SegKeyOfPerm {
    const uint32_t *seg_of, *idx;
    uint64_t *out;
    __device__ void operator()(uint64_t i) const { out[i] = seg_of[idx[i]]; }
};

template <typename V>
struct GatherPairs {
    const uint64_t *kin;
    const V *vin;
    const uint32_t *idx;
    uint64_t *kout;
    V *vout;
    __device__ void operator()(uint64_t i) const
    {
        uint32_t s = idx[i];
        kout[i] = kin[s];
        if (vin) vout[i] = vin[s];
    }
};

struct Pair16 {
    uint64_t a, b;
};

template <typename V>
void segsort_impl(uint64_t *keys, V *vals, uint64_t n, const int *segs, uint32_t nseg, hipStream_t st)
{
    DevBuf<uint32_t> seg_of(n), i0(n), i1(n);
    DevBuf<uint64_t> k0(n), k1(n);
    DevBuf<unsigned long long> vary(1);
    BMSP_HIP(hipMemsetAsync(vary.p, 0, 8, st));
    device_for_each(SegOfElement{segs, nseg, seg_of.p}, n, st);
    {
        // every lane of a wave must reach the shuffles: round the launch up to whole waves, clamping the index
        uint64_t padded = (n + kWave - 1) / kWave * kWave;
        device_for_each(VaryingBitsClamped{VaryingBits{keys, n, vary.p}}, padded, st);
    }
    device_for_each(CopyKeysIota{keys, k0.p, i0.p}, n, st);
    uint64_t vb = read_back(vary.p, st);
    PingPong<uint64_t> kk{k0.p, k1.p};
    PingPong<uint32_t> ii{i0.p, i1.p};
    if (vb) {
        int lo = __builtin_ctzll(vb), hi = 64 - __builtin_clzll(vb);
        device_radix_sort_pairs<uint32_t>(kk, ii, n, lo, hi, st);
    }
    device_for_each(SegKeyOfPerm{seg_of.p, ii.cur, kk.cur}, n, st);
    device_radix_sort_pairs<uint32_t>(kk, ii, n, 0, ceil_log2_u64((uint64_t)nseg + 1), st);
    // gather into scratch, copy back in place (bb_segsort does the same, bb_segsort.h:175-178)
    DevBuf<V> vtmp(vals ? n : 1);
    device_for_each(GatherPairs<V>{keys, vals, ii.cur, kk.alt, vtmp.p}, n, st);
    BMSP_HIP(hipMemcpyAsync(keys, kk.alt, 8 * n, hipMemcpyDeviceToDevice, st));
    if (vals) BMSP_HIP(hipMemcpyAsync(vals, vtmp.p, sizeof(V) * n, hipMemcpyDeviceToDevice, st));
    BMSP_HIP(hipStreamSynchronize(st));
}

// ------------------------------------------------------------------------------------------------
// LDS bitonic paths
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kWaveSegMax = 256, kBlockSegMax = 4096;

struct SegClassify {
    const int *segs;
    uint32_t nseg;
    uint64_t n;
    uint32_t wave_max;  // longest segment the one-wave kernel takes
    __device__ uint32_t len(uint64_t s) const
    {
        const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
        return hi > lo ? (uint32_t)(hi - lo) : 0u;
    }
    // packed counts: [0,21) wave-class segments, [21,42) block-class, [42,63) long
    __device__ uint64_t operator()(uint64_t s) const
    {
        if (s >= nseg) return 0;
        const uint32_t l = len(s);
        if (l <= 1) return 0;
        return l <= wave_max ? 1ull : (l <= kBlockSegMax ? 1ull << 21 : 1ull << 42);
    }
};
struct SegLists {
    SegClassify c;
    uint32_t *wave_list, *block_list;
    uint64_t *totals;
    uint32_t *perm;  // identity is written for unit segments
    __device__ void operator()(uint64_t s, uint64_t ex, uint64_t cls) const
    {
        if (s == c.nseg) { *totals = ex; return; }
        if (cls == 0) {  // length 0 or 1
            if (c.len(s) == 1) perm[c.segs[s]] = (uint32_t)c.segs[s];
            return;
        }
        if (cls == 1ull) wave_list[ex & 0x1fffffu] = (uint32_t)s;
        else if (cls == (1ull << 21)) block_list[(ex >> 21) & 0x1fffffu] = (uint32_t)s;
    }
};

__device__ __forceinline__ bool composite_greater(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib)
{
    return ka > kb || (ka == kb && ia > ib);
}

// sorts P (power of two) composite elements held in LDS; every participating thread calls this with the same P.
template <int THREADS, bool BLOCK_SYNC>
__device__ __forceinline__ void bitonic_lds(uint64_t *key, uint16_t *idx, uint32_t P, uint32_t tid)
{
    for (uint32_t k = 2; k <= P; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t p = tid; p < P / 2; p += THREADS) {
                const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const uint32_t q = i | j;
                const uint64_t ka = key[i], kb = key[q];
                const uint32_t ia = idx[i], ib = idx[q];
                const bool up = (i & k) == 0;
                if (composite_greater(ka, ia, kb, ib) == up) {
                    key[i] = kb; key[q] = ka;
                    idx[i] = (uint16_t)ib; idx[q] = (uint16_t)ia;
                }
            }
            if (BLOCK_SYNC) __syncthreads();
            else __builtin_amdgcn_wave_barrier();
        }
    }
}

__global__ __launch_bounds__(kThreads) void segsort_wave_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, const int *__restrict__ segs,
                                                                uint32_t nseg, uint64_t n, const uint32_t *__restrict__ list, uint32_t count)
{
    __shared__ uint64_t s_key[4][kWaveSegMax];
    __shared__ uint16_t s_idx[4][kWaveSegMax];
    const int w = wave_id(), lane = lane_id();
    const uint32_t li = blockIdx.x * 4 + w;
    if (li >= count) return;
    const uint32_t s = list[li];
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    const uint32_t len = (uint32_t)(hi - lo);
    uint32_t P = 2;
    while (P < len) P <<= 1;
    uint64_t *key = s_key[w];
    uint16_t *idx = s_idx[w];
    for (uint32_t e = lane; e < P; e += 64) {
        key[e] = e < len ? keys[lo + e] : ~0ull;
        idx[e] = e < len ? (uint16_t)e : (uint16_t)0xffff;
    }
    __builtin_amdgcn_wave_barrier();
    bitonic_lds<64, false>(key, idx, P, (uint32_t)lane);
    for (uint32_t e = lane; e < len; e += 64) {
        keys[lo + e] = key[e];
        perm[lo + e] = (uint32_t)(lo + idx[e]);
    }
}

__global__ __launch_bounds__(512) void segsort_block_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, const int *__restrict__ segs,
                                                            uint32_t nseg, uint64_t n, const uint32_t *__restrict__ list)
{
    __shared__ uint64_t key[kBlockSegMax];
    __shared__ uint16_t idx[kBlockSegMax];
    const uint32_t s = list[blockIdx.x];
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    const uint32_t len = (uint32_t)(hi - lo);
    uint32_t P = 512;
    while (P < len) P <<= 1;
    for (uint32_t e = threadIdx.x; e < P; e += 512) {
        key[e] = e < len ? keys[lo + e] : ~0ull;
        idx[e] = e < len ? (uint16_t)e : (uint16_t)0xffff;
    }
    __syncthreads();
    bitonic_lds<512, true>(key, idx, P, threadIdx.x);
    for (uint32_t e = threadIdx.x; e < len; e += 512) {
        keys[lo + e] = key[e];
        perm[lo + e] = (uint32_t)(lo + idx[e]);
    }
}

template <typename V>
struct GatherVals {
    const V *vin;
    const uint32_t *perm;
    V *vout;
    __device__ void operator()(uint64_t i) const { vout[i] = vin[perm[i]]; }
};

// Sorts every segment in place through LDS and writes the permutation that was applied (perm[i] = source position of
// the element now at i).  Returns false -- with nothing modified -- when some segment is too long for the LDS paths.
// `segs[0]` must be 0 unless `perm` was pre-filled with the identity.
bool segsort_lds_perm(uint64_t *keys, uint32_t *perm, uint64_t n, const int *segs, uint32_t nseg, hipStream_t st)
{
    if (nseg >= (1u << 21)) return false;  // packed 21-bit class counters
    SegClassify cls{segs, nseg, n, kWaveSegMax};
    DevBuf<uint32_t> wave_list(nseg), block_list(nseg);
    DevBuf<uint64_t> tot(1);
    device_exclusive_scan<uint64_t>(cls, SegLists{cls, wave_list.p, block_list.p, tot.p, perm}, (uint64_t)nseg + 1, st);
    const uint64_t t = read_back(tot.p, st);
    const uint32_t n_wave = (uint32_t)(t & 0x1fffffu), n_block = (uint32_t)((t >> 21) & 0x1fffffu), n_long = (uint32_t)(t >> 42);
    if (n_long) return false;
    if (n_wave) {
        hipLaunchKernelGGL(segsort_wave_kernel, dim3((n_wave + 3) / 4), dim3(kThreads), 0, st, keys, perm, segs, nseg, n, wave_list.p, n_wave);
        BMSP_CHECK_LAUNCH();
    }
    if (n_block) {
        hipLaunchKernelGGL(segsort_block_kernel, dim3(n_block), dim3(512), 0, st, keys, perm, segs, nseg, n, block_list.p);
        BMSP_CHECK_LAUNCH();
    }
    return true;
}

template <typename V>
bool segsort_lds(uint64_t *keys, V *vals, uint64_t n, const int *segs, uint32_t nseg, hipStream_t st)
{
    DevBuf<uint32_t> perm(n);
    device_for_each(Iota32Seg{perm.p}, n, st);  // positions outside every segment stay put
    if (!segsort_lds_perm(keys, perm.p, n, segs, nseg, st)) return false;
    if (vals) {
        DevBuf<V> vtmp(n);
        device_for_each(GatherVals<V>{vals, perm.p, vtmp.p}, n, st);
        BMSP_HIP(hipMemcpyAsync(vals, vtmp.p, sizeof(V) * n, hipMemcpyDeviceToDevice, st));
    }
    BMSP_HIP(hipStreamSynchronize(st));
    return true;
}

struct RunHead {
    const uint64_t *keys;
    uint64_t n;
    int shift;
    __device__ uint32_t operator()(uint64_t i) const
    {
        if (i >= n) return 0;
        return (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
    }
};
struct EmitRunStarts {
    const uint64_t *keys;
    uint64_t n;
    int shift;
    int *segs;
    uint32_t *count;      // pinned host scalar (the general path reads it back)
    uint32_t *count_dev;  // device copy (the read-back-free path's kernels test their segment index against it)
    __device__ void operator()(uint64_t i, uint32_t ex, uint32_t is_head) const
    {
        if (i == n) {
            if (count) *count = ex;
            if (count_dev) *count_dev = ex;
            return;
        }
        if (is_head) segs[ex] = (int)i;  // RunHead's value for this element
    }
};
}  // namespace

void segsort_u64(uint64_t *d_keys, void *d_vals, int val_bytes, int64_t n, const int *d_segs, int64_t num_segs, hipStream_t st)
{
    if (n < 0 || num_segs < 0) fail(BMSP_ERR_INVALID, "negative size");
    if (n == 0 || num_segs == 0) return;
    if (n >= (1ll << 31)) fail(BMSP_ERR_LIMIT, "segmented sort handles < 2^31 elements (int segment starts)");
    if (!d_keys || !d_segs) fail(BMSP_ERR_INVALID, "null pointer");
    const uint64_t un = (uint64_t)n;
    const uint32_t ns = (uint32_t)num_segs;
    if (!d_vals) { if (!segsort_lds<uint32_t>(d_keys, nullptr, un, d_segs, ns, st)) segsort_impl<uint32_t>(d_keys, nullptr, un, d_segs, ns, st); }
    else if (val_bytes == 4) { if (!segsort_lds<uint32_t>(d_keys, (uint32_t *)d_vals, un, d_segs, ns, st)) segsort_impl<uint32_t>(d_keys, (uint32_t *)d_vals, un, d_segs, ns, st); }
    else if (val_bytes == 8) { if (!segsort_lds<uint64_t>(d_keys, (uint64_t *)d_vals, un, d_segs, ns, st)) segsort_impl<uint64_t>(d_keys, (uint64_t *)d_vals, un, d_segs, ns, st); }
    else if (val_bytes == 16) { if (!segsort_lds<Pair16>(d_keys, (Pair16 *)d_vals, un, d_segs, ns, st)) segsort_impl<Pair16>(d_keys, (Pair16 *)d_vals, un, d_segs, ns, st); }
    else fail(BMSP_ERR_INVALID, "val_bytes must be 4, 8 or 16");
}

// ------------------------------------------------------------------------------------------------
// SpGEMM task lists: inside a block-row segment only the column part of the packed key varies, and a position inside
// the segment needs <= 12 bits, so (column << idx_bits) | position is ONE machine word -- 32 bits whenever
// jbits + idx_bits <= 32 -- and the bitonic network is a plain compare-exchange on LDS words (half / a quarter of the
// LDS traffic of the generic (u64 key, u16 position) network above).  Segments up to 1024 tasks stay inside one wave
// (no s_barrier between the network's steps); longer ones take a 512-thread workgroup.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kTaskWaveMax = 2048, kTaskBlockIdxBits = 12;

template <typename W, int THREADS, bool BLOCK_SYNC, uint32_t IDX_BITS>
__device__ __forceinline__ void sort_task_segment(W *a, uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, uint64_t lo, uint32_t len, uint32_t p_min,
                                                  uint64_t col_mask, uint32_t tid)
{
    uint32_t P = p_min;
    while (P < len) P <<= 1;
    for (uint32_t e = tid; e < P; e += THREADS) a[e] = e < len ? (W)(((keys[lo + e] & col_mask) << IDX_BITS) | e) : (W)~(W)0;
    const uint64_t row_part = keys[lo] & ~col_mask;  // read before any key of the segment is overwritten (barrier below)
    if (BLOCK_SYNC) __syncthreads();
    else __builtin_amdgcn_wave_barrier();
    bitonic_words<W, THREADS, BLOCK_SYNC>(a, P, tid);
    for (uint32_t e = tid; e < len; e += THREADS) {
        const W c = a[e];
        keys[lo + e] = row_part | (uint64_t)(c >> IDX_BITS);
        perm[lo + e] = (uint32_t)(lo + (uint32_t)(c & (W)((1u << IDX_BITS) - 1u)));
    }
}

// ---- one wave per segment, the words in REGISTERS ---------------------------------------------------------------------------------
// Blocked layout: lane l holds words l*E .. l*E + E-1 of the 64*E-word network.  A compare-exchange at distance j < E is two VALU
// instructions on two registers of one lane; at distance j >= E the partner is the same register of lane l ^ (j / E): a DPP operand for
// lane distances 1, 2 and 8 (14 of the 21 cross-lane steps of any size), ds_swizzle for 4 and 16, one bpermute for 32.  The LDS network
// above reads and writes LDS twice per compare-exchange and computes two indices for it; here 45 of the 66 steps of a 2048-word sort never
// leave the registers.  Descending runs are sorted ascending on complemented words (one xor per word and phase) so that every
// compare-exchange is a plain (min, max).
template <int M>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t v)
{
    if constexpr (M == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    else if constexpr (M == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true);  // row_ror:8
    else if constexpr (M == 4 || M == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (M << 10) | 0x1f);  // bit mode: xor M
    else return (uint32_t)__shfl_xor((int)v, 32, kWave);
}
template <int M>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) { return lane_xor_u32<M>(v); }
template <int M>
__device__ __forceinline__ uint64_t lane_xor(uint64_t v)
{
    return ((uint64_t)lane_xor_u32<M>((uint32_t)(v >> 32)) << 32) | lane_xor_u32<M>((uint32_t)v);
}

template <typename W, int E, int K, int J>
struct BitonicStep {
    static __device__ __forceinline__ void run(W (&x)[E], int lane)
    {
        if constexpr (J >= E) {
            constexpr int M = J / E;
            const bool lower = (lane & M) == 0;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const W y = lane_xor<M>(x[e]);
                const W mn = x[e] < y ? x[e] : y, mx = x[e] < y ? y : x[e];
                x[e] = lower ? mn : mx;
            }
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) {
                if ((e & J) == 0) {
                    const bool down = K < E && (e & K) != 0;  // inside a lane the direction is a property of the register
                    const W a = x[e], b = x[e | J];
                    const W mn = a < b ? a : b, mx = a < b ? b : a;
                    x[e] = down ? mx : mn;
                    x[e | J] = down ? mn : mx;
                }
            }
        }
        if constexpr (J > 1) BitonicStep<W, E, K, J / 2>::run(x, lane);
    }
};
template <typename W, int E, int K>
struct BitonicPhase {
    static __device__ __forceinline__ void run(W (&x)[E], int lane)
    {
        // K >= E: the run's direction belongs to the lane (bit K / E of it; the last phase is ascending everywhere)
        const bool flip = K >= E && K < 64 * E && (lane & (K / E)) != 0;
        if constexpr (K >= E && K < 64 * E) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = flip ? (W)~x[e] : x[e];
        }
        BitonicStep<W, E, K, K / 2>::run(x, lane);
        if constexpr (K >= E && K < 64 * E) {
#pragma unroll
            for (int e = 0; e < E; e++) x[e] = flip ? (W)~x[e] : x[e];
        }
        if constexpr (K < 64 * E) BitonicPhase<W, E, K * 2>::run(x, lane);
    }
};

template <typename W, int E, uint32_t IDX_BITS>
__device__ __forceinline__ void sort_task_segment_regs(uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, uint64_t lo, uint32_t len, uint64_t col_mask,
                                                       int lane)
{
    W x[E];
    const uint64_t row_part = keys[lo] & ~col_mask;  // constant inside a segment
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t i = (uint32_t)lane * E + (uint32_t)e;
        x[e] = i < len ? (W)(((keys[lo + i] & col_mask) << IDX_BITS) | i) : (W)~(W)0;
    }
    BitonicPhase<W, E, 2>::run(x, lane);
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t i = (uint32_t)lane * E + (uint32_t)e;
        if (i < len) {
            keys[lo + i] = row_part | (uint64_t)(x[e] >> IDX_BITS);
            perm[lo + i] = (uint32_t)(lo + (uint32_t)(x[e] & (W)((1u << IDX_BITS) - 1u)));
        }
    }
}

// (two launches over the same list -- segments of <= 512 words in a kernel of their own at full occupancy -- were measured: the second
// launch costs more than the occupancy returns, cage-like T_5 195 -> 220 us)
template <typename W>
__global__ __launch_bounds__(kThreads) void segsort_tasks_wave_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, const int *__restrict__ segs,
                                                                      uint32_t nseg, uint64_t n, const uint32_t *__restrict__ list, uint32_t count,
                                                                      uint64_t col_mask)
{
    const uint32_t li = blockIdx.x * 4 + (uint32_t)wave_id();
    if (li >= count) return;
    const uint32_t s = list[li];
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hi - lo));
    const int lane = lane_id();
    if (len <= 128) sort_task_segment_regs<W, 2, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (len <= 256) sort_task_segment_regs<W, 4, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (len <= 512) sort_task_segment_regs<W, 8, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (len <= 1024) sort_task_segment_regs<W, 16, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (sizeof(W) == 8 || len <= 2048) sort_task_segment_regs<W, 32, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if constexpr (sizeof(W) == 4) sort_task_segment_regs<W, 64, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);  // 64 words per lane
}

// the same sort without segment lists: wave i takes segment i of the run-start array.  For products whose block-row segments are known
// to fit a wave BEFORE the tasks exist (most blocks per block-row of A x most blocks per block-row of B <= 4096), so that neither the
// segment count nor the class counts have to travel to the host: the grid is sized by A's block-rows, surplus waves leave.
template <typename W>
__global__ __launch_bounds__(kThreads) void segsort_tasks_direct_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, const int *__restrict__ segs,
                                                                        const uint32_t *__restrict__ nseg_dev, uint64_t n, uint64_t col_mask)
{
    const uint32_t s = blockIdx.x * 4 + (uint32_t)wave_id();
    const uint32_t nseg = *nseg_dev;
    if (s >= nseg) return;
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hi - lo));
    const int lane = lane_id();
    if (len == 1) {
        if (lane == 0) perm[lo] = (uint32_t)lo;
    } else if (len <= 128) sort_task_segment_regs<W, 2, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (len <= 256) sort_task_segment_regs<W, 4, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (len <= 512) sort_task_segment_regs<W, 8, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (len <= 1024) sort_task_segment_regs<W, 16, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if (len <= 2048) sort_task_segment_regs<W, 32, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
    else if constexpr (sizeof(W) == 4) sort_task_segment_regs<W, 64, kTaskBlockIdxBits>(keys, perm, lo, len, col_mask, lane);
}

template <typename W>
__global__ __launch_bounds__(512) void segsort_tasks_block_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, const int *__restrict__ segs,
                                                                  uint32_t nseg, uint64_t n, const uint32_t *__restrict__ list, uint64_t col_mask)
{
    __shared__ W s_a[kBlockSegMax];
    const uint32_t s = list[blockIdx.x];
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    sort_task_segment<W, 512, true, kTaskBlockIdxBits>(s_a, keys, perm, lo, (uint32_t)(hi - lo), 1024u, col_mask, threadIdx.x);
}

template <typename W>
bool segsort_tasks_lds(uint64_t *keys, uint32_t *perm, uint64_t n, const int *segs, uint32_t nseg, int jbits, hipStream_t st)
{
    if (nseg >= (1u << 21)) return false;
    SegClassify cls{segs, nseg, n, sizeof(W) == 4 ? 2 * kTaskWaveMax : kTaskWaveMax};  // 32-bit words: a wave holds a 4096-word segment in registers
    DevBuf<uint32_t> wave_list(nseg), block_list(nseg);
    HostScalar<uint64_t> tot;
    device_exclusive_scan<uint64_t>(cls, SegLists{cls, wave_list.p, block_list.p, tot.dev(), perm}, (uint64_t)nseg + 1, st);
    const uint64_t t = tot.wait(st);
    const uint32_t n_wave = (uint32_t)(t & 0x1fffffu), n_block = (uint32_t)((t >> 21) & 0x1fffffu), n_long = (uint32_t)(t >> 42);
    if (n_long) return false;
    const uint64_t col_mask = (1ull << jbits) - 1ull;
    if (n_wave) {
        hipLaunchKernelGGL((segsort_tasks_wave_kernel<W>), dim3((n_wave + 3) / 4), dim3(kThreads), 0, st, keys, perm, segs, nseg, n, wave_list.p, n_wave, col_mask);
        BMSP_CHECK_LAUNCH();
    }
    if (n_block) {
        hipLaunchKernelGGL((segsort_tasks_block_kernel<W>), dim3(n_block), dim3(512), 0, st, keys, perm, segs, nseg, n, block_list.p, col_mask);
        BMSP_CHECK_LAUNCH();
    }
    return true;
}

bool segsort_tasks_by_column(PingPong<uint64_t> &keys, PingPong<uint64_t> &vals, uint64_t n, int jbits, hipStream_t st, uint64_t max_seg_bound,
                             uint64_t seg_count_bound)
{
    if (n >= (1ull << 31)) return false;
    // segments = runs of equal block-row (reference: :982-1004).  Inside a run the row part of the packed key is
    // constant, so comparing whole keys orders by column: no masking pass is needed.
    DevBuf<int> segs(n);
    const bool narrow = jbits + (int)kTaskBlockIdxBits <= 32 && !getenv("BMSP_SEGSORT_WIDE");  // the variable forces the 64-bit sort words (tests)
    const uint64_t col_mask = (1ull << jbits) - 1ull;
    // (moving the payload inside the sort kernels -- no permutation array, no gather pass -- was measured: T_5 605 -> 983 us on the
    // FEM-like product; E scattered 8-byte gathers per lane at 3 waves per SIMD are slower than one fully parallel gather pass)
    DevBuf<uint32_t> perm(n);
    const uint64_t wave_cap = narrow ? 2 * kTaskWaveMax : kTaskWaveMax;
    if (max_seg_bound && max_seg_bound <= wave_cap && seg_count_bound && !getenv("BMSP_SEGSORT_READBACK")) {
        // every segment fits one wave, known from the operands alone: no segment lists, no count on the host
        DevBuf<uint32_t> nseg_dev(1);
        device_exclusive_scan<uint32_t>(RunHead{keys.cur, n, jbits}, EmitRunStarts{keys.cur, n, jbits, segs.p, nullptr, nseg_dev.p}, n + 1, st);
        const uint64_t waves = std::min<uint64_t>(seg_count_bound, n);
        if (narrow)
            hipLaunchKernelGGL((segsort_tasks_direct_kernel<uint32_t>), dim3((unsigned)((waves + 3) / 4)), dim3(kThreads), 0, st, keys.cur, perm.p, segs.p,
                               nseg_dev.p, n, col_mask);
        else
            hipLaunchKernelGGL((segsort_tasks_direct_kernel<uint64_t>), dim3((unsigned)((waves + 3) / 4)), dim3(kThreads), 0, st, keys.cur, perm.p, segs.p,
                               nseg_dev.p, n, col_mask);
        BMSP_CHECK_LAUNCH();
        device_for_each(GatherVals<uint64_t>{vals.cur, perm.p, vals.alt}, n, st);
        vals.flip();
        return true;
    }
    HostScalar<uint32_t> cnt;
    device_exclusive_scan<uint32_t>(RunHead{keys.cur, n, jbits}, EmitRunStarts{keys.cur, n, jbits, segs.p, cnt.dev(), nullptr}, n + 1, st);
    const uint32_t nseg = cnt.wait(st);
    const bool ok = narrow ? segsort_tasks_lds<uint32_t>(keys.cur, perm.p, n, segs.p, nseg, jbits, st)
                           : segsort_tasks_lds<uint64_t>(keys.cur, perm.p, n, segs.p, nseg, jbits, st);
    if (!ok) return false;  // hub rows: caller takes the global sort
    device_for_each(GatherVals<uint64_t>{vals.cur, perm.p, vals.alt}, n, st);
    vals.flip();
    return true;
}

}  // namespace bmsp
