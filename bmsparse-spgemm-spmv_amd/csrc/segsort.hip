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
#include <utility>

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
    __device__ bool is_long(uint32_t l) const { return l > wave_max && l > kBlockSegMax; }
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
__device__ __forceinline__ void sort_task_segment(W *a, const uint64_t *keys, uint64_t *keys_out, uint32_t *__restrict__ perm, uint64_t lo, uint32_t len,
                                                  uint32_t p_min, uint64_t col_mask, uint32_t tid)
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
        keys_out[lo + e] = row_part | (uint64_t)(c >> IDX_BITS);
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
__device__ __forceinline__ void sort_task_segment_regs(const uint64_t *keys, uint64_t *keys_out, uint32_t *__restrict__ perm, uint64_t lo, uint32_t len,
                                                       uint64_t col_mask, int lane)
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
            keys_out[lo + i] = row_part | (uint64_t)(x[e] >> IDX_BITS);
            perm[lo + i] = (uint32_t)(lo + (uint32_t)(x[e] & (W)((1u << IDX_BITS) - 1u)));
        }
    }
}

// (two launches over the same list -- segments of <= 512 words in a kernel of their own at full occupancy -- were measured: the second
// launch costs more than the occupancy returns, cage-like T_5 195 -> 220 us)
template <typename W>
__global__ __launch_bounds__(kThreads) void segsort_tasks_wave_kernel(const uint64_t *keys, uint64_t *keys_out, uint32_t *__restrict__ perm,
                                                                      const int *__restrict__ segs, uint32_t nseg, uint64_t n,
                                                                      const uint32_t *__restrict__ list, uint32_t count, uint64_t col_mask)
{
    const uint32_t li = blockIdx.x * 4 + (uint32_t)wave_id();
    if (li >= count) return;
    const uint32_t s = list[li];
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hi - lo));
    const int lane = lane_id();
    if (len <= 128) sort_task_segment_regs<W, 2, kTaskBlockIdxBits>(keys, keys_out, perm, lo, len, col_mask, lane);
    else if (len <= 256) sort_task_segment_regs<W, 4, kTaskBlockIdxBits>(keys, keys_out, perm, lo, len, col_mask, lane);
    else if (len <= 512) sort_task_segment_regs<W, 8, kTaskBlockIdxBits>(keys, keys_out, perm, lo, len, col_mask, lane);
    else if (len <= 1024) sort_task_segment_regs<W, 16, kTaskBlockIdxBits>(keys, keys_out, perm, lo, len, col_mask, lane);
    else if (sizeof(W) == 8 || len <= 2048) sort_task_segment_regs<W, 32, kTaskBlockIdxBits>(keys, keys_out, perm, lo, len, col_mask, lane);
    else if constexpr (sizeof(W) == 4) sort_task_segment_regs<W, 64, kTaskBlockIdxBits>(keys, keys_out, perm, lo, len, col_mask, lane);  // 64 words per lane
}

// the same sort without segment lists: wave i takes segment i of the run-start array.  For products whose block-row segments are known
// to fit a wave BEFORE the tasks exist (most blocks per block-row of A x most blocks per block-row of B <= 4096), so that neither the
// segment count nor the class counts have to travel to the host: the grid is sized by A's block-rows, surplus waves leave.
template <typename W>
__global__ __launch_bounds__(kThreads) void segsort_tasks_direct_kernel(uint64_t *__restrict__ keys, uint32_t *__restrict__ perm, const int *__restrict__ segs,
                                                                        const uint32_t *__restrict__ nseg_dev, uint64_t n, uint64_t col_mask,
                                                                        uint32_t *__restrict__ violation)
{
    const uint32_t s = blockIdx.x * 4 + (uint32_t)wave_id();
    const uint32_t nseg = *nseg_dev;
    if (s >= nseg) return;
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(hi - lo));
    const int lane = lane_id();
    if (len == 1) {
        if (lane == 0) perm[lo] = (uint32_t)lo;
    } else if (len <= 128) sort_task_segment_regs<W, 2, kTaskBlockIdxBits>(keys, keys, perm, lo, len, col_mask, lane);
    else if (len <= 256) sort_task_segment_regs<W, 4, kTaskBlockIdxBits>(keys, keys, perm, lo, len, col_mask, lane);
    else if (len <= 512) sort_task_segment_regs<W, 8, kTaskBlockIdxBits>(keys, keys, perm, lo, len, col_mask, lane);
    else if (len <= 1024) sort_task_segment_regs<W, 16, kTaskBlockIdxBits>(keys, keys, perm, lo, len, col_mask, lane);
    else if (len <= 2048) sort_task_segment_regs<W, 32, kTaskBlockIdxBits>(keys, keys, perm, lo, len, col_mask, lane);
    else {
        if constexpr (sizeof(W) == 4) {
            if (len <= 4096) {
                sort_task_segment_regs<W, 64, kTaskBlockIdxBits>(keys, keys, perm, lo, len, col_mask, lane);
                return;
            }
        }
        if (lane == 0) *violation = 1u;  // a segment beyond the wave's capacity: the caller's bound was wrong -- never left unsorted in silence
    }
}

template <typename W>
__global__ __launch_bounds__(512) void segsort_tasks_block_kernel(const uint64_t *keys, uint64_t *keys_out, uint32_t *__restrict__ perm,
                                                                  const int *__restrict__ segs, uint32_t nseg, uint64_t n,
                                                                  const uint32_t *__restrict__ list, uint64_t col_mask)
{
    __shared__ W s_a[kBlockSegMax];
    const uint32_t s = list[blockIdx.x];
    const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
    sort_task_segment<W, 512, true, kTaskBlockIdxBits>(s_a, keys, keys_out, perm, lo, (uint32_t)(hi - lo), 1024u, col_mask, threadIdx.x);
}

// ---- segments beyond a wave's / workgroup's capacity: pieces + merge passes --------------------------------------------------------
// bb_segsort sorts segments above 2048 with a block sort of 2048-key pieces followed by doubling merge passes (kern_block_sort /
// kern_block_merge, include/bb_segsort-master/bb_comput_l.h:1155-1284).  Same plan here: a long segment is cut into pieces of at most
// `cap` words, the pieces go through the register / LDS sorts above like ordinary segments, then ceil(log2(pieces)) merge passes join
// them.  A merge pass works on output tiles of kMergeTile elements (merge path: two binary searches on the tile's diagonals find the
// input ranges, the ranges meet in LDS, every thread merges four outputs); what is merged is the composite word (column << 32) | source
// position, which is unique, so the result is the STABLE order whatever the merge does with ties.  Before round 3 one long segment sent
// the whole call to the global radix sort.
constexpr uint32_t kMergeTile = 1024;

struct PiecesIn {  // pieces a segment is cut into (1 for segments that fit)
    SegClassify c;
    uint32_t cap;
    __device__ uint32_t operator()(uint64_t s) const { return s < c.nseg ? max(1u, (c.len(s) + cap - 1u) / cap) : 0u; }
};
struct PiecesOut {
    SegClassify c;
    uint32_t cap;
    int *segs2;
    uint32_t *total, *max_len, *long_elems;
    __device__ void operator()(uint64_t s, uint32_t ex, uint32_t cnt) const
    {
        if (s == c.nseg) { *total = ex; return; }
        const uint32_t lo = (uint32_t)c.segs[s];
        for (uint32_t p = 0; p < cnt; p++) segs2[ex + p] = (int)(lo + p * cap);
        if (cnt > 1) { atomicMax(max_len, c.len(s)); atomicAdd(long_elems, c.len(s)); }
    }
};
struct MergeWork {
    uint32_t lo, len, tile;
};
struct MergeTilesIn {  // output tiles of the LONG segments (the others are not touched by the merge passes)
    SegClassify c;
    uint32_t cap;
    __device__ uint32_t operator()(uint64_t s) const
    {
        if (s >= c.nseg) return 0u;
        const uint32_t l = c.len(s);
        return l > cap ? (l + kMergeTile - 1u) / kMergeTile : 0u;
    }
};
struct MergeTilesOut {
    SegClassify c;
    MergeWork *work;
    uint32_t *total;
    __device__ void operator()(uint64_t s, uint32_t ex, uint32_t cnt) const
    {
        if (s == c.nseg) { *total = ex; return; }
        for (uint32_t t = 0; t < cnt; t++) work[ex + t] = MergeWork{(uint32_t)c.segs[s], c.len(s), t};
    }
};

// one pass: runs of length R inside every long segment are merged pairwise; (src) -> (dst)
__global__ __launch_bounds__(kThreads) void merge_pass_kernel(const uint64_t *__restrict__ src_keys, const uint32_t *__restrict__ src_perm,
                                                              uint64_t *__restrict__ dst_keys, uint32_t *__restrict__ dst_perm,
                                                              const MergeWork *__restrict__ work, uint32_t R, uint64_t col_mask)
{
    __shared__ uint64_t buf[kMergeTile];
    __shared__ uint32_t split[2];
    const MergeWork w = work[blockIdx.x];
    const uint32_t o0 = w.tile * kMergeTile, o1 = min(o0 + kMergeTile, w.len);
    const uint32_t base = o0 / (2u * R) * (2u * R);
    const uint32_t a_len = min(R, w.len - base), b_len = min(R, w.len - base - a_len);
    const uint64_t A = (uint64_t)w.lo + base, Bq = A + a_len;
    auto comp = [&](uint64_t i) { return ((src_keys[i] & col_mask) << 32) | (uint64_t)src_perm[i]; };
    // number of A elements among the first d outputs of the pair
    auto path = [&](uint32_t d) {
        uint32_t lo = d > b_len ? d - b_len : 0u, hi = min(d, a_len);
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (comp(A + mid) < comp(Bq + (d - 1u - mid))) lo = mid + 1u;
            else hi = mid;
        }
        return lo;
    };
    if (threadIdx.x == 0) split[0] = path(o0 - base);
    if (threadIdx.x == 64) split[1] = path(o1 - base);
    __syncthreads();
    const uint32_t a0 = split[0], a1 = split[1];
    const uint32_t b0 = (o0 - base) - a0, n_out = o1 - o0, na = a1 - a0, nb = n_out - na;
    for (uint32_t e = threadIdx.x; e < n_out; e += kThreads) buf[e] = e < na ? comp(A + a0 + e) : comp(Bq + b0 + (e - na));
    const uint64_t row_part = src_keys[w.lo] & ~col_mask;
    __syncthreads();
    // thread t: outputs 4 t .. 4 t + 3 of the tile
    const uint32_t d = min(4u * threadIdx.x, n_out);
    uint32_t lo = d > nb ? d - nb : 0u, hi = min(d, na);
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (buf[mid] < buf[na + (d - 1u - mid)]) lo = mid + 1u;
        else hi = mid;
    }
    uint32_t ia = lo, ib = d - lo;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t o = d + (uint32_t)k;
        if (o < n_out) {
            const bool take_a = ib >= nb || (ia < na && buf[ia] < buf[na + ib]);
            const uint64_t c = take_a ? buf[ia] : buf[na + ib];
            ia += take_a ? 1u : 0u;
            ib += take_a ? 0u : 1u;
            dst_keys[(uint64_t)w.lo + o0 + o] = row_part | (c >> 32);
            dst_perm[(uint64_t)w.lo + o0 + o] = (uint32_t)c;
        }
    }
}

// ---- long segments, second form (round 4): segmented LSD radix sort on the column bits --------------------------------------------
// The merge passes above move every element of every long segment ceil(log2(longest / cap)) times: R-MAT scale 22 x 1 has 568 M of its
// 690 M tasks in segments beyond a wave's capacity, the longest of 4 M tasks: 10 passes of 13.6 GB each (38 ms of a 120 ms product)
// after the piece sort.  Inside a segment only the column varies (jbits <= 21 bits whenever the narrow words apply): ceil(jbits / 7) stable
// counting passes order it, whatever its length, and the pieces need no sort of their own.
//   work item  = a tile of kSegRadixTile consecutive elements of ONE long segment (a workgroup of 8 waves; wave w owns elements 512 w ..)
//   count pass = digit histogram of the tile -> table[segment's first tile * RADIX + digit * tiles_of_segment + tile]: DIGIT-MAJOR inside
//                the segment, so that one exclusive scan of the whole table (minus its value at the segment's first entry) is the
//                place of (digit, tile) inside the segment
//   scatter    = rank of an element among the tile's elements of its digit: 64 elements per wave and round, peers by one ballot per
//                digit bit, earlier rounds through per-wave counters in LDS, earlier waves by a prefix over those counters; the tile
//                is laid out by digit in LDS first, so that neighbouring lanes write neighbouring words (runs of T / RADIX on average)
// What moves is the composite word (column << 32) | source position (8 bytes; the first pass forms it from the keys, the last pass
// writes keys and permutation apart); the passes alternate between the key array and its scratch partner, and the short segments are
// sorted by the wave / workgroup kernels INTO the array the last pass writes, so nothing is copied home.
constexpr uint32_t kSegRadixTile = 4096;
constexpr int kSegRadixThreads = 512, kSegRadixWaves = kSegRadixThreads / 64, kSegRadixBits = 7, kSegRadixBins = 1 << kSegRadixBits;
constexpr uint32_t kSegRadixRounds = kSegRadixTile / kSegRadixThreads;  // elements per lane

struct RadixWork {
    uint32_t lo, len, tile, first;  // segment start and length, tile inside the segment, index of the segment's first work item
    uint64_t row_part;              // the segment's block-row bits of the key: the composite words do not carry them
};
struct RadixTilesIn {
    SegClassify c;
    __device__ uint32_t operator()(uint64_t s) const
    {
        if (s >= c.nseg) return 0u;
        const uint32_t l = c.len(s);
        return c.is_long(l) ? (l + kSegRadixTile - 1u) / kSegRadixTile : 0u;
    }
};
struct RadixTilesOut {
    SegClassify c;
    const uint64_t *keys;
    uint64_t col_mask;
    RadixWork *work;
    uint32_t *stats;  // [0] work items, [1] longest segment, [2] elements in long segments
    __device__ void operator()(uint64_t s, uint32_t ex, uint32_t cnt) const
    {
        if (s == c.nseg) { stats[0] = ex; return; }
        if (!cnt) return;
        const uint32_t lo = (uint32_t)c.segs[s], len = c.len(s);
        const uint64_t row_part = keys[lo] & ~col_mask;
        for (uint32_t t = 0; t < cnt; t++) work[ex + t] = RadixWork{lo, len, t, ex, row_part};
        atomicMax(stats + 1, len);
        atomicAdd(stats + 2, len);
    }
};

template <bool FIRST>
__device__ __forceinline__ uint64_t radix_word(const uint64_t *__restrict__ src, uint64_t i, uint64_t col_mask)
{
    const uint64_t k = src[i];
    return FIRST ? (((k & col_mask) << 32) | (uint64_t)(uint32_t)i) : k;
}

template <bool FIRST>
__global__ __launch_bounds__(kSegRadixThreads) void segradix_count_kernel(const uint64_t *__restrict__ src, const RadixWork *__restrict__ work,
                                                                       uint32_t *__restrict__ table, int shift, int bits, uint64_t col_mask)
{
    __shared__ uint32_t h[kSegRadixBins];
    const RadixWork w = work[blockIdx.x];
    const uint32_t radix = 1u << bits, o0 = w.tile * kSegRadixTile, n_t = min(kSegRadixTile, w.len - o0);
    if (threadIdx.x < radix) h[threadIdx.x] = 0u;
    __syncthreads();
    const uint64_t base = (uint64_t)w.lo + o0;
#pragma unroll
    for (uint32_t r = 0; r < kSegRadixRounds; r++) {
        const uint32_t e = r * kSegRadixThreads + threadIdx.x;
        if (e < n_t) {
            const uint64_t k = src[base + e];
            const uint32_t col = FIRST ? (uint32_t)(k & col_mask) : (uint32_t)(k >> 32);
            atomicAdd(&h[(col >> shift) & (radix - 1u)], 1u);
        }
    }
    __syncthreads();
    const uint32_t ntiles = (w.len + kSegRadixTile - 1u) / kSegRadixTile;
    if (threadIdx.x < radix) table[(uint64_t)w.first * radix + (uint64_t)threadIdx.x * ntiles + w.tile] = h[threadIdx.x];
}

template <bool FIRST, bool LAST>
__global__ __launch_bounds__(kSegRadixThreads) void segradix_scatter_kernel(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst,
                                                                         uint32_t *__restrict__ perm_out, const RadixWork *__restrict__ work,
                                                                         const uint32_t *__restrict__ table, int shift, int bits, uint64_t col_mask)
{
    __shared__ uint64_t buf[kSegRadixTile];
    __shared__ uint32_t cnt[kSegRadixWaves][kSegRadixBins];
    __shared__ uint32_t dstart[kSegRadixBins], gbase[kSegRadixBins];
    const RadixWork w = work[blockIdx.x];
    const uint32_t radix = 1u << bits, o0 = w.tile * kSegRadixTile, n_t = min(kSegRadixTile, w.len - o0);
    const int lane = lane_id(), wave = wave_id();
    const uint64_t base = (uint64_t)w.lo + o0;
    for (uint32_t i = threadIdx.x; i < kSegRadixWaves * kSegRadixBins; i += kSegRadixThreads) (&cnt[0][0])[i] = 0u;
    // wave `wave` owns elements wave * 512 + r * 64 + lane: source order = (wave, round, lane)
    uint64_t c[kSegRadixRounds];
#pragma unroll
    for (uint32_t r = 0; r < kSegRadixRounds; r++) {
        const uint32_t e = (uint32_t)wave * (kSegRadixTile / kSegRadixWaves) + r * 64u + (uint32_t)lane;
        c[r] = e < n_t ? radix_word<FIRST>(src, base + e, col_mask) : ~0ull;
    }
    __syncthreads();
    uint32_t rank[kSegRadixRounds];
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (uint32_t r = 0; r < kSegRadixRounds; r++) {
        const uint32_t e = (uint32_t)wave * (kSegRadixTile / kSegRadixWaves) + r * 64u + (uint32_t)lane;
        const bool valid = e < n_t;
        const uint32_t d = (uint32_t)(c[r] >> (32 + shift)) & (radix - 1u);
        uint64_t m = __ballot(valid);
        for (int b = 0; b < bits; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            m &= bit ? bal : ~bal;
        }
        const uint32_t prior = cnt[wave][d];
        const uint32_t below = (uint32_t)__popcll(m & lt);
        rank[r] = prior + below;
        if (valid && (m >> lane) == 1ull) cnt[wave][d] = prior + below + 1u;  // the last lane of the digit's peers leaves the new count
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const uint32_t ntiles = (w.len + kSegRadixTile - 1u) / kSegRadixTile;
    if (threadIdx.x < radix) {
        uint32_t acc = 0u;
#pragma unroll
        for (int v = 0; v < kSegRadixWaves; v++) {
            const uint32_t t = cnt[v][threadIdx.x];
            cnt[v][threadIdx.x] = acc;
            acc += t;
        }
        dstart[threadIdx.x] = acc;
        const uint64_t tb = (uint64_t)w.first * radix;
        gbase[threadIdx.x] = table[tb + (uint64_t)threadIdx.x * ntiles + w.tile] - table[tb];
    } else if (threadIdx.x < (uint32_t)kSegRadixBins) {
        dstart[threadIdx.x] = 0u;
    }
    __syncthreads();
    if (wave == 0) {  // exclusive scan of the tile's digit totals: two digits per lane
        const uint32_t a = dstart[2 * lane], b = dstart[2 * lane + 1];
        const uint32_t inc = wave_inclusive_sum(a + b);
        dstart[2 * lane] = inc - a - b;
        dstart[2 * lane + 1] = inc - b;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kSegRadixRounds; r++) {
        const uint32_t e = (uint32_t)wave * (kSegRadixTile / kSegRadixWaves) + r * 64u + (uint32_t)lane;
        if (e < n_t) {
            const uint32_t d = (uint32_t)(c[r] >> (32 + shift)) & (radix - 1u);
            buf[dstart[d] + cnt[wave][d] + rank[r]] = c[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < kSegRadixRounds; r++) {
        const uint32_t i = r * kSegRadixThreads + threadIdx.x;
        if (i < n_t) {
            const uint64_t x = buf[i];
            const uint32_t d = (uint32_t)(x >> (32 + shift)) & (radix - 1u);
            const uint64_t dest = (uint64_t)w.lo + gbase[d] + (i - dstart[d]);
            if (LAST) {
                dst[dest] = w.row_part | (x >> 32);
                perm_out[dest] = (uint32_t)x;
            } else {
                dst[dest] = x;
            }
        }
    }
}

template <bool FIRST, bool LAST>
void segradix_pass(const uint64_t *src, uint64_t *dst, uint32_t *perm_out, const RadixWork *work, uint32_t n_work, DevBuf<uint32_t> &table, int shift, int bits,
                   uint64_t col_mask, hipStream_t st)
{
    const uint64_t entries = (uint64_t)n_work << bits;
    hipLaunchKernelGGL((segradix_count_kernel<FIRST>), dim3(n_work), dim3(kSegRadixThreads), 0, st, src, work, table.p, shift, bits, col_mask);
    BMSP_CHECK_LAUNCH();
    device_exclusive_scan<uint32_t>(PtrIn<uint32_t>{table.p}, PtrOut<uint32_t>{table.p}, entries, st);
    hipLaunchKernelGGL((segradix_scatter_kernel<FIRST, LAST>), dim3(n_work), dim3(kSegRadixThreads), 0, st, src, dst, perm_out, work, (const uint32_t *)table.p, shift,
                       bits, col_mask);
    BMSP_CHECK_LAUNCH();
}

struct CopyLoneKeys {  // segments of one element when the sorted keys are written to the scratch array
    const int *segs;
    uint32_t nseg;
    uint64_t n;
    const uint64_t *keys;
    uint64_t *keys_out;
    uint32_t *perm;
    __device__ void operator()(uint64_t s) const
    {
        if (s >= nseg) return;
        const uint64_t lo = (uint64_t)segs[s], hi = s + 1 < nseg ? (uint64_t)segs[s + 1] : n;
        if (hi == lo + 1) keys_out[lo] = keys[lo];
    }
};

struct CopyKeyPerm {
    const uint64_t *ks;
    const uint32_t *ps;
    uint64_t *kd;
    uint32_t *pd;
    __device__ void operator()(uint64_t i) const { kd[i] = ks[i]; pd[i] = ps[i]; }
};

// sorts the segments of (keys, perm) -- keys in place, perm[i] = source position of the element now at i.  `alt_keys` is a scratch array
// of n keys the caller provides (the ping-pong partner), `perm_alt` an empty buffer the merge passes allocate.  Returns false when the
// call is not handled (nothing modified).  *in_alt = true: the sorted KEYS sit in alt_keys (the caller flips its buffers) -- an odd number
// of radix passes (the short segments were sorted into alt_keys as well), or an odd number of merge passes over a list in which EVERY
// segment was long; *perm_in_alt says the same of the permutation (merge passes only: the radix passes write `perm` whatever their number).
template <typename W>
bool segsort_tasks_lds(uint64_t *keys, uint32_t *perm, uint64_t n, const int *segs, uint32_t nseg, int jbits, hipStream_t st, uint64_t *alt_keys = nullptr,
                       DevBuf<uint32_t> *perm_alt = nullptr, bool *in_alt = nullptr, bool *perm_in_alt = nullptr, int *long_mode = nullptr)
{
    if (in_alt) *in_alt = false;
    if (perm_in_alt) *perm_in_alt = false;
    if (nseg >= (1u << 21)) return false;
    const uint32_t cap = sizeof(W) == 4 ? 2 * kTaskWaveMax : kTaskWaveMax;  // 32-bit words: a wave holds a 4096-word segment in registers
    SegClassify cls{segs, nseg, n, cap};
    DevBuf<uint32_t> wave_list(nseg), block_list(nseg);
    HostScalar<uint64_t> tot;
    device_exclusive_scan<uint64_t>(cls, SegLists{cls, wave_list.p, block_list.p, tot.dev(), perm}, (uint64_t)nseg + 1, st);
    const uint64_t t = tot.wait(st);
    const uint32_t n_wave = (uint32_t)(t & 0x1fffffu), n_block = (uint32_t)((t >> 21) & 0x1fffffu), n_long = (uint32_t)(t >> 42);
    const uint64_t col_mask = (1ull << jbits) - 1ull;
    uint64_t *keys_out = keys;  // where the short segments' kernels write
    bool radix_done = false;
    const char *radix_env = getenv("BMSP_SEGSORT_RADIX");  // 0: merge passes, 1: radix passes whatever the lengths (tests, A/B runs)
    const int radix_mode = radix_env ? atoi(radix_env) : -1;
    if (n_long && alt_keys && in_alt && perm_in_alt && radix_mode != 0 && jbits <= 32) {
        // work items of the long segments, the longest of them and what they hold
        const uint64_t max_tiles = n / kSegRadixTile + (uint64_t)n_long + 1;
        DevBuf<RadixWork> work(max_tiles);
        DevBuf<uint32_t> stats(3);
        BMSP_HIP(hipMemsetAsync(stats.p, 0, 12, st));
        device_exclusive_scan<uint32_t>(RadixTilesIn{cls}, RadixTilesOut{cls, keys, col_mask, work.p, stats.p}, (uint64_t)nseg + 1, st);
        uint32_t hs[3];
        read_back_bytes(hs, stats.p, 12, st);
        const uint32_t n_work = hs[0], max_len = hs[1];
        // segments of up to four pieces stay with pieces + merge (two passes of 24 bytes per element after the piece sort; the counting
        // passes move 24 - 28 bytes per element and pass, two or three times, without a piece sort)
        if (n_work && (radix_mode == 1 || max_len > 4u * cap)) {
            const int passes = std::max(1, (jbits + kSegRadixBits - 1) / kSegRadixBits), bits = std::max(1, (jbits + passes - 1) / passes);
            DevBuf<uint32_t> table((uint64_t)n_work << bits);
            for (int p = 0; p < passes; p++) {
                const uint64_t *src = (p & 1) ? alt_keys : keys;
                uint64_t *dst = (p & 1) ? keys : alt_keys;
                const bool first = p == 0, last = p == passes - 1;
                if (first && last) segradix_pass<true, true>(src, dst, perm, work.p, n_work, table, p * bits, bits, col_mask, st);
                else if (first) segradix_pass<true, false>(src, dst, perm, work.p, n_work, table, p * bits, bits, col_mask, st);
                else if (last) segradix_pass<false, true>(src, dst, perm, work.p, n_work, table, p * bits, bits, col_mask, st);
                else segradix_pass<false, false>(src, dst, perm, work.p, n_work, table, p * bits, bits, col_mask, st);
            }
            if (passes & 1) { keys_out = alt_keys; *in_alt = true; }
            radix_done = true;
            if (long_mode) *long_mode = BMSP_SORT_LONG_RADIX;
            BMSP_HIP(hipStreamSynchronize(st));  // work list and table go back to the pool
        }
    }
    if (n_long && !radix_done) {
        if (!perm_alt || !alt_keys || getenv("BMSP_SEGSORT_NO_MERGE")) return false;
        if (long_mode) *long_mode = BMSP_SORT_LONG_MERGE;
        // 1. pieces of at most `cap` words; the pieces are sorted like ordinary segments
        const uint64_t max_pieces = (uint64_t)nseg + n / cap + 1;
        if (max_pieces >= (1u << 21)) return false;
        DevBuf<int> segs2(max_pieces);
        DevBuf<uint32_t> scal(3);
        BMSP_HIP(hipMemsetAsync(scal.p, 0, 12, st));
        device_exclusive_scan<uint32_t>(PiecesIn{cls, cap}, PiecesOut{cls, cap, segs2.p, scal.p, scal.p + 1, scal.p + 2}, (uint64_t)nseg + 1, st);
        uint32_t hs[3];
        read_back_bytes(hs, scal.p, 12, st);
        const uint32_t nseg2 = hs[0], max_len = hs[1];
        const bool all_long = (uint64_t)hs[2] == n;
        if (!segsort_tasks_lds<W>(keys, perm, n, segs2.p, nseg2, jbits, st)) return false;
        // 2. output tiles of the long segments
        SegClassify all{segs, nseg, n, cap};
        DevBuf<uint32_t> wtot(1);
        const uint64_t max_tiles = n / kMergeTile + (uint64_t)n_long + 1;
        DevBuf<MergeWork> work(max_tiles);
        device_exclusive_scan<uint32_t>(MergeTilesIn{all, cap}, MergeTilesOut{all, work.p, wtot.p}, (uint64_t)nseg + 1, st);
        const uint32_t n_work = read_back(wtot.p, st);
        // 3. merge passes, ping-pong between (keys, perm) and (alt_keys, perm2); an odd number of passes ends in the scratch pair, which is
        //    copied back (elements of short segments never leave the primary pair)
        perm_alt->alloc(n);
        uint64_t *ks = keys, *kd = alt_keys;
        uint32_t *ps = perm, *pd = perm_alt->p;
        int passes = 0;
        for (uint64_t R = cap; R < max_len; R *= 2) {
            hipLaunchKernelGGL(merge_pass_kernel, dim3(n_work), dim3(kThreads), 0, st, ks, ps, kd, pd, work.p, (uint32_t)R, col_mask);
            BMSP_CHECK_LAUNCH();
            std::swap(ks, kd); std::swap(ps, pd);
            passes++;
        }
        if ((passes & 1) && all_long && in_alt && perm_in_alt) {
            *in_alt = *perm_in_alt = true;  // every element went through the passes: the scratch pair holds the whole result
        } else if (passes & 1) {
            // the long segments' result sits in (alt_keys, perm_alt): bring it home tile by tile (a "merge" of runs as long as the segment copies)
            hipLaunchKernelGGL(merge_pass_kernel, dim3(n_work), dim3(kThreads), 0, st, ks, ps, kd, pd, work.p, 0x40000000u, col_mask);
            BMSP_CHECK_LAUNCH();
        }
        BMSP_HIP(hipStreamSynchronize(st));  // the work list goes back to the pool
        return true;
    }
    if (n_wave) {
        hipLaunchKernelGGL((segsort_tasks_wave_kernel<W>), dim3((n_wave + 3) / 4), dim3(kThreads), 0, st, (const uint64_t *)keys, keys_out, perm, segs, nseg, n,
                           wave_list.p, n_wave, col_mask);
        BMSP_CHECK_LAUNCH();
    }
    if (n_block) {
        hipLaunchKernelGGL((segsort_tasks_block_kernel<W>), dim3(n_block), dim3(512), 0, st, (const uint64_t *)keys, keys_out, perm, segs, nseg, n, block_list.p,
                           col_mask);
        BMSP_CHECK_LAUNCH();
    }
    if (keys_out != keys) {
        // segments of one element (no class, no kernel) and anything in front of the first segment: the key itself
        device_for_each(CopyLoneKeys{segs, nseg, n, keys, keys_out, perm}, (uint64_t)nseg + 1, st);
    }
    return true;
}

// The read-back-free path trusts a bound on the longest segment that the caller derives from the operands.  Should that bound ever be
// wrong, the kernel raises this word (pinned host memory, one per process) instead of leaving a segment unsorted; the SpGEMM looks at it
// after its final synchronise (segsort_check_violation) and fails the call.
static uint32_t *sort_violation_slot()
{
    static uint32_t *slot = [] {
        uint32_t *p = static_cast<uint32_t *>(host_slot_acquire());
        *(volatile uint32_t *)p = 0u;
        return p;
    }();
    return slot;
}
void segsort_check_violation()
{
    volatile uint32_t *p = sort_violation_slot();
    if (*p) {
        *p = 0u;
        fail(BMSP_ERR_LIMIT, "segmented sort: a task segment exceeded the capacity its bound promised (internal bound violated; result discarded)");
    }
}

bool segsort_tasks_by_column(PingPong<uint64_t> &keys, PingPong<uint64_t> &vals, uint64_t n, int jbits, hipStream_t st, uint64_t max_seg_bound,
                             uint64_t seg_count_bound, int *long_mode)
{
    if (long_mode) *long_mode = 0;
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
                               nseg_dev.p, n, col_mask, sort_violation_slot());
        else
            hipLaunchKernelGGL((segsort_tasks_direct_kernel<uint64_t>), dim3((unsigned)((waves + 3) / 4)), dim3(kThreads), 0, st, keys.cur, perm.p, segs.p,
                               nseg_dev.p, n, col_mask, sort_violation_slot());
        BMSP_CHECK_LAUNCH();
        device_for_each(GatherVals<uint64_t>{vals.cur, perm.p, vals.alt}, n, st);
        vals.flip();
        return true;
    }
    HostScalar<uint32_t> cnt;
    device_exclusive_scan<uint32_t>(RunHead{keys.cur, n, jbits}, EmitRunStarts{keys.cur, n, jbits, segs.p, cnt.dev(), nullptr}, n + 1, st);
    const uint32_t nseg = cnt.wait(st);
    DevBuf<uint32_t> perm_alt;
    bool in_alt = false, perm_in_alt = false;
    const bool ok = narrow ? segsort_tasks_lds<uint32_t>(keys.cur, perm.p, n, segs.p, nseg, jbits, st, keys.alt, &perm_alt, &in_alt, &perm_in_alt, long_mode)
                           : segsort_tasks_lds<uint64_t>(keys.cur, perm.p, n, segs.p, nseg, jbits, st, keys.alt, &perm_alt, &in_alt, &perm_in_alt, long_mode);
    if (!ok) return false;  // not handled: the caller takes the global sort
    if (in_alt) keys.flip();
    device_for_each(GatherVals<uint64_t>{vals.cur, perm_in_alt ? perm_alt.p : perm.p, vals.alt}, n, st);
    vals.flip();
    BMSP_HIP(hipStreamSynchronize(st));  // perm_alt goes back to the pool
    return true;
}

}  // namespace bmsp

BMSP_DEFINE_WARM(segsort)
