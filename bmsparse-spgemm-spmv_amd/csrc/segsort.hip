// segsort.hip -- stable segmented sort of (uint64 key, payload) pairs.
//
// Reference: bb_segsort<K,T>(keys, vals, n, segs, length), include/bb_segsort-master/bb_segsort.h:35-192, called
// by bmSparse_mult with K = uint64_t C keys, T = 16-byte tasks and one segment per block-row of A
// (src/bmSparse_SPGEMM.cu:973-1010).  bb_segsort is unstable; this one is stable.
//
// Version 1 (correctness baseline, also the fallback for very long segments): two stable LSD radix sorts of a
// permutation -- first by the key bits that actually vary, then by segment number -- followed by one gather.
#include "matrix.h"
#include "prims.hip.h"

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
    uint32_t *count;
    __device__ void operator()(uint64_t i, uint32_t ex) const
    {
        if (i == n) { *count = ex; return; }
        if (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) segs[ex] = (int)i;
    }
};
struct MaskLow {
    uint64_t *keys;
    const uint64_t *src;
    uint64_t mask;
    __device__ void operator()(uint64_t i) const { keys[i] = src[i] & mask; }
};
struct RestoreHigh {
    uint64_t *keys;        // sorted low parts
    const uint64_t *orig;  // original keys (any element of the same segment has the right high part)
    uint64_t mask;
    __device__ void operator()(uint64_t i) const { keys[i] = (orig[i] & ~mask) | keys[i]; }
};

}  // namespace

void segsort_u64(uint64_t *d_keys, void *d_vals, int val_bytes, int64_t n, const int *d_segs, int64_t num_segs, hipStream_t st)
{
    if (n < 0 || num_segs < 0) fail(BMSP_ERR_INVALID, "negative size");
    if (n == 0 || num_segs == 0) return;
    if (n >= (1ll << 31)) fail(BMSP_ERR_LIMIT, "segmented sort handles < 2^31 elements (int segment starts)");
    if (!d_keys || !d_segs) fail(BMSP_ERR_INVALID, "null pointer");
    if (!d_vals) segsort_impl<uint32_t>(d_keys, nullptr, (uint64_t)n, d_segs, (uint32_t)num_segs, st);
    else if (val_bytes == 4) segsort_impl<uint32_t>(d_keys, (uint32_t *)d_vals, (uint64_t)n, d_segs, (uint32_t)num_segs, st);
    else if (val_bytes == 8) segsort_impl<uint64_t>(d_keys, (uint64_t *)d_vals, (uint64_t)n, d_segs, (uint32_t)num_segs, st);
    else if (val_bytes == 16) segsort_impl<Pair16>(d_keys, (Pair16 *)d_vals, (uint64_t)n, d_segs, (uint32_t)num_segs, st);
    else fail(BMSP_ERR_INVALID, "val_bytes must be 4, 8 or 16");
}

void segsort_tasks_by_column(PingPong<uint64_t> &keys, PingPong<uint64_t> &vals, uint64_t n, int jbits, int ibits, hipStream_t st)
{
    (void)ibits;
    if (n >= (1ull << 31)) fail(BMSP_ERR_LIMIT, "segmented sort handles < 2^31 tasks");
    // segments = runs of equal block-row (reference: :982-1004)
    DevBuf<int> segs(n);
    DevBuf<uint32_t> cnt(1);
    device_exclusive_scan<uint32_t>(RunHead{keys.cur, n, jbits}, EmitRunStarts{keys.cur, n, jbits, segs.p, cnt.p}, n + 1, st);
    uint32_t nseg = read_back(cnt.p, st);
    // sort the column part inside each segment, then put the row part back
    const uint64_t mask = (1ull << jbits) - 1ull;
    device_for_each(MaskLow{keys.alt, keys.cur, mask}, n, st);
    segsort_u64(keys.alt, vals.cur, 8, (int64_t)n, segs.p, nseg, st);
    device_for_each(RestoreHigh{keys.alt, keys.cur, mask}, n, st);
    keys.flip();
    BMSP_HIP(hipStreamSynchronize(st));
}

}  // namespace bmsp
