// blockmac_f32.hip -- T_7 for fp32 operands with the reference's V15 numerics on the matrix cores: v_mfma_f32_16x16x4_f32.
//
// Reference: multiplyV15<float, float> (src/bmSparse_SPGEMM.cu:204-291): every lane owns one C element and runs, task after task in
// list order, the chain  sum = fmaf(A(i, kk), B(kk, j), sum)  for kk = 0 .. 7 (:269-273; nvcc contracts `sum += a * b` of floats into
// one fused multiply-add).  The fp32 MFMA of gfx950 is exact fp32 and accumulates its four k in ascending order as that same chain
// (MI355X_MICROARCH.md, "F32 (f32 in) ... exact f32 (== fmaf chain, bitwise)") -- bmsp_selftest_mfma_f32_chain checks it on the
// hardware against a host fmaf chain, and the launcher takes this kernel only where that test passes -- so a task is two
// instructions (kk 0-3, then 4-7) instead of 64 lanes x 8 v_fma plus their LDS staging, and the values stay bit-identical with
// the oracle's.  Peak is the fp32 vector rate (157 TFLOP/s): the gain is in instructions per task, not in flops.
//
// Schedule: the direct kernel's (blockmac32.hip).  A wave owns a contiguous range of C tiles (equal task quotas) and takes them in
// pairs packed block-diagonally -- rows / columns 0-7 = tile X, 8-15 = tile Y; lane l = (k slot l >> 4, tile (l >> 3) & 1, line
// l & 7) loads element (line, kk) of its tile's current A tile and element (kk, line) of the B tile from the dense fp32 copies
// (256 B per tile, built once per matrix) for kk = k slot and kk = 4 + k slot.  A pair runs max(n_X, n_Y) steps; a tile that ends
// earlier is stored at ITS last step (the padding steps would add fmaf(0, 0, sum), which turns a -0 sum into +0).
#include "mac_common.hip.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace bmsp {
namespace {

struct MacF32Args {
    const uint64_t *tasks;
    uint32_t n_tasks;
    const uint32_t *task_begin, *c_of_wave;
    const float *a_lanes, *b_lanes;  // the operands' tiles in MFMA lane order (ExpandLaneOrder below)
    uint32_t a_lanes_bytes, b_lanes_bytes;
    const uint64_t *c_bmps, *c_offs;
    float *c_vals;
    uint32_t c_size, quota;
};

__device__ __forceinline__ uint32_t rl32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }

// A tile in the order the MFMA lanes consume it: float 2 * (line * 4 + kq) + h holds element (line, kk = kq + 4 h) -- line = row of a
// normal tile, column of a transposed (B) tile; lane (kq, line) of the kernel fetches its two k (one per instruction of a task) with
// ONE 8-byte load, 32 lanes cover the tile's 256 bytes contiguously.
struct ExpandLaneOrder {
    const uint64_t *bmps, *offsets;
    const float *values;
    float *out;
    __device__ void operator()(uint64_t i) const
    {
        const uint64_t b = i >> 6;
        const int e = (int)(i & 63u);
        const int line = e >> 3, kq = (e >> 1) & 3, h = e & 1;
        const int p = line * 8 + kq + 4 * h;
        const uint64_t bm = bmps[b];
        out[i] = tile_has(bm, p) ? values[offsets[b] + (uint64_t)tile_rank(bm, p)] : 0.f;
    }
};


typedef uint32_t u32x2w_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct F32Pos {
    uint32_t c;        // first tile of the pair
    uint32_t s, steps;  // first step of the block of kU steps; steps of the pair
    uint32_t tb, n;    // per lane: first task and task count of the lane's tile of the pair
    uint32_t n0, n1;   // task counts of both tiles (wave-uniform)
    uint64_t cb, co;   // per lane: C bitmap and value offset of the tile this lane stores (requested when the pair is entered)
};
template <int kU>
struct F32Ops {
    f32x2_t a[kU], b[kU];
};
template <int kU>
struct F32Tasks {
    uint64_t t[kU];
};

// kU = tasks of a C tile whose operands a wave requests together (one memory round trip per kU steps)
template <int kU>
__global__ __launch_bounds__(kThreads) void block_mac_f32_mfma_kernel(MacF32Args g)
{
    const int w = wave_id(), lane = lane_id();
    const int r = lane & 7, sel = (lane >> 3) & 1, kq = lane >> 4;
    const rsrc_t rda = make_rsrc(g.a_lanes, g.a_lanes_bytes), rdb = make_rsrc(g.b_lanes, g.b_lanes_bytes);
    const rsrc_t rtk = make_rsrc(g.tasks, g.n_tasks * 8u);
    uint32_t wg;
    {
        const uint32_t G = gridDim.x, q = G / 8, rm = G % 8, x = blockIdx.x % 8;
        wg = (x < rm ? x * (q + 1) : rm * (q + 1) + (x - rm) * q) + blockIdx.x / 8;
    }
    const uint32_t wv = wg * 4 + w;
    uint32_t rs = g.c_size;
    if (lane < 2) {
        const uint64_t t = (uint64_t)(wv + lane) * g.quota;
        if (t < g.n_tasks) {
            const uint32_t c = g.c_of_wave[t >> 6];
            rs = g.task_begin[c] == (uint32_t)t ? c : c + 1;
        }
    }
    const uint32_t c0 = rl32(rs, 0), ce = rl32(rs, 1);
    if (c0 >= ce) return;
    const uint32_t lane_off = (uint32_t)((r * 4 + kq) * 8);  // this lane's 8 bytes inside a tile
    const int dt = lane >> 5;
    const bool d_lane = dt == sel;

    auto enter = [&](uint32_t c) {
        F32Pos p;
        p.c = c;
        p.s = 0;
        const uint32_t t0 = g.task_begin[c], t1 = g.task_begin[c + 1], t2 = c + 1 < ce ? g.task_begin[c + 2] : t1;
        p.n0 = t1 - t0; p.n1 = t2 - t1;
        p.steps = max(p.n0, p.n1);
        p.tb = sel ? t1 : t0;
        p.n = sel ? p.n1 : p.n0;
        p.cb = 0; p.co = 0;
        if (d_lane && c + (uint32_t)dt < ce) { p.cb = g.c_bmps[c + (uint32_t)dt]; p.co = g.c_offs[c + (uint32_t)dt]; }
        return p;
    };
    auto advance = [&](const F32Pos &p) {
        F32Pos q = p;
        q.s = p.s + (uint32_t)kU;
        if (q.s >= p.steps) {
            if (p.c + 2 < ce) q = enter(p.c + 2);
            else { q.c = ce; q.s = 0; q.steps = 0; q.tb = 0; q.n = 0; q.n0 = 0; q.n1 = 0; q.cb = 0; q.co = 0; }  // past the end: its loads are masked off
        }
        return q;
    };
    auto load_tasks = [&](const F32Pos &p) {
        F32Tasks<kU> k;
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const u32x2w_t t = __builtin_amdgcn_raw_buffer_load_b64(rtk, (p.c < ce && p.s + (uint32_t)u < p.n) ? (p.tb + p.s + (uint32_t)u) * 8u : kOob, 0, 0);
            k.t[u] = ((uint64_t)t[1] << 32) | t[0];
        }
        return k;
    };
    auto load_ops = [&](const F32Pos &p, const F32Tasks<kU> &k) {
        F32Ops<kU> o;
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const bool on = p.c < ce && p.s + (uint32_t)u < p.n;
            o.a[u] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rda, on ? ((uint32_t)(k.t[u] >> 32) << 8) + lane_off : kOob, 0, 0));
            o.b[u] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rdb, on ? ((uint32_t)k.t[u] << 8) + lane_off : kOob, 0, 0));
        }
        return o;
    };

    F32Pos pa = enter(c0);
    F32Tasks<kU> tka = load_tasks(pa);
    F32Pos pb = advance(pa);
    F32Tasks<kU> tkb = load_tasks(pb);
    F32Ops<kU> oa = load_ops(pa, tka);
    float4_t acc = {0.f, 0.f, 0.f, 0.f};

    while (pa.c < ce) {
        // operands of the next block of steps, task words of the one after; then this block's instructions
        const F32Ops<kU> ob = load_ops(pb, tkb);
        const F32Pos pc = advance(pb);
        const F32Tasks<kU> tkc = load_tasks(pc);
#pragma unroll
        for (int u = 0; u < kU; u++) {
            const uint32_t done = pa.s + (uint32_t)u + 1u;
            if (done <= pa.steps) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(oa.a[u][0], oa.b[u][0], acc, 0, 0, 0);  // kk 0 .. 3
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(oa.a[u][1], oa.b[u][1], acc, 0, 0, 0);  // kk 4 .. 7
                // a tile is complete after its own last task: compacted store by its C bitmap
                if (done == pa.n0 || done == pa.n1) {
                    const uint32_t ct = pa.c + (uint32_t)dt;
                    const uint32_t n_mine = dt ? pa.n1 : pa.n0;
                    if (d_lane && ct < ce && done == n_mine) {
                        const uint64_t cb = pa.cb, co = pa.co;
                        const uint32_t row0 = 4u * (uint32_t)((lane >> 4) & 1);
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const uint32_t p = (row0 + (uint32_t)i) * 8u + (uint32_t)r;
                            if ((cb >> (63u - p)) & 1ull) g.c_vals[co + (uint64_t)__popcll(cb >> 1 >> (63u - p))] = acc[i];
                        }
                    }
                }
                if (done == pa.steps) acc = float4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        pa = pb; pb = pc;
        tkb = tkc;
        oa = ob;
    }
}

// hardware check of the accumulation order: D = C + A * B for random fp32 operands must equal, bit for bit, the chain
// fmaf(A[i][3], B[3][j], fmaf(A[i][2], B[2][j], fmaf(A[i][1], B[1][j], fmaf(A[i][0], B[0][j], C[i][j]))))
__global__ void mfma_f32_selftest_kernel(const float *a, const float *b, const float *c, float *d)
{
    const int lane = (int)threadIdx.x;
    float4_t acc;
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = c[(4 * (lane >> 4) + i) * 16 + (lane & 15)];
    // two instructions back to back, as the kernel issues them: k 0-3 then k 4-7 of a 16 x 8 operand
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(lane & 15) * 8 + (lane >> 4)], b[(lane >> 4) * 16 + (lane & 15)], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(lane & 15) * 8 + 4 + (lane >> 4)], b[(4 + (lane >> 4)) * 16 + (lane & 15)], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++) d[(4 * (lane >> 4) + i) * 16 + (lane & 15)] = acc[i];
}

int g_f32_chain_ok = -1;  // -1 not tested yet, 0 differs, 1 the MFMA is the fmaf chain on this device

}  // namespace

// number of result elements (of 16 x 16 x several trials) that differ from the host's k-ascending fmaf chain.  cancel = false: operands of
// ordinary magnitude (every product and every partial sum a normal number).  cancel = true: every PRODUCT is a normal number (~2^-123) but
// the signs alternate, so the partial sums cancel into the subnormal range -- the corner the strip kernel's exponent guard has to know about
// ("a*b + c < 2^-126 while both products are normal").
int mfma_f32_selftest(hipStream_t st, bool cancel)
{
    DevBuf<float> da(128), db(128), dc(256), dd(256);
    int bad = 0;
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() {  // values of mixed magnitude and sign, so that every addition rounds
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int e = (int)((s >> 40) % 9) - 4;
        return (float)std::ldexp((double)((int64_t)(s & 0xffffff) - 0x800000) / 8388608.0, e);
    };
    auto near_one = [&]() {  // 1 + a few units in the last places
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        return 1.0 + (double)(s & 0xff) / 8388608.0;
    };
    for (int trial = 0; trial < 8; trial++) {
        float a[128], b[128], c[256], d[256];
        if (!cancel) {
            for (float &x : a) x = rnd();
            for (float &x : b) x = rnd();
            for (float &x : c) x = trial == 0 ? 0.f : rnd();
            if (trial == 1) c[5] = -0.0f;
        } else {
            for (int i = 0; i < 16; i++)
                for (int k = 0; k < 8; k++) a[i * 8 + k] = (float)std::ldexp(((k & 1) ? -1.0 : 1.0) * near_one(), -60 - (trial & 1));
            for (float &x : b) x = (float)std::ldexp(near_one(), -63);
            for (float &x : c) x = trial < 2 ? 0.f : (float)std::ldexp(near_one() - 1.0, -126 - (trial & 3));  // zero, or subnormal starts
        }
        BMSP_HIP(hipMemcpyAsync(da.p, a, sizeof a, hipMemcpyHostToDevice, st));
        BMSP_HIP(hipMemcpyAsync(db.p, b, sizeof b, hipMemcpyHostToDevice, st));
        BMSP_HIP(hipMemcpyAsync(dc.p, c, sizeof c, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(mfma_f32_selftest_kernel, dim3(1), dim3(64), 0, st, da.p, db.p, dc.p, dd.p);
        BMSP_CHECK_LAUNCH();
        BMSP_HIP(hipMemcpyAsync(d, dd.p, sizeof d, hipMemcpyDeviceToHost, st));
        BMSP_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < 16; i++)
            for (int j = 0; j < 16; j++) {
                float ref = c[i * 16 + j];
                for (int k = 0; k < 8; k++) ref = std::fmaf(a[i * 8 + k], b[k * 16 + j], ref);
                if (memcmp(&ref, &d[i * 16 + j], 4) != 0) bad++;
            }
    }
    return bad;
}

int g_f32_cancel_ok = -1;  // 1: the matrix pipe is the fmaf chain also where normal products cancel into subnormal sums
// smallest sum of the operands' smallest biased exponents the fp32 matrix-core kernels accept: 128 = every product a normal number;
// where cancellation into subnormals is NOT the fmaf chain's, 46 more bits of room (the products of a C entry cannot cancel below
// 2^-126 unless they lose more than two mantissas' worth of bits)
int mac_f32_exp_floor(hipStream_t st)
{
    if (g_f32_cancel_ok < 0) g_f32_cancel_ok = mfma_f32_selftest(st, true) == 0 ? 1 : 0;
    return g_f32_cancel_ok == 1 ? 128 : 128 + 46;
}

// fp32 matrices: every tile expanded to 64 floats in MFMA lane order (256 B per block), cached per matrix like the other operand forms
void ensure_lane_tiles(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->lane_tiles) return;
    if (m->dtype != BMSP_F32) fail(BMSP_ERR_INVALID, "lane-ordered tile copies exist for fp32 matrices");
    m->lane_tiles = pool_alloc(256 * (size_t)(m->block_num ? m->block_num : 1) + 64);
    if (m->block_num) device_for_each(ExpandLaneOrder{m->bmps, m->offsets, (const float *)m->values, (float *)m->lane_tiles}, (uint64_t)m->block_num * 64, st);
}

bool mac_f32_mfma_usable(hipStream_t st)
{
    if (g_f32_chain_ok < 0) g_f32_chain_ok = mfma_f32_selftest(st, false) == 0 ? 1 : 0;
    return g_f32_chain_ok == 1;
}

// fp32 operands through the MFMA kernel; false = not taken (the caller launches the vector-ALU kernel)
bool launch_mac_f32_mfma(const uint64_t *tasks, uint64_t n_tasks, const uint32_t *task_begin, const uint32_t *c_of_wave, bmsp_matrix_s *A,
                         bmsp_matrix_s *B, bmsp_matrix_s *C, hipStream_t st)
{
    // Opt-in (BMSP_MAC_F32MFMA=1).  Measured on MI355X the kernel is bound by the 256 bytes it reads per operand tile and beats neither
    // vector-ALU kernel: FEM-like product (3.7 values per tile) T_7 1.71 ms vs 1.62 ms for the element-gather kernel, full-tile banded
    // 127 us vs 96 us for the dense-staged one (block-MAC log round 3).  Kept, with its parity tests, as the measured answer to
    // "does the fp32 MFMA help V15".
    const char *force = getenv("BMSP_MAC_F32MFMA");
    if (!force || force[0] != '1') return false;
    if (A->dtype != BMSP_F32 || n_tasks >= (1ull << 29) || A->block_num >= (1ll << 24) || B->block_num >= (1ll << 24)) return false;
    const uint32_t cs = (uint32_t)C->block_num;
    if (!mac_f32_mfma_usable(st)) return false;
    // (the instruction is the fmaf chain only while every product is a normal number: see mac_strip_operands_ok)
    ensure_finite_flag(A, st);
    ensure_finite_flag(B, st);
    if (A->values_finite != 1 || B->values_finite != 1 || A->f32_exp_min + B->f32_exp_min < mac_f32_exp_floor(st) || A->f32_exp_max + B->f32_exp_max > 354) return false;
    ensure_lane_tiles(A, st);
    ensure_lane_tiles(B, st);
    MacF32Args g{};
    g.tasks = tasks; g.n_tasks = (uint32_t)n_tasks; g.task_begin = task_begin; g.c_of_wave = c_of_wave;
    g.a_lanes = (const float *)A->lane_tiles; g.a_lanes_bytes = (uint32_t)(A->block_num * 256);
    g.b_lanes = (const float *)B->lane_tiles; g.b_lanes_bytes = (uint32_t)(B->block_num * 256);
    g.c_bmps = C->bmps; g.c_offs = C->offsets; g.c_vals = (float *)C->values; g.c_size = cs;
    const char *qenv = getenv("BMSP_MAC_QUOTA");
    uint64_t quota = qenv ? (uint64_t)atoll(qenv) : (n_tasks + 32767) / 32768;
    quota = std::max<uint64_t>(256, (quota + 63) / 64 * 64);
    if (qenv) quota = std::max<uint64_t>(64, ((uint64_t)atoll(qenv) + 63) / 64 * 64);
    const uint64_t waves = (n_tasks + quota - 1) / quota;
    g.quota = (uint32_t)quota;
    const char *ue = getenv("BMSP_MAC_F32_U");  // experiment switch: steps per pipeline block
    const int du = ue ? atoi(ue) : 2;
    const dim3 grid((uint32_t)((waves + 3) / 4));
    if (du == 4) hipLaunchKernelGGL(block_mac_f32_mfma_kernel<4>, grid, dim3(kThreads), 0, st, g);
    else if (du == 8) hipLaunchKernelGGL(block_mac_f32_mfma_kernel<8>, grid, dim3(kThreads), 0, st, g);
    else hipLaunchKernelGGL(block_mac_f32_mfma_kernel<2>, grid, dim3(kThreads), 0, st, g);
    BMSP_CHECK_LAUNCH();
    return true;
}

}  // namespace bmsp

BMSP_DEFINE_WARM(blockmac_f32)
