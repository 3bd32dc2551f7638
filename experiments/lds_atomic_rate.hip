// micro-benchmark: throughput of LDS operations on gfx950 (cycles per wave-instruction per CU), to size the SpMV design
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int ACTIVE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ float s[256 * 9];
    __shared__ unsigned su[256 * 9];
    __shared__ unsigned long long sl[256 * 9];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 256 * 9; i += 256) { s[i] = 0; su[i] = 0; sl[i] = 0; }
    __syncthreads();
    float acc = 0;
    unsigned uacc = 0;
    const bool on = lane < ACTIVE;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int idx = j * 256 + tid;  // conflict-free, private per lane
            if (MODE == 0) { if (on) __hip_atomic_fetch_add(&s[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
            if (MODE == 1) { if (on) __hip_atomic_fetch_add(&su[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
            if (MODE == 2) { if (on) uacc += __hip_atomic_fetch_add(&su[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
            if (MODE == 3) { if (on) s[idx] = acc + j; }
            if (MODE == 4) { if (on) acc += s[idx]; }
            if (MODE == 5) { if (on) acc += __hip_atomic_fetch_add(&s[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
            if (MODE == 7) { if (on) __hip_atomic_fetch_or(&sl[idx], 1ull << (it & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (MODE == 8) { if (on) __hip_atomic_fetch_or(&su[idx], 1u << (it & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (MODE == 9) { if (on) atomicOr(&sl[(idx * 37) % (256 * 9)], 1ull << (it & 63)); }
            if (MODE == 10) { if (on) atomicAdd(&su[(idx * 37) % (256 * 9)], 1u); }
            if (MODE == 6) { if (on) __hip_atomic_fetch_add(&s[(j * 256 + (tid & ~7))], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }  // 8-way same address
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + tid] = acc + s[tid] + uacc + su[tid] + (float)sl[tid];
}
template <int MODE, int ACTIVE>
void run(const char *name, float *d, int blocks)
{
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE, ACTIVE>), dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE, ACTIVE>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // per CU: blocks/256 resident blocks... we launch 256*4 blocks = 4 per CU (16 waves/CU)
    double wave_instrs_per_cu = (double)blocks / 256.0 * 4 /*waves per block*/ * iters * 8;
    double cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s active=%2d  %8.3f ms  %7.1f cycles per wave-instruction per CU (at 2.4 GHz)\n", name, ACTIVE, ms, cycles / wave_instrs_per_cu);
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4 * 256 * 4);
    const int blocks = 256 * 4;
    run<0, 64>("ds_add_f32 (no return)", d, blocks);
    run<0, 8>("ds_add_f32 (no return)", d, blocks);
    run<0, 1>("ds_add_f32 (no return)", d, blocks);
    run<5, 64>("ds_add_rtn_f32", d, blocks);
    run<6, 64>("ds_add_f32 8 lanes per address", d, blocks);
    run<1, 64>("ds_add_u32 (no return)", d, blocks);
    run<1, 8>("ds_add_u32 (no return)", d, blocks);
    run<2, 64>("ds_add_rtn_u32", d, blocks);
    run<2, 8>("ds_add_rtn_u32", d, blocks);
    run<7, 64>("ds_or_b64 (no return)", d, blocks);
    run<7, 8>("ds_or_b64 (no return)", d, blocks);
    run<8, 64>("ds_or_b32 (no return)", d, blocks);
    run<9, 64>("atomicOr u64, scattered slots", d, blocks);
    run<10, 64>("atomicAdd u32, scattered slots", d, blocks);
    run<3, 64>("ds_write_b32", d, blocks);
    run<4, 64>("ds_read_b32", d, blocks);
    return 0;
}
