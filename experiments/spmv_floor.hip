// The access pattern of the SpMV sweep with everything else stripped: how fast can MI355X do "lane per tile: three 8-byte
// streams (key, bitmap, offset) -> one value load at the streamed offset -> one random 4-byte gather of x -> one product
// accumulated per block-row" on the webbase-1M-like shape (2.07 M tiles, 1.5 values per tile taken as 1, x = 4 MB)?
// No bitmap decode, no LDS atomics, no plan: the floor of the formulation, against which DESIGN.md reads the 38 us sweep.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void floor_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ bmps, const uint64_t *__restrict__ offs,
                                                    const float *__restrict__ vals, const float *__restrict__ x, float *__restrict__ y, uint32_t n, int mode)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i], b = bmps[i], o = offs[i];
    float acc = (float)(uint32_t)(b & 1);
    if (mode >= 1) acc += vals[o];                                   // value at the streamed offset (near-contiguous across lanes)
    if (mode == 2) acc *= x[(uint32_t)(k & 0xffffffffu)];            // random gather
    if (mode >= 3) {  // the same gather through a buffer descriptor with cache-policy bits (aux: 1 = sc0, 2 = nt, 16 = sc1)
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, 1u << 22, 0x00020000);
        const uint32_t off = (uint32_t)(k & 0xffffffffu) * 4u;
        uint32_t v;
        switch (mode) {
        case 3: v = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); break;
        case 4: v = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 1); break;
        case 5: v = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 2); break;
        case 6: v = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 3); break;
        case 7: v = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 16); break;
        default: v = __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 17); break;
        }
        acc *= __builtin_bit_cast(float, v);
    }
    // eight tiles share an output (stand-in for the block-row reduction): DPP-free, one store per 8 lanes
    float s = acc;
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if ((threadIdx.x & 7) == 0) y[i >> 3] = s;
}

int main()
{
    const uint32_t n = 2073692, ncols = 1u << 20, copies = 9;
    std::vector<uint64_t> hk(n), hb(n), ho(n);
    uint64_t z = 12345;
    for (uint32_t i = 0; i < n; i++) {
        z = z * 6364136223846793005ull + 1442695040888963407ull;
        hk[i] = ((uint64_t)(i / 2) << 32) | (uint32_t)((z >> 33) % ncols);
        hb[i] = z | 1;
        ho[i] = (uint64_t)i + (i >> 1);  // 1.5 values per tile
    }
    uint64_t *dk[copies], *db[copies], *dof[copies];
    float *dv[copies], *dx, *dy;
    for (uint32_t c = 0; c < copies; c++) {
        hipMalloc((void **)&dk[c], 8ull * n); hipMalloc((void **)&db[c], 8ull * n); hipMalloc((void **)&dof[c], 8ull * n); hipMalloc((void **)&dv[c], 4ull * (n + n / 2 + 8));
        hipMemcpy(dk[c], hk.data(), 8ull * n, hipMemcpyHostToDevice); hipMemcpy(db[c], hb.data(), 8ull * n, hipMemcpyHostToDevice);
        hipMemcpy(dof[c], ho.data(), 8ull * n, hipMemcpyHostToDevice); hipMemset(dv[c], 0, 4ull * (n + n / 2 + 8));
    }
    hipMalloc((void **)&dx, 4ull * ncols); hipMemset(dx, 0, 4ull * ncols);
    hipMalloc((void **)&dy, 4ull * (n / 8 + 1));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 9; mode++) {
        for (int it = 0; it < 20; it++) hipLaunchKernelGGL(floor_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dk[it % copies], db[it % copies], dof[it % copies], dv[it % copies], dx, dy, n, mode);
        hipEventRecord(e0, 0);
        const int reps = 200;
        for (int it = 0; it < reps; it++) hipLaunchKernelGGL(floor_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dk[it % copies], db[it % copies], dof[it % copies], dv[it % copies], dx, dy, n, mode);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (%s): %.2f us per sweep over %u tiles\n", mode, mode == 0 ? "three streams only" : mode == 1 ? "+ value load" : mode == 2 ? "+ value load + x gather" : "... x gather as a buffer load, aux bits 0 / sc0 / nt / sc0+nt / sc1 / sc1+sc0 for modes 3..8", ms * 1e3 / reps, n);
    }
    return 0;
}
