// calibration of rocprofv3 FETCH_SIZE on gfx950 for the access shapes of the SpMV sweep: 8-byte-per-lane coalesced streams
// (the three metadata arrays), 16-byte-per-lane streams (the guide's reference shape) and 4-byte random gathers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void stream8(const uint64_t *p, size_t n, uint64_t *out)
{
    uint64_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void stream16(const uint4 *p, size_t n, uint64_t *out)
{
    uint64_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0x1234567) out[0] = acc;
}
__global__ void gather4(const uint32_t *p, size_t table, size_t n, uint64_t *out)
{
    uint64_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = i * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
        acc += p[z % table];
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main()
{
    const size_t bytes = 1ull << 30;  // 1 GiB > 256 MiB Infinity Cache
    void *d; uint64_t *out;
    hipMalloc(&d, bytes); hipMalloc(&out, 8); hipMemset(d, 1, bytes);
    hipLaunchKernelGGL(stream8, dim3(256 * 8), dim3(256), 0, 0, (const uint64_t *)d, bytes / 8, out);
    hipLaunchKernelGGL(stream16, dim3(256 * 8), dim3(256), 0, 0, (const uint4 *)d, bytes / 16, out);
    hipLaunchKernelGGL(gather4, dim3(256 * 8), dim3(256), 0, 0, (const uint32_t *)d, bytes / 4, (size_t)1 << 24, out);
    hipDeviceSynchronize();
    printf("bytes streamed per stream kernel: %zu ; gathers: %zu x 4 B\n", bytes, (size_t)1 << 24);
    return 0;
}
