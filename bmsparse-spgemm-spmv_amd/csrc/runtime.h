// runtime.h -- error plumbing, pooled device memory and small RAII helpers shared by the HIP sources.
#ifndef BMSP_RUNTIME_H_
#define BMSP_RUNTIME_H_

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <chrono>
#include <stdexcept>
#include <string>
#include "../../include/bmsp.h"

namespace bmsp {

// thrown inside the library, translated to a bmsp_status at the C boundary
struct Error : std::runtime_error {
    int status;
    Error(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};

// a product has more candidate block pairs than one task list indexes (32 bits): bmsp_spgemm answers by running it in panels
struct TaskRangeExceeded : Error {
    using Error::Error;
};

[[noreturn]] void fail(int status, const char *fmt, ...);
void set_last_error(const std::string &msg);

#define BMSP_HIP(call)                                                                                     \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess)                                                                             \
            ::bmsp::fail(BMSP_ERR_HIP, "%s failed at %s:%d: %s", #call, __FILE__, __LINE__,                \
                         hipGetErrorString(e__));                                                          \
    } while (0)

#define BMSP_CHECK_LAUNCH() BMSP_HIP(hipGetLastError())

// Pooled device allocations.  hipMalloc/hipFree synchronise the device; the SpGEMM pipeline needs a dozen
// temporaries per call, so freed blocks are kept (bucketed by rounded size) and handed out again.
constexpr size_t kPoolSlack = 64;
void *pool_alloc(size_t bytes);
void pool_free(void *p);
void pool_trim();
bool pool_owns(void *p);

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void alloc(size_t count)
    {
        release();
        n = count;
        p = static_cast<T *>(pool_alloc((count ? count : 1) * sizeof(T)));
    }
    void release()
    {
        if (p) pool_free(p);
        p = nullptr;
        n = 0;
    }
    T *take()
    {
        T *q = p;
        p = nullptr;
        n = 0;
        return q;
    }
    T *data() const { return p; }
    size_t size() const { return n; }
};

inline hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }

// device-side timing of pipeline stages: hipEvents recorded on the operator's stream at stage boundaries and read
// once at the end, so timing itself adds no host synchronisation between stages
struct StageTimer {
    hipStream_t st;
    bool on;
    static constexpr int kMax = 24;
    hipEvent_t ev[kMax];
    int stage_of[kMax];
    double host_us[kMax];  // host clock at the mark (BMSP_HOST_TIMES=1 prints them: where the HOST spends a call)
    int n = 0;
    StageTimer(hipStream_t s, bool enable) : st(s), on(enable) {}
    ~StageTimer()
    {
        for (int i = 0; i < n; i++) (void)hipEventDestroy(ev[i]);
    }
    // marks a boundary: the time since the previous mark is charged to `stage` (-1 = not charged)
    void mark(int stage)
    {
        if (!on || n >= kMax) return;
        BMSP_HIP(hipEventCreate(&ev[n]));
        BMSP_HIP(hipEventRecord(ev[n], st));
        stage_of[n] = stage;
        host_us[n] = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3;
        n++;
    }
    void print_host_times(const char *what) const
    {
        if (!on || n < 1) return;
        const double now = (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e-3;
        fprintf(stderr, "[host times] %s:", what);
        for (int i = 1; i < n; i++) fprintf(stderr, " ->%d %.0f", stage_of[i], host_us[i] - host_us[i - 1]);
        fprintf(stderr, " | since last mark %.0f us\n", now - host_us[n - 1]);
    }
    // adds every interval to t_us[stage]; returns the time between the first and the last mark
    double collect(double *t_us)
    {
        if (!on || n < 2) return 0.0;
        BMSP_HIP(hipEventSynchronize(ev[n - 1]));
        for (int i = 1; i < n; i++) {
            float ms = 0.f;
            BMSP_HIP(hipEventElapsedTime(&ms, ev[i - 1], ev[i]));
            if (stage_of[i] >= 0) t_us[stage_of[i]] += (double)ms * 1000.0;
        }
        float tot = 0.f;
        BMSP_HIP(hipEventElapsedTime(&tot, ev[0], ev[n - 1]));
        return (double)tot * 1000.0;
    }
};

// Code objects are loaded lazily, on the first launch of a kernel of their translation unit (0.1 - 0.3 ms each: the first SpMV prepare
// of a process took 2.1 ms, the second 0.16 ms).  Every .hip file with kernels defines an empty one; load_kernels() launches them all
// once per device from the constructors of a matrix (the reference's cudaFree(0) warm-up at the top of main, src/bmSparse_SPMV.cu:237,
// serves the same purpose): what the reference's "bmSparse execution" brackets then starts with the code resident.
#define BMSP_DEFINE_WARM(name)                                                                                        \
    namespace bmsp {                                                                                                  \
    __global__ void warm_kernel_##name() {}                                                                           \
    void warm_##name(hipStream_t st) { hipLaunchKernelGGL(warm_kernel_##name, dim3(1), dim3(1), 0, st); }             \
    }
void load_kernels();
// Host <-> device copies of caller-owned (pageable) memory, staged through the library's own pinned buffers.  A plain hipMemcpy makes the
// runtime register the caller's pages with the driver; when the caller frees them (the MatrixMarket parser's triples: 100 MB), the driver's
// MMU notifier evicts the process's GPU queues and restores them tens of milliseconds later -- the first product of the drop-in
// executable waited 25 ms for that (measured; a 50 ms sleep before the product made it 3 ms).
void copy_h2d_staged(void *dst, const void *src, size_t bytes);
void copy_d2h_staged(void *dst, const void *src, size_t bytes);

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace bmsp
#endif
