// rocsparse_ref.cpp -- vendor comparison column: rocSPARSE CSR SpMV and CSR SpGEMM timed on the same
// matrices as the bmSparse operators.  Reporting only: nothing in libbmsp.so links or calls this; bench.py loads
// vendor_compare/librocsparse_ref.so when it exists.  (The reference carries dead cuSPARSE wrappers for the same purpose,
// src/cuSparse_SPGEMM.cu / cuSparse_SPMV.cu.)
#include <hip/hip_runtime.h>
#include <rocsparse/rocsparse.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define RSCK(x) do { rocsparse_status s_ = (x); if (s_ != rocsparse_status_success) { std::fprintf(stderr, "%s: status %d\n", #x, (int)s_); return 2; } } while (0)

#pragma clang diagnostic ignored "-Wdeprecated-declarations"

namespace {
template <typename T>
int upload(T **d, const T *h, size_t n)
{
    HIPCK(hipMalloc((void **)d, (n ? n : 1) * sizeof(T)));
    if (n) HIPCK(hipMemcpy(*d, h, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}
}  // namespace

// y = A x, fp32 CSR with int32 indices.  alg: 0 default, 1 adaptive (preprocessed), 2 rowsplit ("stream"), 3 lrb
extern "C" int vendor_csr_spmv(int m, int n, int64_t nnz, const int *ptr, const int *col, const float *val, const float *x, float *y,
                               int alg, int iters, double *ms_per_spmv, double *preprocess_ms)
{
    rocsparse_handle h;
    RSCK(rocsparse_create_handle(&h));
    int *dptr, *dcol;
    float *dval, *dx, *dy;
    if (upload(&dptr, ptr, (size_t)m + 1) || upload(&dcol, col, (size_t)nnz) || upload(&dval, val, (size_t)nnz) || upload(&dx, x, (size_t)n)) return 1;
    HIPCK(hipMalloc((void **)&dy, (size_t)(m ? m : 1) * 4));
    rocsparse_spmat_descr A;
    rocsparse_dnvec_descr vx, vy;
    RSCK(rocsparse_create_csr_descr(&A, m, n, nnz, dptr, dcol, dval, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero,
                                    rocsparse_datatype_f32_r));
    RSCK(rocsparse_create_dnvec_descr(&vx, n, dx, rocsparse_datatype_f32_r));
    RSCK(rocsparse_create_dnvec_descr(&vy, m, dy, rocsparse_datatype_f32_r));
    const float alpha = 1.f, beta = 0.f;
    const rocsparse_spmv_alg a = alg == 1 ? rocsparse_spmv_alg_csr_adaptive : alg == 2 ? rocsparse_spmv_alg_csr_rowsplit : alg == 3 ? rocsparse_spmv_alg_csr_lrb
                                                                                                                                       : rocsparse_spmv_alg_default;
    size_t bytes = 0;
    RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_buffer_size, &bytes, nullptr));
    void *buf = nullptr;
    HIPCK(hipMalloc(&buf, bytes ? bytes : 4));
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    HIPCK(hipEventRecord(e0, 0));
    RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_preprocess, &bytes, buf));
    HIPCK(hipEventRecord(e1, 0)); HIPCK(hipEventSynchronize(e1));
    float pms = 0; HIPCK(hipEventElapsedTime(&pms, e0, e1));
    *preprocess_ms = pms;
    for (int i = 0; i < 5; i++)
        RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_compute, &bytes, buf));
    HIPCK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++)
        RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A, vx, &beta, vy, rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_compute, &bytes, buf));
    HIPCK(hipEventRecord(e1, 0)); HIPCK(hipEventSynchronize(e1));
    float ms = 0; HIPCK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_spmv = ms / (iters > 0 ? iters : 1);
    if (m) HIPCK(hipMemcpy(y, dy, (size_t)m * 4, hipMemcpyDeviceToHost));
    rocsparse_destroy_spmat_descr(A); rocsparse_destroy_dnvec_descr(vx); rocsparse_destroy_dnvec_descr(vy);
    hipFree(buf); hipFree(dptr); hipFree(dcol); hipFree(dval); hipFree(dx); hipFree(dy);
    hipEventDestroy(e0); hipEventDestroy(e1);
    rocsparse_destroy_handle(h);
    return 0;
}

// the same product ROTATED over `copies` device-resident copies of the matrix (each with its own descriptor, preprocessed), so that a
// sweep streams from HBM instead of the Infinity Cache: the protocol bench.py's headline `value` uses for the bmSparse sweep
extern "C" int vendor_csr_spmv_rotated(int m, int n, int64_t nnz, const int *ptr, const int *col, const float *val, const float *x, int alg, int iters,
                                       int copies, double *ms_per_spmv)
{
    rocsparse_handle h;
    RSCK(rocsparse_create_handle(&h));
    if (copies < 1) copies = 1;
    std::vector<int *> dptr((size_t)copies), dcol((size_t)copies);
    std::vector<float *> dval((size_t)copies), dy((size_t)copies);
    std::vector<rocsparse_spmat_descr> A((size_t)copies);
    std::vector<rocsparse_dnvec_descr> vy((size_t)copies);
    std::vector<void *> buf((size_t)copies, nullptr);
    float *dx;
    if (upload(&dx, x, (size_t)n)) return 1;
    rocsparse_dnvec_descr vx;
    RSCK(rocsparse_create_dnvec_descr(&vx, n, dx, rocsparse_datatype_f32_r));
    const float alpha = 1.f, beta = 0.f;
    const rocsparse_spmv_alg a = alg == 1 ? rocsparse_spmv_alg_csr_adaptive : alg == 2 ? rocsparse_spmv_alg_csr_rowsplit : alg == 3 ? rocsparse_spmv_alg_csr_lrb
                                                                                                                                       : rocsparse_spmv_alg_default;
    size_t bytes = 0;
    for (int c = 0; c < copies; c++) {
        if (upload(&dptr[(size_t)c], ptr, (size_t)m + 1) || upload(&dcol[(size_t)c], col, (size_t)nnz) || upload(&dval[(size_t)c], val, (size_t)nnz)) return 1;
        HIPCK(hipMalloc((void **)&dy[(size_t)c], (size_t)(m ? m : 1) * 4));
        RSCK(rocsparse_create_csr_descr(&A[(size_t)c], m, n, nnz, dptr[(size_t)c], dcol[(size_t)c], dval[(size_t)c], rocsparse_indextype_i32, rocsparse_indextype_i32,
                                        rocsparse_index_base_zero, rocsparse_datatype_f32_r));
        RSCK(rocsparse_create_dnvec_descr(&vy[(size_t)c], m, dy[(size_t)c], rocsparse_datatype_f32_r));
        RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A[(size_t)c], vx, &beta, vy[(size_t)c], rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_buffer_size, &bytes, nullptr));
        HIPCK(hipMalloc(&buf[(size_t)c], bytes ? bytes : 4));
        RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A[(size_t)c], vx, &beta, vy[(size_t)c], rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_preprocess, &bytes, buf[(size_t)c]));
        RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A[(size_t)c], vx, &beta, vy[(size_t)c], rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_compute, &bytes, buf[(size_t)c]));
    }
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    HIPCK(hipDeviceSynchronize());
    HIPCK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) {
        const size_t c = (size_t)(i % copies);
        RSCK(rocsparse_spmv(h, rocsparse_operation_none, &alpha, A[c], vx, &beta, vy[c], rocsparse_datatype_f32_r, a, rocsparse_spmv_stage_compute, &bytes, buf[c]));
    }
    HIPCK(hipEventRecord(e1, 0)); HIPCK(hipEventSynchronize(e1));
    float ms = 0; HIPCK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_spmv = ms / (iters > 0 ? iters : 1);
    for (int c = 0; c < copies; c++) {
        rocsparse_destroy_spmat_descr(A[(size_t)c]); rocsparse_destroy_dnvec_descr(vy[(size_t)c]);
        hipFree(buf[(size_t)c]); hipFree(dptr[(size_t)c]); hipFree(dcol[(size_t)c]); hipFree(dval[(size_t)c]); hipFree(dy[(size_t)c]);
    }
    rocsparse_destroy_dnvec_descr(vx);
    hipFree(dx);
    hipEventDestroy(e0); hipEventDestroy(e1);
    rocsparse_destroy_handle(h);
    return 0;
}

// C = A B, fp32 CSR.  Every timed iteration runs the nnz stage and the compute stage (the full product with the C arrays and the
// work buffer kept from the first call, i.e. what a pooled allocator gives); *first_ms includes the allocations.
extern "C" int vendor_csr_spgemm(int m, int k, int n, int64_t nnzA, const int *ptrA, const int *colA, const float *valA, int64_t nnzB, const int *ptrB,
                                 const int *colB, const float *valB, int iters, double *ms_per_product, double *first_ms, int64_t *c_nnz, double *c_sum)
{
    rocsparse_handle h;
    RSCK(rocsparse_create_handle(&h));
    int *dpa, *dca, *dpb, *dcb, *dpc, *dcc = nullptr, *dpd;
    float *dva, *dvb, *dvc = nullptr;
    if (upload(&dpa, ptrA, (size_t)m + 1) || upload(&dca, colA, (size_t)nnzA) || upload(&dva, valA, (size_t)nnzA)) return 1;
    if (upload(&dpb, ptrB, (size_t)k + 1) || upload(&dcb, colB, (size_t)nnzB) || upload(&dvb, valB, (size_t)nnzB)) return 1;
    HIPCK(hipMalloc((void **)&dpc, ((size_t)m + 1) * 4));
    HIPCK(hipMalloc((void **)&dpd, ((size_t)m + 1) * 4));
    HIPCK(hipMemset(dpd, 0, ((size_t)m + 1) * 4));
    rocsparse_spmat_descr A, B, C, D;
    RSCK(rocsparse_create_csr_descr(&A, m, k, nnzA, dpa, dca, dva, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    RSCK(rocsparse_create_csr_descr(&B, k, n, nnzB, dpb, dcb, dvb, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    RSCK(rocsparse_create_csr_descr(&D, m, n, 0, dpd, nullptr, nullptr, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    RSCK(rocsparse_create_csr_descr(&C, m, n, 0, dpc, nullptr, nullptr, rocsparse_indextype_i32, rocsparse_indextype_i32, rocsparse_index_base_zero, rocsparse_datatype_f32_r));
    const float alpha = 1.f, beta = 0.f;
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0)); HIPCK(hipEventCreate(&e1));
    size_t bytes = 0;
    void *buf = nullptr;
    int64_t rows_c = 0, cols_c = 0, nz = 0;
    HIPCK(hipEventRecord(e0, 0));
    RSCK(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, D, C, rocsparse_datatype_f32_r, rocsparse_spgemm_alg_default,
                          rocsparse_spgemm_stage_buffer_size, &bytes, nullptr));
    HIPCK(hipMalloc(&buf, bytes ? bytes : 4));
    RSCK(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, D, C, rocsparse_datatype_f32_r, rocsparse_spgemm_alg_default,
                          rocsparse_spgemm_stage_nnz, &bytes, buf));
    RSCK(rocsparse_spmat_get_size(C, &rows_c, &cols_c, &nz));
    HIPCK(hipMalloc((void **)&dcc, (size_t)(nz ? nz : 1) * 4));
    HIPCK(hipMalloc((void **)&dvc, (size_t)(nz ? nz : 1) * 4));
    RSCK(rocsparse_csr_set_pointers(C, dpc, dcc, dvc));
    RSCK(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, D, C, rocsparse_datatype_f32_r, rocsparse_spgemm_alg_default,
                          rocsparse_spgemm_stage_compute, &bytes, buf));
    HIPCK(hipEventRecord(e1, 0)); HIPCK(hipEventSynchronize(e1));
    float fms = 0; HIPCK(hipEventElapsedTime(&fms, e0, e1));
    *first_ms = fms;
    HIPCK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) {
        RSCK(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, D, C, rocsparse_datatype_f32_r, rocsparse_spgemm_alg_default,
                              rocsparse_spgemm_stage_nnz, &bytes, buf));
        RSCK(rocsparse_csr_set_pointers(C, dpc, dcc, dvc));
        RSCK(rocsparse_spgemm(h, rocsparse_operation_none, rocsparse_operation_none, &alpha, A, B, &beta, D, C, rocsparse_datatype_f32_r, rocsparse_spgemm_alg_default,
                              rocsparse_spgemm_stage_compute, &bytes, buf));
    }
    HIPCK(hipEventRecord(e1, 0)); HIPCK(hipEventSynchronize(e1));
    float ms = 0; HIPCK(hipEventElapsedTime(&ms, e0, e1));
    *ms_per_product = ms / (iters > 0 ? iters : 1);
    *c_nnz = nz;
    std::vector<float> hv((size_t)nz);
    if (nz) HIPCK(hipMemcpy(hv.data(), dvc, (size_t)nz * 4, hipMemcpyDeviceToHost));
    double s = 0;
    for (float v : hv) s += v;
    *c_sum = s;
    rocsparse_destroy_spmat_descr(A); rocsparse_destroy_spmat_descr(B); rocsparse_destroy_spmat_descr(C); rocsparse_destroy_spmat_descr(D);
    hipFree(buf); hipFree(dpa); hipFree(dca); hipFree(dva); hipFree(dpb); hipFree(dcb); hipFree(dvb); hipFree(dpc); hipFree(dcc); hipFree(dvc); hipFree(dpd);
    hipEventDestroy(e0); hipEventDestroy(e1);
    rocsparse_destroy_handle(h);
    return 0;
}
