// cpp_api_check.cpp -- the reference's C++ surface (include/bmSpMatrix.h, include/CSRMatrix.h) used the way its mains and its
// commented-out checks use it: path constructor, array constructor (adopts device vectors), generate_coo / compare / print,
// bmSparse_SpMV, bmSparse_SpMM, bmSparse_mult, CSRMatrix::multiply.  Built by tests/test_abi.py (compile + link, no GPU needed)
// and run by tests/test_gpu_parity.py on the data/real fixture; prints CHECK lines the test compares.
#include "bmSpMatrix.h"
#include "CSRMatrix.h"
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s A.mtx B.mtx\n", argv[0]); return 2; }
    try {
        const std::string a_path = argv[1], b_path = argv[2];
        bmSpMatrix<float> A(a_path, false), Bt(b_path, true);
        std::printf("CHECK A %d %d %d %d\n", A.num_rows, A.num_cols, A.nnz, A.block_num);

        // array constructor: copy A's four arrays into fresh device vectors and hand them over (src/bmSpMatrix.cu:30-43)
        bmsp::device_vector<uint64_t> k(A.keys.to_host()), b(A.bmps.to_host());
        std::vector<uint64_t> ho = A.offsets.to_host();
        ho.resize((size_t)A.block_num);  // the reference's builder holds block_num offsets
        bmsp::device_vector<uint64_t> o(ho);
        bmsp::device_vector<float> v(A.values.to_host());
        bmSpMatrix<float> A2(A.num_rows, A.num_cols, A.block_num, k, b, o, v);
        std::printf("CHECK adopted %d %d %zu %zu\n", A2.nnz, A2.block_num, k.size(), v.size());

        // generate_coo + compare against itself: mean relative error 0
        const bmsp::coo_matrix<double> &c = A2.host_coo();
        bmsp::coo_matrix<float> cf;
        cf.num_rows = c.num_rows; cf.num_cols = c.num_cols; cf.num_entries = c.num_entries;
        cf.row_indices = c.row_indices; cf.column_indices = c.column_indices;
        cf.values.assign(c.values.begin(), c.values.end());
        std::printf("CHECK compare ");
        A.compare(cf);
        std::printf("\n");

        // SpMV with v = 1 (src/bmSparse_SPMV.cu:279-285) and the k-vector form on 3 copies of it
        bmsp::device_vector<float> x(std::vector<float>((size_t)A.num_cols, 1.0f)), u((size_t)A.num_rows);
        bmSparse_SpMV(A, x.data(), u.data(), false);
        double su = 0;
        for (float f : u.to_host()) su += f;
        std::printf("CHECK spmv %g\n", su);
        const int kv = 3;
        bmsp::device_vector<float> X(std::vector<float>((size_t)A.num_cols * kv, 1.0f)), U((size_t)A.num_rows * kv);
        bmSparse_SpMM(A, X.data(), U.data(), kv);
        double sU = 0;
        for (float f : U.to_host()) sU += f;
        std::printf("CHECK spmm %g\n", sU);

        // product, V15 numerics (tc_version 5), then the same through CSRMatrix
        bmSpMatrix<float> C;
        bmsp_spgemm_stats st;
        bmSparse_mult(A, Bt, C, false, false, 5, &st);
        double sc = 0;
        for (float f : C.values.to_host()) sc += f;
        std::printf("CHECK mult %d %d %g %lld\n", C.block_num, C.nnz, sc, (long long)st.surviving_tasks);
        CSRMatrix ca(a_path), cb(b_path);
        CSRMatrix cc = ca.multiply(cb);
        const bmsp_host_csr<int, float> hc = cc.host();
        double scsr = 0;
        for (float f : hc.values) scsr += f;
        std::printf("CHECK csr %zu %zu %g\n", hc.num_rows, hc.num_entries, scsr);
        // the reference's own path for this class: cusp::multiply on the host container
        const bmsp_host_csr<int, float> hh = ca.multiply_host(cb, 2).host();
        double shost = 0;
        for (float f : hh.values) shost += f;
        double sy = 0;
        for (float f : ca.multiply_host(std::vector<float>(hh.num_cols, 1.0f))) sy += f;
        std::printf("CHECK csr_host %zu %zu %g %g\n", hh.num_rows, hh.num_entries, shost, sy);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
