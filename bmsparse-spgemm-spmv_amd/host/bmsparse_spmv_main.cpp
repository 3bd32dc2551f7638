// bmsparse_spmv_float -- drop-in for the reference executable built by `make spmv` (Makefile:63-64; main at
// src/bmSparse_SPMV.cu:232-312).  Same argv (folder, matrix, matrix, batched) and the same stdout labels in the same
// order; errors go to stderr with a non-zero exit instead of silently continuing.
#include "bmSpMatrix.h"
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

typedef float OUTPUT_TYPE;  // src/bmSparse_SPMV.cu:45

static long long us_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
}

int main(int argc, char **argv)
{
    if (argc < 3) {  // the reference tests argc < 2 and then reads argv[2]
        std::cout << "./main MatrixFolder A_Matrix" << std::endl;
        return 1;
    }
    // The script passes `batched` as the 4th argument (spmv_run_batch.sh:13); the reference reads the 3rd by mistake
    // (:244-245).  Honour the 4th.
    bool batched = argc > 4 && argv[4][0] == '1';
    std::string A_path = std::string(argv[1]) + "/" + std::string(argv[2]);
    std::cout << "A matrix: " << A_path << std::endl;
    try {
        bmsp::check(bmsp_set_device(0));
        bmsp::check(bmsp_synchronize());
        auto t0 = std::chrono::steady_clock::now();
        bmSpMatrix<OUTPUT_TYPE> A_matrix(A_path, false);  // accepts the name with or without ".mtx"
        std::cout << "Parsing mtx files / Loading matrices from disk BMSP: " << us_since(t0) << " \xce\xbcs" << std::endl;
        std::cout << "Running SpMV \n";

        std::vector<OUTPUT_TYPE> vcpu((size_t)A_matrix.num_cols, 1.0f);  // :279-281
        bmsp::device_vector<OUTPUT_TYPE> u((size_t)A_matrix.num_rows);
        t0 = std::chrono::steady_clock::now();
        bmsp::device_vector<OUTPUT_TYPE> v(vcpu);
        std::cout << "Parsing mtx files / Loading matrix and vectors: " << us_since(t0) << " \xce\xbcs" << std::endl;

        bmsp::check(bmsp_synchronize());
        t0 = std::chrono::steady_clock::now();
        bmSparse_SpMV(A_matrix, v.data(), u.data(), batched);
        bmsp::check(bmsp_synchronize());
        std::cout << "bmSparse SpMV execution: " << us_since(t0) << " \xce\xbcs" << std::endl;

        std::vector<OUTPUT_TYPE> u_bmsp = u.to_host();  // :308-309
        if (getenv("BMSP_PRINT_CHECKSUM")) {
            double s = 0;
            for (float x : u_bmsp) s += x;
            std::cout << "u checksum: " << s << std::endl;
        }
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
