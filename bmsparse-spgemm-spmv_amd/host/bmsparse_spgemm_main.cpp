// bmsparse_spgemm_float -- drop-in for the reference executable built by `make spgemm` (Makefile:60-61; main at
// src/bmSparse_SPGEMM.cu:1226-1288).  argv = folder, A, B, segmented, tc_version, verbose; inputs are read as fp16,
// C is fp32 (:1261-1262,1272); stdout labels and order as in SURVEY.md Appendix B.
#include "bmSpMatrix.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

typedef float OUTPUT_TYPE;  // src/bmSparse_SPGEMM.cu:51

int main(int argc, char **argv)
{
    long segmented = 0, tc_version = 5;
    bool VERBOSE = false;
    if (argc < 4) {  // the reference tests argc < 3 and then reads argv[3]
        std::cout << "./main MatrixFolder A_Matrix B_Matrix" << std::endl;
        return 1;
    }
    if (argc > 4) segmented = strtol(argv[4], NULL, 10);
    if (argc > 5) tc_version = strtol(argv[5], NULL, 10);
    if (argc > 6) VERBOSE = argv[6][0] == '1';

    std::string A_path = std::string(argv[1]) + "/" + std::string(argv[2]);
    std::string B_path = std::string(argv[1]) + "/" + std::string(argv[3]);
    std::cout << "A matrix: " << A_path << std::endl;
    std::cout << "B matrix: " << B_path << std::endl;
    try {
        bmsp::check(bmsp_set_device(0));
        auto t0 = std::chrono::steady_clock::now();
        bmSpMatrix<half> A_bmSp(A_path + ".mtx", false);
        bmSpMatrix<half> B_bmSp(B_path + ".mtx", true);
        auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "Parsing mtx files / Loading matrices from disk BMSP: " << us << " \xce\xbcs" << std::endl;
        bmsp::check(bmsp_synchronize());

        bmSpMatrix<OUTPUT_TYPE> C;
        t0 = std::chrono::steady_clock::now();
        bmSparse_mult(A_bmSp, B_bmSp, C, segmented != 0, VERBOSE, tc_version);
        us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "bmSparse execution: " << us << " \xce\xbcs" << std::endl;
        std::cout << "C blocks: " << C.keys.size() << std::endl;
        std::cout << "C nnz: " << C.nnz << std::endl;
        if (getenv("BMSP_PRINT_CHECKSUM")) {
            double s = 0;
            for (float x : C.values.to_host()) s += x;
            std::cout << "C checksum: " << s << std::endl;
        }
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
