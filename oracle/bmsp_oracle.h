/*
 * bmsp_oracle.h -- CPU restatement of the bmSparse hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (libbmsp.so and everything under
 * bmsparse-spgemm-spmv_amd/) never links, imports or calls anything in oracle/.
 *
 * Pinning status: the reference's own sources cannot be compiled in this image
 * (src/ *.cu need nvcc + CUDA Thrust + mma.h; CUSP 0.6.0 needs Thrust-1.x internals),
 * so the oracle is pinned by
 *   (1) the known answers SURVEY.md 8(c) / BASELINE.md 2 derive for data/real/{A,B}_matrix.mtx
 *       (9 bitmaps, y = A*1, 27/27 tasks, 9 C blocks, 255 nnz, 446 products, sums 1070 / 1002),
 *   (2) an independent dict-of-keys / dense product (numpy/scipy) on the reference's .mtx fixtures,
 *   (3) include/half.hpp compiled where it lies (oracle/_ref) for the fp16 rounding rules.
 * The order of fp32 additions above BORDER tasks is "order parity unpinned" in the reference
 * itself (bb_segsort is unstable); the oracle fixes k-ascending order.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef BMSP_ORACLE_H_
#define BMSP_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_F32 = 0, ORC_F16 = 1, ORC_F64 = 2 };

/* bmSparse matrix on the host.  include/bmSpMatrix.h:20-40.
 * values are held as double for every dtype; for ORC_F16 / ORC_F32 every entry is exactly
 * representable in that type (it was cast when the matrix was built, bmSpMatrix.cu:141). */
typedef struct {
    int num_rows, num_cols;
    int64_t nnz, block_num;
    int dtype;
    int transposed;
    uint64_t *keys, *bmps, *offsets; /* block_num each (offsets may carry block_num+1 for a product) */
    double *values;                  /* nnz */
} orc_bmsp;

typedef struct {
    int num_rows, num_cols;
    int64_t nnz;
    int *rows, *cols;
    double *vals;
} orc_coo;

typedef struct {
    int num_rows, num_cols;
    int64_t nnz;
    int *row_offsets; /* num_rows+1 */
    int *cols;
    float *vals;
} orc_csr;

/* stage counters of one SpGEMM (what the reference prints when VERBOSE) */
typedef struct {
    int64_t task_list_size;   /* "Task list size"  SPGEMM.cu:897 */
    int64_t bmp_reduction;    /* "Bmp reduction"   SPGEMM.cu:953 */
    int64_t surviving_tasks;
    int64_t c_blocks;         /* "C blocks"        SPGEMM.cu:1284 */
    int64_t c_nnz;            /* "C nnz"           SPGEMM.cu:1285 */
    int64_t scalar_products;  /* number of a_ik*b_kj products with both operands stored */
} orc_spgemm_stats;

/* fp16 helpers (round-to-nearest-even, as half.hpp:373-374 and CUDA __double2half) */
uint16_t orc_f64_to_f16_bits(double x);
double   orc_f16_bits_to_f64(uint16_t h);
double   orc_round_to_dtype(double x, int dtype);

/* MatrixMarket -> COO exactly as the reference's constructor parses it (bmSpMatrix.cu:111-161).
 * strict = 0: reference behaviour ("symmetric" substring, row col value triples);
 * strict = 1: CUSP reader behaviour for CSRMatrix(std::string) (matrix_market.inl:71-97,171-196,245-295):
 *             pattern -> 1, symmetric mirrored, skew/hermitian rejected, sorted by (row,col).
 * returns 0 on success. */
int  orc_mtx_read(const char *path, int strict, orc_coo *out);
void orc_coo_free(orc_coo *m);

/* COO -> bmSparse (bmSpMatrix.cu:163-216). */
int  orc_bmsp_from_coo(const orc_coo *coo, int dtype, int transposed, orc_bmsp *out);
void orc_bmsp_free(orc_bmsp *m);

/* bmSparse -> COO sorted by (row,col) (bmSpMatrix.cu:320-363); honours m->transposed. */
int  orc_bmsp_to_coo(const orc_bmsp *m, orc_coo *out);

/* mean relative error as bmSpMatrix::compare prints it (bmSpMatrix.cu:381-432). */
double orc_bmsp_compare(const orc_bmsp *m, const orc_coo *other);

/* u = A*v as spmv_kernel computes it (SPMV.cu:72-82,153-189), fp32. */
int  orc_spmv_f32(const orc_bmsp *A, const float *v, float *u);

/* C = A*B as bmSparse_mult computes it (SPGEMM.cu:827-1223); B must be built transposed.
 * exact_products = 0: V15 semantics (product rounded to the input type, SPGEMM.cu:269-273)
 * exact_products = 1: tensor-core semantics (exact products, fp32 accumulate, SPGEMM.cu:294-417) */
int  orc_spgemm(const orc_bmsp *A, const orc_bmsp *B, int exact_products, orc_bmsp *C,
                orc_spgemm_stats *stats);

/* intermediate stages exposed for stage-by-stage parity (Appendix A of SURVEY.md) */
uint64_t orc_bmp_product(uint64_t bmpA, uint64_t bmpB_transposed);  /* SPGEMM.cu:787-810 */
int      orc_bmp_product_empty(uint64_t bmpA, uint64_t bmpB_transposed); /* SPGEMM.cu:742-757 */

/* gold for the segmented sort (bb_segsort-master/main.cu:121-143): stable sort of each segment. */
int  orc_segsort_u64_kv(uint64_t *keys, uint64_t *vals2 /* n pairs of 2 u64 */, int64_t n,
                        const int64_t *segs, int64_t nseg);

/* CPU baseline = restatement of cusp::multiply on host CSR. */
int  orc_csr_from_coo(const orc_coo *coo, orc_csr *out); /* coo must be sorted by (row,col) */
void orc_csr_free(orc_csr *m);
/* cusp/system/detail/sequential/multiply/csr_spmv.h:56-73 (threads=1) and
 * cusp/system/omp/detail/multiply/csr_spmv.h:67-85 (threads>1) */
int  orc_csr_spmv(const orc_csr *A, const float *x, float *y, int threads);
/* cusp/system/detail/sequential/multiply/csr_spgemm.h:39-157,165-198 and the omp variant
 * cusp/system/omp/detail/multiply/csr_spgemm.h:40-87,93-.. ; returns scalar products in *products */
int  orc_csr_spgemm(const orc_csr *A, const orc_csr *B, orc_csr *C, int threads, int64_t *products);
int  orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
