/*
 * bmsp.h -- C ABI of the MI355X-native bmSparse engine (libbmsp.so).
 *
 * This is the drop-in boundary.  The reference (GonzaBerger/bmSparse-SPGEMM-SPMV) has no FFI: its
 * operators are C++ templates compiled into the two executables.  Every entry point below names the
 * reference interface it replaces (paths relative to the reference repository root).  The C++ headers
 * include/bmSpMatrix.h and include/CSRMatrix.h keep the reference's class / function names on top of
 * this ABI, so a maintainer swaps the reference's nvcc translation units for `-lbmsp`.
 *
 * Conventions
 *   - plain pointers and sizes only; `void *stream` is a hipStream_t (NULL = the null stream);
 *   - "device pointer" means memory visible to the current HIP device (bmsp_malloc, hipMalloc, or a
 *     PyTorch-ROCm tensor's data_ptr());
 *   - every call returns BMSP_OK (0) or a negative bmsp_status; bmsp_last_error() gives the text for
 *     the calling thread.  Nothing calls exit() (the reference exits on CUDA errors,
 *     src/bmSparse_SPMV.cu:62-70);
 *   - a missing / unreadable file is an error (the reference silently builds an empty matrix,
 *     src/bmSpMatrix.cu:114-127);
 *   - not thread-safe per matrix handle; distinct handles may be used from distinct threads.
 */
#ifndef BMSP_H_
#define BMSP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BMSP_BLOCK_WIDTH 8   /* include/bmSpMatrix.h:15 */
#define BMSP_BLOCK_HEIGHT 8  /* include/bmSpMatrix.h:16 */

typedef enum {
    BMSP_OK = 0,
    BMSP_ERR_INVALID = -1,   /* bad argument / shape mismatch */
    BMSP_ERR_IO = -2,        /* file missing or malformed */
    BMSP_ERR_HIP = -3,       /* a HIP runtime call failed */
    BMSP_ERR_NOMEM = -4,
    BMSP_ERR_UNSUPPORTED = -5,
    BMSP_ERR_LIMIT = -6      /* a 32-bit internal limit would overflow */
} bmsp_status;

typedef enum { BMSP_F32 = 0, BMSP_F16 = 1, BMSP_F64 = 2 } bmsp_dtype;

typedef struct bmsp_matrix_s *bmsp_matrix_t; /* replaces class bmSpMatrix<T>, include/bmSpMatrix.h:20-40 */
typedef struct bmsp_csr_s *bmsp_csr_t;       /* replaces class CSRMatrix,     include/CSRMatrix.h:13-21 */

const char *bmsp_last_error(void);
const char *bmsp_version(void);

/* ---- device plumbing (lets a plain-C/C++ host manage v and u the way the reference's main does
 *      with cudaMalloc/cudaMemcpy, src/bmSparse_SPMV.cu:276-285,309) ---- */
int bmsp_device_count(int *count);
int bmsp_set_device(int device);
int bmsp_malloc(void **dptr, size_t bytes);
int bmsp_free(void *dptr);
int bmsp_memcpy_h2d(void *dst, const void *src, size_t bytes);
int bmsp_memcpy_d2h(void *dst, const void *src, size_t bytes);
int bmsp_memcpy_d2d(void *dst, const void *src, size_t bytes);
int bmsp_memset(void *dptr, int value, size_t bytes);
int bmsp_synchronize(void);
int bmsp_trim_pool(void); /* return cached device memory to the driver */
/* device-side timing (hipEvent) on the stream the operators are launched on */
int bmsp_event_create(void **event);
int bmsp_event_record(void *event, void *stream);
int bmsp_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */
int bmsp_event_destroy(void *event);

/* ---- container / builder ------------------------------------------------------------------- */

/* bmSpMatrix<T>::bmSpMatrix(std::string path, bool transpose)  -- src/bmSpMatrix.cu:111-219.
 * Parses a MatrixMarket coordinate file (real / integer / pattern; general / symmetric) on the host and
 * builds keys/bmps/offsets/values on the device.  `transposed` lays tiles out column-major (the B operand
 * of bmsp_spgemm, src/bmSparse_SPGEMM.cu:1262).  Accepts `path` with or without the ".mtx" suffix
 * (the reference's two mains disagree, src/bmSparse_SPMV.cu:257,270). */
int bmsp_matrix_from_mtx(const char *path, int transposed, bmsp_dtype dtype, bmsp_matrix_t *out);

/* Same builder from host COO triples (0-based).  Values are double, cast to dtype with round-to-nearest-even
 * exactly like the reference's `(valueType)data` (src/bmSpMatrix.cu:141).  Duplicate coordinates are summed
 * (the reference corrupts its popcount addressing on duplicates, src/bmSpMatrix.cu:183-216). */
int bmsp_matrix_from_coo(int num_rows, int num_cols, int64_t nnz, const int *rows, const int *cols,
                         const double *vals, int transposed, bmsp_dtype dtype, bmsp_matrix_t *out);

/* Same builder from COO triples already resident on the device (rows/cols int32, vals float64). */
int bmsp_matrix_from_coo_device(int num_rows, int num_cols, int64_t nnz, const int *d_rows, const int *d_cols,
                                const double *d_vals, int transposed, bmsp_dtype dtype, void *stream,
                                bmsp_matrix_t *out);

/* bmSpMatrix<T>::bmSpMatrix(int,int,int, keys&, bmps&, offsets&, values&)  -- src/bmSpMatrix.cu:30-43.
 * ownership: 0 = copy the four device arrays; 1 = adopt them (they must come from bmsp_malloc; this is the
 * reference's swap semantics); 2 = borrow (caller keeps them alive and frees them). */
int bmsp_matrix_from_arrays(int num_rows, int num_cols, int64_t block_num, int64_t nnz, uint64_t *d_keys,
                            uint64_t *d_bmps, uint64_t *d_offsets, void *d_values, bmsp_dtype dtype,
                            int transposed, int ownership, bmsp_matrix_t *out);

/* Binary cache of the built matrix (the reference only hints at "Dumping bmSparse matrices to disk",
 * src/bmSparse_SPGEMM.cu:21,27): header + keys + bmps + offsets + values, so a SuiteSparse file is parsed and
 * sorted once.  bmsp_matrix_load restores exactly what bmsp_matrix_save wrote (bit-identical arrays). */
int bmsp_matrix_save(bmsp_matrix_t m, const char *path);
int bmsp_matrix_load(const char *path, bmsp_matrix_t *out);

int bmsp_matrix_free(bmsp_matrix_t m);

/* public members of bmSpMatrix<T>: num_rows, num_cols, nnz, block_num (include/bmSpMatrix.h:32) */
int bmsp_matrix_info(bmsp_matrix_t m, int *num_rows, int *num_cols, int64_t *nnz, int64_t *block_num,
                     bmsp_dtype *dtype, int *transposed);
/* public members keys, bmps, offsets, values (include/bmSpMatrix.h:28-31) as device pointers.
 * offsets carries block_num+1 entries (the last equals nnz); the reference holds block_num from the builder
 * (src/bmSpMatrix.cu:194) and block_num+1 after a product (src/bmSparse_SPGEMM.cu:1087,1179). */
int bmsp_matrix_arrays(bmsp_matrix_t m, uint64_t **d_keys, uint64_t **d_bmps, uint64_t **d_offsets,
                       void **d_values);
/* The arrays above are writable (bmSpMatrix<T>::values.data() is public in the reference, include/bmSpMatrix.h:31, and the reference
 * keeps no derived state, so it always sees current values).  This engine caches derived structures on the handle: after writing
 * VALUES in place call bmsp_matrix_invalidate(m, 0) before the next product (drops the dense tile copies the block-MAC kernels read);
 * after changing keys / bitmaps / offsets call it with structure_changed = 1 (drops the block-row pointer, SpMV plan and position
 * cache, block records too).  Synchronises the device.  SpMV reads values directly: a value-only update needs no call for it. */
int bmsp_matrix_invalidate(bmsp_matrix_t m, int structure_changed);
/* dense block-row pointer (num_block_rows+1 uint32 entries) the operators use; built once and cached
 * (the reference rebuilds a compressed one on every call, src/bmSparse_SPMV.cu:199-206). */
int bmsp_matrix_block_row_ptr(bmsp_matrix_t m, const uint32_t **d_rowptr, int64_t *num_block_rows);

/* Builds, ahead of the first product, the per-matrix structures the operators derive lazily and cache (none of them is part of
 * the reference's public state): `what` bit 0 = the SpMV sweep plan (the reference rebuilds its block-row pointer inside every
 * timed SpMV, src/bmSparse_SPMV.cu:199-206) and, for matrices of sparse tiles, its position cache: one 16-bit {tile slot, position}
 * entry per stored value (2 * nnz bytes of device memory; BMSP_SPMV_NO_POSCACHE=1 or a size above BMSP_SPMV_POSCACHE_MAX bytes,
 * default 4 GiB, keeps the bitmap decode inside the kernel instead); bit 1 = the packed operand records of the SpGEMM block-MAC (both
 * operand roles, incl. the dense fp16 tile copies of the MFMA kernels: 128 bytes per block).
 * Idempotent; asynchronous on `stream` except for the scalar read-backs of the plan. */
#define BMSP_PREPARE_SPMV 1
#define BMSP_PREPARE_SPGEMM 2
int bmsp_matrix_prepare(bmsp_matrix_t m, int what, void *stream);

/* bmSpMatrix<T>::generate_coo()  -- src/bmSpMatrix.cu:320-363.  Expands to host COO sorted by (row,col);
 * rows/cols/vals must hold nnz entries.  Unlike the reference it honours the transposed layout. */
int bmsp_matrix_to_coo_host(bmsp_matrix_t m, int *rows, int *cols, double *vals);

/* SURVEY 8(f)2 -- the same expansion with the result left on the device, as COO or CSR sorted by (row, col), so a product can
 * feed CSR consumers without a host round trip.  d_rows/d_cols/d_vals hold nnz entries, d_row_offsets num_rows+1. */
int bmsp_matrix_to_coo_device(bmsp_matrix_t m, int *d_rows, int *d_cols, double *d_vals, void *stream);
int bmsp_matrix_to_csr_device(bmsp_matrix_t m, int *d_row_offsets, int *d_cols, double *d_vals, void *stream);
/* ... and the builder from a device-resident CSR (int32 offsets/columns, float64 values; duplicates summed as in from_coo). */
int bmsp_matrix_from_csr_device(int num_rows, int num_cols, int64_t nnz, const int *d_row_offsets, const int *d_cols,
                                const double *d_vals, int transposed, bmsp_dtype dtype, void *stream, bmsp_matrix_t *out);

/* bmSpMatrix<T>::compare(coo)  -- src/bmSpMatrix.cu:381-432.  Mean relative error against a host COO
 * comparand (entries of the comparand that are absent from m are skipped).  *missing counts entries of m
 * that the comparand lacks (the reference would walk out of bounds). */
int bmsp_matrix_compare(bmsp_matrix_t m, int64_t nnz, const int *rows, const int *cols, const double *vals,
                        double *mean_rel_err, int64_t *missing);
/* the same comparison with a device-resident comparand, entirely on the device (large products) */
int bmsp_matrix_compare_device(bmsp_matrix_t m, int64_t nnz, const int *d_rows, const int *d_cols, const double *d_vals,
                               double *mean_rel_err, int64_t *missing, void *stream);

/* ---- operators ------------------------------------------------------------------------------ */

/* variants of the SpMV sweep.  0 = reference's default path (`batched`=false, spmv_kernel
 * src/bmSparse_SPMV.cu:153-189), 1 = the wavefront-reduce path `batched`=true was meant to select
 * (spmv_kernel_new :84-150).  Both give the same u; they differ in the lane mapping. */
#define BMSP_SPMV_DEFAULT 0
#define BMSP_SPMV_BATCHED 1

/* bmSparse_SpMV<VI,VO>(A, v, u, batched)  -- src/bmSparse_SPMV.cu:191-230.   u = A * v
 * v: device, num_cols entries of A's dtype; u: device, num_rows entries (float for F32/F16, double for F64).
 * Asynchronous on `stream`; A is not modified.  Rows of empty block-rows are written as 0.
 * One stream per handle at a time: the cached sweep plan holds the hub rows' carry slots and arrival counters, so two sweeps of the
 * SAME handle must not be in flight on different streams (different handles may). */
int bmsp_spmv(bmsp_matrix_t A, const void *d_v, void *d_u, int variant, void *stream);

/* What bmsp_spmv(A, ..., variant, ...) launches and what that launch must move (the measurement contract of SURVEY 8(d): "if the build
 * stores a more compact internal layout, use that layout's compulsory bytes"): kernel_name receives the name of the kernel the launcher
 * picks for this matrix (the cached plan and position cache are built if they are not yet), *compulsory_bytes the bytes of every array
 * that kernel reads or writes, each once, counted from the plan; *format_bytes the layout-independent figure 24 B per block + values +
 * block-row pointer + x + y.  Reporting only: no reference line to replace (the reference prints a time, src/bmSparse_SPMV.cu:306). */
int bmsp_spmv_launch_info(bmsp_matrix_t A, int variant, char *kernel_name, size_t kernel_name_cap, int64_t *compulsory_bytes, int64_t *format_bytes);

/* SURVEY 8(f)3 -- Y = A * X for k vectors at once (what the reference's unfinished `batched` path points at,
 * src/bmSparse_SPMV.cu:84-150,191).  X is row-major num_cols x k with leading dimension ldx (elements of A's dtype),
 * Y row-major num_rows x k with leading dimension ldy (float, double for F64): one pass over A's tiles for all k. */
int bmsp_spmm(bmsp_matrix_t A, const void *d_X, int64_t ldx, void *d_Y, int64_t ldy, int k, void *stream);

/* per-stage figures of one product: the lines the reference prints when VERBOSE
 * (src/bmSparse_SPGEMM.cu:849-1220) plus what the roofline needs. */
typedef struct {
    int64_t task_list_size;  /* candidate block pairs ("Task list size") */
    int64_t bmp_reduction;   /* pairs dropped by the bitmap filter ("Bmp reduction") */
    int64_t surviving_tasks;
    int64_t c_blocks;        /* "C blocks" */
    int64_t c_nnz;           /* "C nnz" (symbolic) */
    double t_us[10];         /* device time per stage, microseconds: [1]=T_1 [2]=T_2 [3]=T_3 [4]=T_4 [5]=T_5
                                [6]=T_6 [7]=T_7 [9]=T_9 ; [0]=whole call ("Toda F"); [8]=segmented sort only */
    int sort_path;           /* 0 = global radix sort (reference: thrust::sort), 1 = segmented sort, 2 = none: C's structure formed
                              * block-row by block-row in LDS (BMSP_SORT_PATH_ROWMERGE; T_3 then holds the whole symbolic pass) */
    int mac_kernel;          /* which block-MAC kernel ran (see tc_version) */
    int mac_variant;         /* which implementation of it: BMSP_MAC_* below */
    int sort_long;           /* detail of sort_path.  sort_path 1: how block-rows of more tasks than one wave sorts in registers were ordered -- 0 =
                              * there were none, 1 = pieces + merge passes, 2 = stable counting passes on the column bits (BMSP_SORT_LONG_*).
                              * sort_path 2 (row-merge): 1 = strip mode (C's structure only, no task list), 2 = task-list mode (BMSP_ROWMERGE_*) */
} bmsp_spgemm_stats;
/* implementations behind one tc_version (the launcher picks by the product's shape; all give the tc_version's numerics) */
#define BMSP_SORT_PATH_ROWMERGE 2
#define BMSP_ROWMERGE_STRIP 1
#define BMSP_ROWMERGE_TASKLIST 2
#define BMSP_SORT_LONG_MERGE 1
#define BMSP_SORT_LONG_RADIX 2
#define BMSP_SORT_PATH_ROWWINDOW 3 /* none either: block-rows of C formed window by window of block columns in dense LDS tables (operands with hub block-rows) */
#define BMSP_MAC_DEFAULT 0 /* the only kernel of that tc_version (V15 vector-ALU kernels, K = 16 MFMA kernels) */
#define BMSP_MAC_STAGED 1  /* tc 4: K = 32 MFMA, operands staged through LDS per task (sparse task lists) */
#define BMSP_MAC_DIRECT 2  /* tc 4: K = 32 MFMA, operand lines loaded per task straight into the MFMA lanes */
#define BMSP_MAC_STRIP 3   /* tc 4: K = 32 MFMA, two block-rows of C per wave, A tiles register-resident, B tiles loaded once per strip */
#define BMSP_MAC_ROWSPARSE 5 /* V15 numerics (fp32 operands; fp16 operands under tc 5) on operands of nearly empty tiles (row-merge strip mode): the
                             * reference's chain over the products of STORED values only, row-wise over CSR copies, accumulators per C value in LDS */
#define BMSP_MAC_F32MFMA 4 /* tc 5, fp32, opt-in (BMSP_MAC_F32MFMA=1): V15's fmaf chain on v_mfma_f32_16x16x4_f32, operands from lane-ordered tile copies */

/* sort modes = the reference's `segmented` argument (src/bmSparse_SPGEMM.cu:963-1016):
 * 0 = global sort below BMSP_SORT_BORDER surviving tasks, segmented sort above; 1 = always segmented;
 * 2 = always global. */
#define BMSP_SORT_AUTO 0
#define BMSP_SORT_SEGMENTED 1
#define BMSP_SORT_GLOBAL 2
#define BMSP_SORT_BORDER 2730000 /* src/bmSparse_SPGEMM.cu:53 */

/* bmSparse_mult<VI,VO>(A, B, C, mode, VERBOSE, tc_version)  -- src/bmSparse_SPGEMM.cu:827-1223.   C = A * B
 * A: normal layout; B: built with transposed=1; same dtype.  *C receives a new fp32 (fp64 for F64 inputs)
 * matrix in normal layout.  tc_version selects the block multiply-accumulate kernel like the reference's
 * switch (:1132-1155): 5 = vector-ALU kernel with the reference's V15 numerics (each product rounded to the
 * input type, fp32 accumulate); 1..4 = matrix-core (MFMA) kernel: exact products, fp32 accumulate.
 * verbose != 0 prints the reference's stage lines to stdout.  stats may be NULL.
 * Synchronous with respect to the host on return (the reference ends with cudaDeviceSynchronize, :1158).
 * A product with 2^32 or more candidate block pairs (more than one task list can index) is run block-row panel after panel
 * and concatenated; stats then hold the sums over the panels (stage lines are printed per panel). */
int bmsp_spgemm(bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, int verbose,
                void *stream, bmsp_spgemm_stats *stats);

/* The two halves of bmsp_spgemm for callers that multiply the same sparsity pattern again and again (new values in A / B, same
 * structure: time stepping, AMG set-ups).  The reference has one entry point, bmSparse_mult, whose stages T_1 ... T_9 are the symbolic half
 * (src/bmSparse_SPGEMM.cu:849-1107: C's keys, bitmaps, offsets, nnz) and T_7 the numeric half (:1109-1158).
 *   bmsp_spgemm_symbolic: C's structure only; *C's values are allocated and zero; where the product was formed through a task list, C
 *                         keeps that list (8 bytes per surviving pair + 4 per C tile, freed with C).  Same arguments as bmsp_spgemm otherwise.
 *   bmsp_spgemm_numeric : C must hold the structure of A x B (from bmsp_spgemm / _symbolic on operands of the SAME structure); its values
 *                         are overwritten with those of A x B under tc_version's numerics: for the V15 numerics (tc_version 5, and fp32 /
 *                         fp64 operands under any tc_version) exactly what bmsp_spgemm would store, bit for bit; for the matrix-core
 *                         kernels (fp16 operands, tc_version 1..4: exact products, fp32 accumulation in the order of whichever kernel
 *                         runs) within the tolerance stated for them.  Every product is stamped with a fingerprint of its operands'
 *                         structures (dimensions, keys, bitmaps): a C stamped for other operands is refused (BMSP_ERR_INVALID).  For a
 *                         matching stamp: where a strip block-MAC applies (fp16 operands with tc_version 4, fp32 operands; block-rows of
 *                         C of at most 256 tiles) only that kernel runs; a C from bmsp_spgemm_symbolic that kept its task list runs the
 *                         tc_version's block-MAC kernel from it (any value type).  Otherwise -- no kept list, or a C without a stamp
 *                         (adopted arrays) -- the whole product runs, its structure is checked against C's (BMSP_ERR_INVALID on a
 *                         mismatch) and its values are copied.  After changing an operand's values in place call
 *                         bmsp_matrix_invalidate(m, 0) first (cached tile copies); after changing its structure, (m, 1). */
int bmsp_spgemm_symbolic(bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, void *stream,
                         bmsp_spgemm_stats *stats);
int bmsp_spgemm_numeric(bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t C, int tc_version, void *stream, bmsp_spgemm_stats *stats);

/* Hardware self test of the operand / result lane layout the K = 32 block-MAC relies on (v_mfma_f32_16x16x32_f16: lane l holds
 * A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15], D[4*(l>>4)+i][l&15]): one 16x16x32 product of asymmetric small integers against a host
 * loop.  *mismatches = number of wrong elements (0 on gfx950).  The reference's analogue is the fragment-layout assumption of
 * multiplyV12..V14 (src/bmSparse_SPGEMM.cu:548-552), which it never checks. */
int bmsp_selftest_mfma_layout(int *mismatches);
/* Hardware self test of the ACCUMULATION ORDER of v_mfma_f32_16x16x4_f32: D = C + A * B on random fp32 operands against the host chain
 * fmaf(a_k, b_k, sum) for k ascending -- the order of multiplyV15<float, float> (src/bmSparse_SPGEMM.cu:269-273 as nvcc contracts it).
 * *mismatches = number of result elements that differ in any bit (0 on gfx950).  With 0 the fp32 product (tc_version 5) CAN run its
 * block-MAC on the matrix cores with values bit-identical to the vector-ALU kernel (BMSP_MAC_F32MFMA=1 selects that kernel; it is not the
 * default: measured on MI355X it is bound by the 256-byte operand tiles it reads and no faster than the vector-ALU kernels). */
int bmsp_selftest_mfma_f32_chain(int *mismatches);
/* The same on the corner the exponent guard of the fp32 matrix-core kernels must know about: every product a normal number (~2^-123), signs
 * alternating, so that the partial sums cancel into the subnormal range.  *mismatches = result elements that differ from the host's fmaf
 * chain; *exp_floor (may be NULL) = the smallest sum of the two operands' smallest biased exponents those kernels then accept: 128 when
 * the instruction is the chain there too, 174 otherwise (products can then not cancel below 2^-126) -- below the floor V15's vector-ALU
 * kernel computes the product. */
int bmsp_selftest_mfma_f32_cancel(int *mismatches, int *exp_floor);
/* Hardware self test of the byte-permute form of bmp_calculator (src/bmSparse_SPGEMM.cu:787-810) the row-merge passes use: 2^20 pairs of
 * bitmaps of every density, the v_perm_b32 sign-replication product on the row-major copy of B's tile against the multiply form on the
 * tile as stored.  *mismatches = pairs whose products differ (0 on gfx950). */
int bmsp_selftest_tile_product(int *mismatches);

/* bb_segsort<K,T>(keys, vals, n, segs, length)  -- include/bb_segsort-master/bb_segsort.h:35-192,
 * instantiated by the reference with K = uint64_t, T = 16-byte task_list_elem (src/bmSparse_SPGEMM.cu:1010).
 * Sorts every segment [segs[i], segs[i+1]) (last ends at n) ascending by key, in place, STABLY
 * (bb_segsort is unstable).  d_vals may be NULL (keys only). val_bytes in {4, 8, 16}. */
int bmsp_segsort_u64(uint64_t *d_keys, void *d_vals, int val_bytes, int64_t n, const int *d_segs,
                     int64_t num_segs, void *stream);

/* ---- row-panel sharding (new; the reference is single-GPU).  One process per GPU: each rank calls these
 *      with its own rank id.  These three are the building blocks; bmsp_spgemm_sharded below runs the whole thing. ---- */

/* Splits A's block-rows into `parts` contiguous panels balanced by candidate-task count
 * (sum over the panel's A blocks of B's blocks in the matching block-row).  bounds receives parts+1
 * block-row indices. */
int bmsp_partition_rows(bmsp_matrix_t A, bmsp_matrix_t B, int parts, int64_t *bounds);
/* The sub-matrix of block-rows [brow_begin, brow_end) as a matrix that borrows m's arrays (no copy);
 * num_rows stays global so keys stay global.  Free the view before m. */
int bmsp_matrix_row_panel(bmsp_matrix_t m, int64_t brow_begin, int64_t brow_end, bmsp_matrix_t *view);
/* Concatenates `parts` row panels (each a product of bmsp_spgemm on a panel, given by its four device arrays)
 * into one matrix, re-basing offsets.  Arrays of pointers/sizes are host arrays of length parts. */
int bmsp_matrix_concat_panels(int num_rows, int num_cols, int parts, const int64_t *block_nums,
                              const int64_t *nnzs, uint64_t *const *d_keys, uint64_t *const *d_bmps,
                              uint64_t *const *d_offsets, void *const *d_values, bmsp_dtype dtype,
                              bmsp_matrix_t *out);

/* ---- the sharded operators themselves: one process per GPU, RCCL over xGMI (nothing in the reference to
 *      replace -- it is single-GPU: src/bmSparse_SPGEMM.cu:1226-1288 runs one product on device 0).  librccl is opened on the first
 *      bmsp_comm_* call, not at load time. -------------------------------------------------------------------------------- */
typedef struct bmsp_comm_s *bmsp_comm_t;
#define BMSP_COMM_ID_BYTES 128
/* rank 0 creates the 128-byte rendezvous id (ncclUniqueId) and hands it to the other ranks by any means (file, MPI, torch.distributed) */
int bmsp_comm_unique_id(void *id_bytes);
/* collective: every rank calls it with the same id, after bmsp_set_device(its GPU) */
int bmsp_comm_init(const void *id_bytes, int world, int rank, bmsp_comm_t *out);
/* the same from the environment, for the drop-in executables: BMSP_WORLD, BMSP_RANK and, when BMSP_WORLD > 1, BMSP_COMM_FILE (a path
 * every rank can reach: rank 0 writes the id there, the others wait for it) */
int bmsp_comm_init_from_env(bmsp_comm_t *out);
/* BMSP_COMM_NONCE (optional, a number the launcher gives every rank of one run) is written into / required from the rendezvous file;
 * rank 0 removes a pre-existing file first, readers refuse files without this run's nonce (and, when no nonce is set, files older than
 * 120 s), so a file left by a crashed run is never joined. */
/* An in-process "communicator" of `world` panels on the CURRENT device, no RCCL: bmsp_spgemm_sharded / bmsp_spmv_sharded then compute
 * the `world` panels one after another and move every panel into its slice with a device copy.  It shares the size gather -> slice
 * layout -> offset re-basing -> terminal offset code with the RCCL transport (only the byte movement differs): the way to run the
 * sharded operators' P > 1 logic on one GPU.  Nothing in the reference to replace (single-GPU). */
int bmsp_comm_init_loopback(int world, bmsp_comm_t *out);
int bmsp_comm_info(bmsp_comm_t c, int *rank, int *world);
/* The slice-layout arithmetic of the two exchanges, host only (no GPU call): where panel r of the sharded product lands in the whole C
 * (block_start / value_start: parts+1 entries, exclusive sums of the panels' block / value counts; offsets of panel r are re-based by
 * value_start[r]), and which rows of u panel r = block-rows [bounds[r], bounds[r+1]) delivers. */
int bmsp_shard_layout(int parts, const int64_t *block_nums, const int64_t *nnzs, int64_t *block_start, int64_t *value_start);
int bmsp_shard_row_slices(int num_rows, int parts, const int64_t *bounds, int64_t *row_start, int64_t *row_count);
int bmsp_comm_free(bmsp_comm_t c);

typedef struct {
    int world, rank;
    int64_t panel_block_row_begin, panel_block_row_end; /* this rank's block-rows of A */
    int64_t panel_tasks;                                /* surviving tasks of this rank's panel (SpGEMM) */
    int64_t exchange_bytes;                             /* bytes every rank holds after the exchange (whole C / whole y) */
    double exchange_us;                                 /* device time of the exchange on this rank (all rounds) */
    double exchange_exposed_us;                         /* ... of it, the part after this rank's last panel product had finished (not hidden behind compute) */
    double exchange_hidden_frac;                        /* 1 - exposed / total: share of the exchange that ran while panels were being multiplied */
    int rounds, gathered;                               /* rounds of panels per rank; 0 = owner keeps (no exchange: C is this rank's panel) */
} bmsp_shard_stats;

/* C = A * B with A cut into `world` block-row panels balanced by candidate-task count, B replicated (every rank passes the same A
 * and B), every rank multiplies its panel and ALL ranks return the whole C: the panels are broadcast straight into their final
 * slices (an allgatherv without padding or staging copies), offsets re-based in place.  stats = this rank's panel product. */
int bmsp_spgemm_sharded(bmsp_comm_t c, bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, int verbose,
                        void *stream, bmsp_spgemm_stats *stats, bmsp_shard_stats *shard);
/* The same with its two knobs.  gather = 1: as above, in `rounds` rounds per rank (0: the library's choice, 4 when world > 1; BMSP_SHARD_ROUNDS):
 * A is cut into rounds x world panels, rank r multiplies panels r, world + r, ...; after every round the panel sizes are gathered and the
 * round's panels are broadcast into their final slices on a second stream while the next round multiplies (shard->exchange_hidden_frac).
 * gather = 0, "owner keeps" ("If only the owner needs C, skip the gather and report it separately"): world panels, no
 * exchange; *C is this rank's panel of C (block-rows [panel_block_row_begin, panel_block_row_end), global keys) -- the loopback
 * communicator owns every panel and returns them concatenated. */
int bmsp_spgemm_sharded_ex(bmsp_comm_t c, bmsp_matrix_t A, bmsp_matrix_t B, bmsp_matrix_t *C, int mode, int tc_version, int verbose,
                           void *stream, bmsp_spgemm_stats *stats, bmsp_shard_stats *shard, int gather, int rounds);
/* u = A * v with A cut into block-row panels balanced by stored values, v replicated; every rank sweeps its panel (writing only its
 * own rows of u), the row slices are exchanged in place and all ranks return the whole u (num_rows entries).  The panel view and its sweep plan are cached on A across calls. */
int bmsp_spmv_sharded(bmsp_comm_t c, bmsp_matrix_t A, const void *d_v, void *d_u, int variant, void *stream, bmsp_shard_stats *shard);

/* ---- host CSR (class CSRMatrix, include/CSRMatrix.h:13-21; declared only in the reference; backed by
 *      cusp::csr_matrix<int,float,host_memory> and cusp::multiply) ---------------------------------- */
int bmsp_csr_from_mtx(const char *path, bmsp_csr_t *out);                    /* CSRMatrix(std::string) */
int bmsp_csr_from_arrays(int num_rows, int num_cols, int64_t nnz, const int *row_offsets, const int *cols,
                         const float *vals, bmsp_csr_t *out);                /* CSRMatrix(csr_matrix*) */
int bmsp_csr_info(bmsp_csr_t m, int *num_rows, int *num_cols, int64_t *nnz);
int bmsp_csr_arrays(bmsp_csr_t m, const int **row_offsets, const int **cols, const float **vals);
int bmsp_csr_multiply(bmsp_csr_t A, bmsp_csr_t B, bmsp_csr_t *C);            /* CSRMatrix::multiply */
int bmsp_csr_spmv(bmsp_csr_t A, const float *x, float *y);
/* The HOST forms of the two (the reference's CSRMatrix holds a cusp host matrix, so its multiply is cusp's host path --
 * cusp/system/detail/sequential/multiply/csr_spmv.h:56-73, csr_spgemm.h:39-157 and the omp variants): no GPU call is made; `threads`
 * host threads (0 = all).  The product's columns come out in the list order of the Gustavson accumulator (unsorted within a row,
 * csr_spgemm.h:153), numeric zeros dropped. */
int bmsp_csr_multiply_host(bmsp_csr_t A, bmsp_csr_t B, bmsp_csr_t *C, int threads);
int bmsp_csr_spmv_host(bmsp_csr_t A, const float *x, float *y, int threads);
int bmsp_csr_free(bmsp_csr_t m);

#ifdef __cplusplus
}
#endif
#endif /* BMSP_H_ */
