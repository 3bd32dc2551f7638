/*
 * bmsp_oracle.c -- CPU restatement of the bmSparse hot path.  TEST INFRASTRUCTURE ONLY
 * (see bmsp_oracle.h for who may load this and how it is pinned).
 *
 * Plain C11, no dependencies beyond libc/libm/OpenMP.  Written from the behaviour of the
 * reference (file:line cited per function, paths relative to /root/reference); no reference
 * source text is reproduced here.
 */
#define _GNU_SOURCE
#include "bmsp_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <ctype.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * fp16 rounding.  The reference casts every parsed double with (half)double
 * (bmSpMatrix.cu:141): CUDA's __double2half and include/half.hpp:373-374 both round to nearest
 * even directly from the double.  Checked against half.hpp itself by oracle/ref_half_check.cpp.
 * ---------------------------------------------------------------------------------------- */
uint16_t orc_f64_to_f16_bits(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    uint16_t sign = (uint16_t)((u >> 48) & 0x8000u);
    int64_t exp = (int64_t)((u >> 52) & 0x7ff);
    uint64_t man = u & 0xfffffffffffffull;
    if (exp == 0x7ff) /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (man ? (0x200u | (uint16_t)(man >> 42)) : 0));
    int64_t e = exp - 1023; /* unbiased */
    if (exp == 0) return sign; /* double subnormal -> 0 in half */
    if (e > 15) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf (RNE) */
    uint64_t full = man | (1ull << 52); /* 53-bit significand */
    int shift;                          /* bits to drop */
    uint32_t hexp;
    if (e >= -14) { shift = 42; hexp = (uint32_t)(e + 15); }
    else {
        shift = 42 + (int)(-14 - e); hexp = 0;
        if (shift > 54) return sign; /* below half of the smallest subnormal */
    }
    uint64_t kept = full >> shift;
    uint64_t rem = full & ((1ull << shift) - 1);
    uint64_t half = 1ull << (shift - 1);
    if (rem > half || (rem == half && (kept & 1))) kept++;
    uint32_t bits;
    if (hexp == 0) bits = (uint32_t)kept;                 /* subnormal; may carry into exp 1 */
    else bits = ((hexp - 1) << 10) + (uint32_t)kept;     /* kept includes the implicit 1 -> +1 exp */
    if (bits >= 0x7c00u) bits = 0x7c00u;
    return (uint16_t)(sign | bits);
}

double orc_f16_bits_to_f64(uint16_t h)
{
    int sign = h >> 15;
    int e = (h >> 10) & 0x1f;
    int m = h & 0x3ff;
    double v;
    if (e == 0) v = ldexp((double)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexp((double)(m | 0x400), e - 25);
    return sign ? -v : v;
}

double orc_round_to_dtype(double x, int dtype)
{
    if (dtype == ORC_F16) return orc_f16_bits_to_f64(orc_f64_to_f16_bits(x));
    if (dtype == ORC_F32) return (double)(float)x;
    return x;
}

/* ------------------------------------------------------------------------------------------
 * MatrixMarket reader
 * ---------------------------------------------------------------------------------------- */
void orc_coo_free(orc_coo *m)
{
    if (!m) return;
    free(m->rows); free(m->cols); free(m->vals);
    memset(m, 0, sizeof(*m));
}

static int cmp_rowcol_idx(const void *a, const void *b, void *ctx)
{
    const orc_coo *c = (const orc_coo *)ctx;
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    if (c->rows[x] != c->rows[y]) return c->rows[x] < c->rows[y] ? -1 : 1;
    if (c->cols[x] != c->cols[y]) return c->cols[x] < c->cols[y] ? -1 : 1;
    return x < y ? -1 : (x > y);
}

static void coo_permute(orc_coo *c, const int64_t *perm)
{
    int64_t n = c->nnz;
    int *r = malloc(sizeof(int) * (n ? n : 1)), *cc = malloc(sizeof(int) * (n ? n : 1));
    double *v = malloc(sizeof(double) * (n ? n : 1));
    for (int64_t i = 0; i < n; i++) { r[i] = c->rows[perm[i]]; cc[i] = c->cols[perm[i]]; v[i] = c->vals[perm[i]]; }
    free(c->rows); free(c->cols); free(c->vals);
    c->rows = r; c->cols = cc; c->vals = v;
}

static void coo_sort_rowcol(orc_coo *c)
{
    int64_t n = c->nnz;
    int64_t *perm = malloc(sizeof(int64_t) * (n ? n : 1));
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    qsort_r(perm, n, sizeof(int64_t), cmp_rowcol_idx, c);
    coo_permute(c, perm);
    free(perm);
}

int orc_mtx_read(const char *path, int strict, orc_coo *out)
{
    memset(out, 0, sizeof(*out));
    FILE *f = fopen(path, "r");
    if (!f) return -1; /* the reference silently builds an empty matrix (bmSpMatrix.cu:114-127); we refuse */
    char *line = NULL; size_t cap = 0;
    if (getline(&line, &cap, f) < 0) { fclose(f); free(line); return -2; }
    int symmetric, pattern = 0, complex_ = 0;
    if (!strict) {
        /* bmSpMatrix.cu:116-120: any first line containing "symmetric" (also skew-symmetric) */
        symmetric = strstr(line, "symmetric") != NULL;
        if (strstr(line, "pattern")) { fclose(f); free(line); return -3; } /* reference mis-parses: undefined */
        if (strstr(line, "complex")) { fclose(f); free(line); return -3; }
    } else {
        /* matrix_market.inl:71-97 banner: %%MatrixMarket matrix <storage> <type> <symmetry> */
        char t0[64], t1[64], t2[64], t3[64], t4[64];
        if (sscanf(line, "%63s %63s %63s %63s %63s", t0, t1, t2, t3, t4) != 5 ||
            strcmp(t0, "%%MatrixMarket") || strcmp(t1, "matrix")) { fclose(f); free(line); return -4; }
        if (strcmp(t2, "coordinate")) { fclose(f); free(line); return -5; }
        pattern = !strcmp(t3, "pattern");
        complex_ = !strcmp(t3, "complex");
        if (!pattern && !complex_ && strcmp(t3, "real") && strcmp(t3, "integer")) { fclose(f); free(line); return -4; }
        if (!strcmp(t4, "general")) symmetric = 0;
        else if (!strcmp(t4, "symmetric")) symmetric = 1;
        else { fclose(f); free(line); return -6; } /* hermitian / skew: not_implemented (matrix_market.inl:279-287) */
    }
    /* comment lines (bmSpMatrix.cu:123-124) */
    long pos;
    for (;;) {
        pos = ftell(f);
        if (getline(&line, &cap, f) < 0) { fclose(f); free(line); return -2; }
        const char *p = line;
        while (*p && isspace((unsigned char)*p)) p++;
        if (*p == '%' || *p == 0) continue;
        break;
    }
    (void)pos;
    long long nr, nc, nz;
    if (sscanf(line, "%lld %lld %lld", &nr, &nc, &nz) != 3) { fclose(f); free(line); return -2; }
    int64_t capn = symmetric ? 2 * nz : nz;
    out->num_rows = (int)nr; out->num_cols = (int)nc;
    out->rows = malloc(sizeof(int) * (capn ? capn : 1));
    out->cols = malloc(sizeof(int) * (capn ? capn : 1));
    out->vals = malloc(sizeof(double) * (capn ? capn : 1));
    int64_t n = 0;
    for (long long l = 0; l < nz; l++) {
        long long r, c; double v = 1.0, im;
        if (fscanf(f, "%lld %lld", &r, &c) != 2) { fclose(f); free(line); orc_coo_free(out); return -7; }
        if (!pattern && fscanf(f, "%lf", &v) != 1) { fclose(f); free(line); orc_coo_free(out); return -7; }
        if (complex_ && fscanf(f, "%lf", &im) != 1) { fclose(f); free(line); orc_coo_free(out); return -7; }
        if (strict && (r < 1 || c < 1 || r > nr || c > nc)) { fclose(f); free(line); orc_coo_free(out); return -8; }
        out->rows[n] = (int)(r - 1); out->cols[n] = (int)(c - 1); out->vals[n] = v; n++;
        if (symmetric && r != c) { /* bmSpMatrix.cu:142-147 ; matrix_market.inl:256-272 */
            out->rows[n] = (int)(c - 1); out->cols[n] = (int)(r - 1); out->vals[n] = v; n++;
        }
    }
    out->nnz = n;
    fclose(f); free(line);
    if (strict) coo_sort_rowcol(out); /* matrix_market.inl:295 */
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * builder: COO -> bmSparse  (bmSpMatrix.cu:45-109,163-216)
 * ---------------------------------------------------------------------------------------- */
typedef struct { const orc_coo *coo; int transposed; } sort_ctx;

/* block_order (bmSpMatrix.cu:45-74): (r/8, c/8, then (r,c) or, transposed, (c,r)); ties keep file order */
static int cmp_block_order(const void *a, const void *b, void *vctx)
{
    const sort_ctx *s = (const sort_ctx *)vctx;
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    int rx = s->coo->rows[x], cx = s->coo->cols[x], ry = s->coo->rows[y], cy = s->coo->cols[y];
    if (rx / 8 != ry / 8) return rx / 8 < ry / 8 ? -1 : 1;
    if (cx / 8 != cy / 8) return cx / 8 < cy / 8 ? -1 : 1;
    if (s->transposed) {
        if (cx != cy) return cx < cy ? -1 : 1;
        if (rx != ry) return rx < ry ? -1 : 1;
    } else {
        if (rx != ry) return rx < ry ? -1 : 1;
        if (cx != cy) return cx < cy ? -1 : 1;
    }
    return x < y ? -1 : (x > y);
}

void orc_bmsp_free(orc_bmsp *m)
{
    if (!m) return;
    free(m->keys); free(m->bmps); free(m->offsets); free(m->values);
    memset(m, 0, sizeof(*m));
}

int orc_bmsp_from_coo(const orc_coo *coo, int dtype, int transposed, orc_bmsp *out)
{
    memset(out, 0, sizeof(*out));
    int64_t n = coo->nnz;
    out->num_rows = coo->num_rows; out->num_cols = coo->num_cols;
    out->dtype = dtype; out->transposed = transposed;
    int64_t *perm = malloc(sizeof(int64_t) * (n ? n : 1));
    for (int64_t i = 0; i < n; i++) perm[i] = i;
    sort_ctx ctx = { coo, transposed };
    qsort_r(perm, n, sizeof(int64_t), cmp_block_order, &ctx);

    out->keys = malloc(8 * (n ? n : 1)); out->bmps = malloc(8 * (n ? n : 1));
    out->offsets = malloc(8 * (n ? n + 1 : 1)); out->values = malloc(8 * (n ? n : 1));
    int64_t nb = 0, nv = 0;
    uint64_t prev_key = ~0ull; int prev_pos = -1;
    for (int64_t i = 0; i < n; i++) {
        int r = coo->rows[perm[i]], c = coo->cols[perm[i]];
        /* coord_to_key (bmSpMatrix.cu:76-83) */
        uint64_t key = ((uint64_t)(r / 8) << 32) | (uint64_t)(uint32_t)(c / 8);
        /* coord_to_bmp (bmSpMatrix.cu:85-98) */
        int pos = transposed ? (c % 8) * 8 + (r % 8) : (r % 8) * 8 + (c % 8);
        double v = orc_round_to_dtype(coo->vals[perm[i]], dtype); /* (T)double, bmSpMatrix.cu:141 */
        if (nb == 0 || key != prev_key) { /* reduce_by_key (bmSpMatrix.cu:183-188) */
            out->keys[nb] = key; out->bmps[nb] = 0; out->offsets[nb] = (uint64_t)nv; nb++;
            prev_pos = -1;
        }
        if (pos == prev_pos) {
            /* duplicate coordinate: the reference counts it twice and corrupts the popcount addressing
             * (SURVEY.md 7 "hard parts"); the build's defined behaviour is to sum duplicates in file
             * order in the matrix's own precision. */
            out->values[nv - 1] = orc_round_to_dtype(out->values[nv - 1] + v, dtype);
        } else {
            out->bmps[nb - 1] |= 1ull << (63 - pos); /* bmp_sum (bmSpMatrix.cu:103-109) */
            out->values[nv++] = v;
        }
        prev_key = key; prev_pos = pos;
    }
    out->block_num = nb; out->nnz = nv;
    out->offsets[nb] = (uint64_t)nv; /* terminal element is an extra; the reference has block_num entries */
    free(perm);
    return 0;
}

/* generate_coo (bmSpMatrix.cu:320-363): bits MSB->LSB; row-major tiles unless built transposed */
int orc_bmsp_to_coo(const orc_bmsp *m, orc_coo *out)
{
    memset(out, 0, sizeof(*out));
    int64_t n = m->nnz;
    out->num_rows = m->num_rows; out->num_cols = m->num_cols; out->nnz = n;
    out->rows = malloc(sizeof(int) * (n ? n : 1)); out->cols = malloc(sizeof(int) * (n ? n : 1));
    out->vals = malloc(sizeof(double) * (n ? n : 1));
    int64_t k = 0;
    for (int64_t b = 0; b < m->block_num; b++) {
        int64_t brow = (int64_t)(m->keys[b] >> 32), bcol = (int64_t)(m->keys[b] & 0xffffffffull);
        for (int i = 0; i < 64; i++) {
            if (!(m->bmps[b] & (1ull << (63 - i)))) continue;
            int hi = i / 8, lo = i % 8;
            out->rows[k] = (int)(brow * 8 + (m->transposed ? lo : hi));
            out->cols[k] = (int)(bcol * 8 + (m->transposed ? hi : lo));
            out->vals[k] = m->values[k];
            k++;
        }
    }
    if (k != n) return -1;
    coo_sort_rowcol(out);
    return 0;
}

/* compare (bmSpMatrix.cu:381-432): walk both sorted COOs, skip comparand entries absent from the bmSparse
 * matrix, accumulate |e-r|/max(|e|,eps) with values below eps flushed to 0, return the mean over nnz */
double orc_bmsp_compare(const orc_bmsp *m, const orc_coo *other)
{
    orc_coo mine; orc_bmsp_to_coo(m, &mine);
    orc_coo oth = *other;
    oth.rows = malloc(sizeof(int) * (oth.nnz ? oth.nnz : 1)); oth.cols = malloc(sizeof(int) * (oth.nnz ? oth.nnz : 1));
    oth.vals = malloc(sizeof(double) * (oth.nnz ? oth.nnz : 1));
    memcpy(oth.rows, other->rows, sizeof(int) * other->nnz); memcpy(oth.cols, other->cols, sizeof(int) * other->nnz);
    memcpy(oth.vals, other->vals, sizeof(double) * other->nnz);
    coo_sort_rowcol(&oth);
    const double eps = 1e-8;
    double count = 0; int64_t off = 0;
    for (int64_t i = 0; i < mine.nnz; i++) {
        while (i + off < oth.nnz && (oth.rows[i + off] != mine.rows[i] || oth.cols[i + off] != mine.cols[i])) off++;
        if (i + off >= oth.nnz) { count = INFINITY; break; }
        double e = fabs(oth.vals[i + off]) < eps ? 0 : oth.vals[i + off];
        double r = fabs(mine.vals[i]) < eps ? 0 : mine.vals[i];
        count += fabs(e - r) / fmax(fabs(e), eps);
    }
    double res = mine.nnz ? count / (double)mine.nnz : 0.0;
    orc_coo_free(&mine); orc_coo_free(&oth);
    return res;
}

/* ------------------------------------------------------------------------------------------
 * SpMV (SPMV.cu:72-82,153-189).  Dense block-row pointer (the reference's compressed pointer is
 * only right when no block-row is empty: SURVEY.md 7).
 * ---------------------------------------------------------------------------------------- */
static int64_t *block_row_ptr(const orc_bmsp *m, int64_t nbr)
{
    int64_t *ptr = calloc((size_t)nbr + 1, sizeof(int64_t));
    for (int64_t b = 0; b < m->block_num; b++) {
        int64_t r = (int64_t)(m->keys[b] >> 32);
        if (r < nbr) ptr[r + 1]++;
    }
    for (int64_t r = 0; r < nbr; r++) ptr[r + 1] += ptr[r];
    return ptr;
}

/* shmem_load (SPMV.cu:72-82): element at tile position p, 0 when the bit is clear;
 * rank = popcount(bmp >> (64-p)) with p==0 meaning rank 0 (the reference shifts by 64 there) */
static inline double tile_elem(uint64_t bmp, const double *vals, int p)
{
    if (!(bmp & (1ull << (63 - p)))) return 0.0;
    int rank = p ? __builtin_popcountll(bmp >> (64 - p)) : 0;
    return vals[rank];
}

int orc_spmv_f32(const orc_bmsp *A, const float *v, float *u)
{
    if (A->transposed) return -1;
    int64_t nbr = ((int64_t)A->num_rows + 7) / 8;
    int64_t *ptr = block_row_ptr(A, nbr);
    for (int64_t br = 0; br < nbr; br++) {
        float res[64];
        for (int t = 0; t < 64; t++) res[t] = 0.0f;
        for (int64_t b = ptr[br]; b < ptr[br + 1]; b++) {
            int64_t bcol = (int64_t)(A->keys[b] & 0xffffffffull);
            const double *vals = A->values + A->offsets[b];
            for (int t = 0; t < 64; t++) {
                int64_t c = bcol * 8 + (t % 8);
                float x = c < A->num_cols ? v[c] : 0.0f; /* SPMV.cu:176 reads past the end; defined as 0 */
                float a = (float)tile_elem(A->bmps[b], vals, t);
                res[t] = fmaf(a, x, res[t]); /* res += (VO)(a*x), contracted by nvcc for float */
            }
        }
        /* 8-lane shfl_down tree (SPMV.cu:180-181): +4, +2, +1 */
        for (int r = 0; r < 8; r++) {
            float *p = res + r * 8;
            float s = ((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7]));
            int64_t row = br * 8 + r;
            if (row < A->num_rows) u[row] = s;
        }
    }
    free(ptr);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * SpGEMM (SPGEMM.cu:827-1223)
 * ---------------------------------------------------------------------------------------- */
/* bmp_calculator (SPGEMM.cu:787-810): bit(i,j) = OR_k A(i,k) & B(k,j); A row-bytes, B column-bytes */
uint64_t orc_bmp_product(uint64_t a, uint64_t bt)
{
    uint64_t res = 0;
    for (int i = 0; i < 8; i++) {
        uint64_t ra = (a >> (56 - 8 * i)) & 0xff;
        for (int j = 0; j < 8; j++) {
            uint64_t cb = (bt >> (56 - 8 * j)) & 0xff;
            if (ra & cb) res |= 1ull << (63 - (i * 8 + j));
        }
    }
    return res;
}

/* multiplication_checker (SPGEMM.cu:742-757): true (= remove) iff the boolean product is empty */
int orc_bmp_product_empty(uint64_t a, uint64_t bt) { return orc_bmp_product(a, bt) == 0; }

typedef struct { int64_t a, b; uint64_t ckey; int64_t seq; } task_t;

static int cmp_task(const void *x, const void *y)
{
    const task_t *p = (const task_t *)x, *q = (const task_t *)y;
    if (p->ckey != q->ckey) return p->ckey < q->ckey ? -1 : 1; /* is_less_ik (SPGEMM.cu:85-94) */
    return p->seq < q->seq ? -1 : (p->seq > q->seq);          /* stable: k ascending */
}

int orc_spgemm(const orc_bmsp *A, const orc_bmsp *B, int exact_products, orc_bmsp *C, orc_spgemm_stats *st)
{
    memset(C, 0, sizeof(*C));
    orc_spgemm_stats local; if (!st) st = &local;
    memset(st, 0, sizeof(*st));
    if (A->transposed || !B->transposed) return -1; /* SPGEMM.cu:1261-1262 */
    if (A->num_cols != B->num_rows) return -2;
    if (A->dtype != B->dtype) return -3;
    int64_t nbrB = ((int64_t)B->num_rows + 7) / 8;
    int64_t *pos = block_row_ptr(B, nbrB); /* T_1 + first scan of T_3 (SPGEMM.cu:839-847,877-878) */

    /* T_2/T_3: expansion, tasks ordered by a then by B block index (SPGEMM.cu:857-932) */
    int64_t total = 0;
    for (int64_t a = 0; a < A->block_num; a++) {
        int64_t col = (int64_t)(A->keys[a] & 0xffffffffull);
        if (col < nbrB) total += pos[col + 1] - pos[col];
    }
    st->task_list_size = total;
    task_t *tasks = malloc(sizeof(task_t) * (total ? total : 1));
    int64_t nt = 0;
    for (int64_t a = 0; a < A->block_num; a++) {
        int64_t col = (int64_t)(A->keys[a] & 0xffffffffull);
        if (col >= nbrB) continue;
        for (int64_t b = pos[col]; b < pos[col + 1]; b++) {
            /* T_4: bitmap filter (SPGEMM.cu:944-948) */
            if (orc_bmp_product_empty(A->bmps[a], B->bmps[b])) continue;
            tasks[nt].a = a; tasks[nt].b = b;
            /* task_elem_to_C_key (SPGEMM.cu:111-119) */
            tasks[nt].ckey = (A->keys[a] & 0xffffffff00000000ull) | (B->keys[b] & 0xffffffffull);
            tasks[nt].seq = nt; nt++;
        }
    }
    st->surviving_tasks = nt; st->bmp_reduction = total - nt;
    /* T_5: group by C key (SPGEMM.cu:963-1016) */
    qsort(tasks, nt, sizeof(task_t), cmp_task);

    /* T_6 + T_9: C layout (SPGEMM.cu:1031-1107) */
    int64_t cs = 0;
    for (int64_t t = 0; t < nt; t++) if (t == 0 || tasks[t].ckey != tasks[t - 1].ckey) cs++;
    C->num_rows = A->num_rows; C->num_cols = B->num_cols; C->dtype = A->dtype == ORC_F64 ? ORC_F64 : ORC_F32; C->transposed = 0;
    C->block_num = cs;
    C->keys = malloc(8 * (cs ? cs : 1)); C->bmps = calloc(cs ? cs : 1, 8); C->offsets = malloc(8 * (cs + 1));
    int64_t *first_task = malloc(sizeof(int64_t) * (cs + 1)); /* inclusive ends, stored shifted by one */
    int64_t c = -1;
    for (int64_t t = 0; t < nt; t++) {
        if (t == 0 || tasks[t].ckey != tasks[t - 1].ckey) { c++; C->keys[c] = tasks[t].ckey; first_task[c] = t; }
        C->bmps[c] |= orc_bmp_product(A->bmps[tasks[t].a], B->bmps[tasks[t].b]);
    }
    first_task[cs] = nt;
    uint64_t nnz = 0;
    for (int64_t i = 0; i < cs; i++) { C->offsets[i] = nnz; nnz += (uint64_t)__builtin_popcountll(C->bmps[i]); }
    C->offsets[cs] = nnz; C->nnz = (int64_t)nnz;
    st->c_blocks = cs; st->c_nnz = (int64_t)nnz;
    C->values = malloc(8 * (nnz ? nnz : 1));

    /* T_7: block multiply-accumulate (multiplyV15, SPGEMM.cu:204-291 ; tensor variants :294-733) */
    for (int64_t i = 0; i < cs; i++) {
        float acc[64];
        double accd[64]; /* fp64 inputs accumulate in fp64 (the build's bmSpMatrix<double> product) */
        for (int l = 0; l < 64; l++) { acc[l] = 0.0f; accd[l] = 0.0; }
        for (int64_t t = first_task[i]; t < first_task[i + 1]; t++) {
            int64_t a = tasks[t].a, b = tasks[t].b;
            double ta[64], tb[64];
            for (int p = 0; p < 64; p++) {
                ta[p] = tile_elem(A->bmps[a], A->values + A->offsets[a], p);
                /* B is stored column-major inside the tile: B(k,j) sits at position j*8+k (SPGEMM.cu:219,263-264) */
                int k = p / 8, j = p % 8;
                tb[p] = tile_elem(B->bmps[b], B->values + B->offsets[b], j * 8 + k);
            }
            for (int k = 0; k < 8; k++) { /* stored a_ik times stored b_kj, by bitmap (explicit zeros count) */
                uint64_t colk = 0x0101010101010101ull << (7 - k);
                st->scalar_products += (int64_t)__builtin_popcountll(A->bmps[a] & colk) *
                                       (int64_t)__builtin_popcountll(B->bmps[b] & colk);
            }
            for (int l = 0; l < 64; l++) {
                int r = l / 8, j = l % 8;
                for (int k = 0; k < 8; k++) {
                    double av = ta[r * 8 + k], bv = tb[k * 8 + j];
                    if (A->dtype == ORC_F16 && !exact_products) {
                        /* __half * __half rounds to half, then added in float (SPGEMM.cu:271-272) */
                        float prod = (float)orc_round_to_dtype(av * bv, ORC_F16);
                        acc[l] = acc[l] + prod;
                    } else if (A->dtype == ORC_F64) {
                        accd[l] = fma(av, bv, accd[l]);
                    } else {
                        /* fp32: one rounding per step; fp16 exact-product path: the product is exact in fp32 */
                        acc[l] = fmaf((float)av, (float)bv, acc[l]);
                    }
                }
            }
        }
        /* write bitmap-selected entries (SPGEMM.cu:278-287) */
        uint64_t bmp = C->bmps[i]; int64_t w = (int64_t)C->offsets[i];
        for (int p = 0; p < 64; p++)
            if (bmp & (1ull << (63 - p))) C->values[w++] = A->dtype == ORC_F64 ? accd[p] : (double)acc[p];
    }
    free(first_task); free(tasks); free(pos);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * segmented sort gold (bb_segsort-master/main.cu:121-143): stable sort inside each segment.
 * segs has nseg starts; segment s is [segs[s], segs[s+1]) and the last one ends at n (bb_bin.h:67).
 * ---------------------------------------------------------------------------------------- */
typedef struct { uint64_t k, v0, v1; int64_t seq; } kv_t;
static int cmp_kv(const void *x, const void *y)
{
    const kv_t *p = (const kv_t *)x, *q = (const kv_t *)y;
    if (p->k != q->k) return p->k < q->k ? -1 : 1;
    return p->seq < q->seq ? -1 : (p->seq > q->seq);
}
int orc_segsort_u64_kv(uint64_t *keys, uint64_t *vals2, int64_t n, const int64_t *segs, int64_t nseg)
{
    for (int64_t s = 0; s < nseg; s++) {
        int64_t lo = segs[s], hi = (s + 1 < nseg) ? segs[s + 1] : n;
        if (lo < 0 || hi > n || lo > hi) return -1;
        int64_t m = hi - lo;
        kv_t *tmp = malloc(sizeof(kv_t) * (m ? m : 1));
        for (int64_t i = 0; i < m; i++) { tmp[i].k = keys[lo + i]; tmp[i].v0 = vals2[2 * (lo + i)]; tmp[i].v1 = vals2[2 * (lo + i) + 1]; tmp[i].seq = i; }
        qsort(tmp, m, sizeof(kv_t), cmp_kv);
        for (int64_t i = 0; i < m; i++) { keys[lo + i] = tmp[i].k; vals2[2 * (lo + i)] = tmp[i].v0; vals2[2 * (lo + i) + 1] = tmp[i].v1; }
        free(tmp);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CPU baseline: host CSR SpMV / SpGEMM as cusp::multiply runs them
 * ---------------------------------------------------------------------------------------- */
void orc_csr_free(orc_csr *m)
{
    if (!m) return;
    free(m->row_offsets); free(m->cols); free(m->vals);
    memset(m, 0, sizeof(*m));
}

int orc_csr_from_coo(const orc_coo *coo, orc_csr *out)
{
    memset(out, 0, sizeof(*out));
    int64_t n = coo->nnz;
    out->num_rows = coo->num_rows; out->num_cols = coo->num_cols; out->nnz = n;
    out->row_offsets = calloc((size_t)coo->num_rows + 1, sizeof(int));
    out->cols = malloc(sizeof(int) * (n ? n : 1)); out->vals = malloc(sizeof(float) * (n ? n : 1));
    for (int64_t i = 0; i < n; i++) {
        if (i && (coo->rows[i] < coo->rows[i - 1])) return -1; /* must be row-sorted */
        out->row_offsets[coo->rows[i] + 1]++;
        out->cols[i] = coo->cols[i]; out->vals[i] = (float)coo->vals[i];
    }
    for (int r = 0; r < coo->num_rows; r++) out->row_offsets[r + 1] += out->row_offsets[r];
    return 0;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* csr_spmv.h:56-73 (sequential), omp/.../csr_spmv.h:67-85 (row-parallel).  accumulator starts at 0
 * (generic/multiply.inl:98-105), summed in column order */
int orc_csr_spmv(const orc_csr *A, const float *x, float *y, int threads)
{
    int nr = A->num_rows;
    if (threads <= 1) {
        for (int i = 0; i < nr; i++) {
            float acc = 0.0f;
            for (int jj = A->row_offsets[i]; jj < A->row_offsets[i + 1]; jj++) acc = acc + A->vals[jj] * x[A->cols[jj]];
            y[i] = acc;
        }
    } else {
#pragma omp parallel for num_threads(threads) schedule(static)
        for (int i = 0; i < nr; i++) {
            float acc = 0.0f;
            for (int jj = A->row_offsets[i]; jj < A->row_offsets[i + 1]; jj++) acc = acc + A->vals[jj] * x[A->cols[jj]];
            y[i] = acc;
        }
    }
    return 0;
}

/* Gustavson two-pass SpGEMM.  threads<=1: sequential/multiply/csr_spgemm.h:39-157 (numeric zeros dropped,
 * :135; columns unsorted within a row, :153).  threads>1: omp/.../csr_spgemm.h:40-87,93-.. (per-thread
 * mask/next/sums, numeric zeros kept). */
int orc_csr_spgemm(const orc_csr *A, const orc_csr *B, orc_csr *C, int threads, int64_t *products)
{
    memset(C, 0, sizeof(*C));
    if (A->num_cols != B->num_rows) return -1;
    int nr = A->num_rows, ncol = B->num_cols;
    C->num_rows = nr; C->num_cols = ncol;
    C->row_offsets = calloc((size_t)nr + 1, sizeof(int));
    int64_t prods = 0;
    if (threads <= 1) {
        /* pass 1 */
        int64_t *mask = malloc(sizeof(int64_t) * (ncol ? ncol : 1));
        for (int k = 0; k < ncol; k++) mask[k] = -1;
        int64_t nnz = 0;
        for (int i = 0; i < nr; i++)
            for (int jj = A->row_offsets[i]; jj < A->row_offsets[i + 1]; jj++) {
                int j = A->cols[jj];
                for (int kk = B->row_offsets[j]; kk < B->row_offsets[j + 1]; kk++) {
                    int k = B->cols[kk];
                    if (mask[k] != i) { mask[k] = i; nnz++; }
                }
            }
        free(mask);
        C->cols = malloc(sizeof(int) * (nnz ? nnz : 1)); C->vals = malloc(sizeof(float) * (nnz ? nnz : 1));
        /* pass 2 */
        int *next = malloc(sizeof(int) * (ncol ? ncol : 1)); float *sums = calloc(ncol ? ncol : 1, sizeof(float));
        for (int k = 0; k < ncol; k++) next[k] = -1;
        int64_t out = 0;
        for (int i = 0; i < nr; i++) {
            int head = -2, length = 0;
            for (int jj = A->row_offsets[i]; jj < A->row_offsets[i + 1]; jj++) {
                int j = A->cols[jj]; float v = A->vals[jj];
                for (int kk = B->row_offsets[j]; kk < B->row_offsets[j + 1]; kk++) {
                    int k = B->cols[kk];
                    sums[k] = sums[k] + v * B->vals[kk]; prods++;
                    if (next[k] == -1) { next[k] = head; head = k; length++; }
                }
            }
            for (int jj = 0; jj < length; jj++) {
                if (sums[head] != 0.0f) { C->cols[out] = head; C->vals[out] = sums[head]; out++; }
                int tmp = head; head = next[head]; next[tmp] = -1; sums[tmp] = 0.0f;
            }
            C->row_offsets[i + 1] = (int)out;
        }
        C->nnz = out;
        free(next); free(sums);
    } else {
#pragma omp parallel num_threads(threads)
        {
            int *mask = malloc(sizeof(int) * (ncol ? ncol : 1));
            for (int k = 0; k < ncol; k++) mask[k] = -1;
#pragma omp for schedule(dynamic, 64)
            for (int i = 0; i < nr; i++) {
                int cnt = 0;
                for (int jj = A->row_offsets[i]; jj < A->row_offsets[i + 1]; jj++) {
                    int j = A->cols[jj];
                    for (int kk = B->row_offsets[j]; kk < B->row_offsets[j + 1]; kk++) {
                        int k = B->cols[kk];
                        if (mask[k] != i) { mask[k] = i; cnt++; }
                    }
                }
                C->row_offsets[i + 1] = cnt;
            }
            free(mask);
        }
        for (int i = 0; i < nr; i++) C->row_offsets[i + 1] += C->row_offsets[i];
        int64_t nnz = C->row_offsets[nr];
        C->nnz = nnz;
        C->cols = malloc(sizeof(int) * (nnz ? nnz : 1)); C->vals = malloc(sizeof(float) * (nnz ? nnz : 1));
#pragma omp parallel num_threads(threads) reduction(+ : prods)
        {
            int *next = malloc(sizeof(int) * (ncol ? ncol : 1)); float *sums = calloc(ncol ? ncol : 1, sizeof(float));
            for (int k = 0; k < ncol; k++) next[k] = -1;
#pragma omp for schedule(dynamic, 64)
            for (int i = 0; i < nr; i++) {
                int head = -2, length = 0;
                for (int jj = A->row_offsets[i]; jj < A->row_offsets[i + 1]; jj++) {
                    int j = A->cols[jj]; float v = A->vals[jj];
                    for (int kk = B->row_offsets[j]; kk < B->row_offsets[j + 1]; kk++) {
                        int k = B->cols[kk];
                        sums[k] = sums[k] + v * B->vals[kk]; prods++;
                        if (next[k] == -1) { next[k] = head; head = k; length++; }
                    }
                }
                int64_t off = C->row_offsets[i];
                for (int jj = 0; jj < length; jj++) {
                    C->cols[off] = head; C->vals[off] = sums[head]; off++;
                    int tmp = head; head = next[head]; next[tmp] = -1; sums[tmp] = 0.0f;
                }
            }
            free(next); free(sums);
        }
    }
    if (products) *products = prods;
    return 0;
}
