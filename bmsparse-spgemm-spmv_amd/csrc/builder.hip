// builder.hip -- MatrixMarket ingestion and the COO -> 8x8 bitmap-block (bmSparse) builder on the device.
//
// Reference behaviour: bmSpMatrix<T>::bmSpMatrix(path, transposed), src/bmSpMatrix.cu:111-219 -- host text
// parse, comparator sort of (row,col,val) tuples by (block row, block col, intra-tile order), then
// reduce_by_key passes for keys / offsets / bitmaps.  Here the comparator sort becomes ONE radix sort of a
// packed integer key  (block_row | block_col | 6-bit tile position)  restricted to the bits the matrix uses,
// and keys / offsets / bitmaps / duplicate-summed values come out of a single scan-fused pass.
#include "matrix.h"
#include "prims.hip.h"
#include <algorithm>
#include <cctype>
#include <memory>
#include <thread>
#include <string>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace bmsp {

// ------------------------------------------------------------------------------------------------
// MatrixMarket reader (host)
// ------------------------------------------------------------------------------------------------
namespace {
struct MappedFile {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~MappedFile()
    {
        if (p && n) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};

inline const char *skip_ws(const char *s, const char *e)
{
    while (s < e && (*s == ' ' || *s == '\t' || *s == '\r' || *s == '\n')) s++;
    return s;
}
inline const char *skip_line(const char *s, const char *e)
{
    while (s < e && *s != '\n') s++;
    return s < e ? s + 1 : e;
}
inline bool parse_i64(const char *&s, const char *e, long long &out)
{
    s = skip_ws(s, e);
    if (s >= e) return false;
    bool neg = false;
    if (*s == '-' || *s == '+') { neg = *s == '-'; s++; }
    if (s >= e || *s < '0' || *s > '9') return false;
    long long v = 0;
    while (s < e && *s >= '0' && *s <= '9') { v = v * 10 + (*s - '0'); s++; }
    out = neg ? -v : v;
    return true;
}
// decimal -> double.  Fast path (exactly rounded): at most 15 significant digits and |exponent| <= 22, so mantissa and
// power of ten are both exact doubles and one multiplication / division rounds once.  Everything else goes to strtod.
inline bool parse_f64(const char *&s, const char *e, double &out)
{
    s = skip_ws(s, e);
    if (s >= e) return false;
    {
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
        const char *q = s;
        bool neg = false;
        if (q < e && (*q == '-' || *q == '+')) { neg = *q == '-'; q++; }
        uint64_t mant = 0;
        int digits = 0, frac = 0;
        bool any = false, ok = true;
        while (q < e && *q >= '0' && *q <= '9') { if (mant || *q != '0') digits++; mant = mant * 10 + (uint64_t)(*q - '0'); q++; any = true; if (digits > 15) ok = false; }
        if (q < e && *q == '.') {
            q++;
            while (q < e && *q >= '0' && *q <= '9') { if (mant || *q != '0') digits++; mant = mant * 10 + (uint64_t)(*q - '0'); q++; frac++; any = true; if (digits > 15) ok = false; }
        }
        int ex = 0;
        if (any && ok && q < e && (*q == 'e' || *q == 'E')) {
            const char *r = q + 1;
            bool eneg = false;
            if (r < e && (*r == '-' || *r == '+')) { eneg = *r == '-'; r++; }
            if (r < e && *r >= '0' && *r <= '9') {
                int v = 0;
                while (r < e && *r >= '0' && *r <= '9' && v < 10000) { v = v * 10 + (*r - '0'); r++; }
                ex = eneg ? -v : v;
                q = r;
            } else ok = false;
        }
        const bool at_end = q >= e || *q == ' ' || *q == '\t' || *q == '\r' || *q == '\n';
        const int e10 = ex - frac;
        if (any && ok && at_end && e10 >= -22 && e10 <= 22) {
            double v = (double)mant;
            v = e10 < 0 ? v / p10[-e10] : v * p10[e10];
            out = neg ? -v : v;
            s = q;
            return true;
        }
    }
    char buf[64];
    size_t k = 0;
    while (s + k < e && k < sizeof(buf) - 1 && !(s[k] == ' ' || s[k] == '\t' || s[k] == '\r' || s[k] == '\n')) {
        buf[k] = s[k];
        k++;
    }
    buf[k] = 0;
    char *endp = nullptr;
    out = strtod(buf, &endp);
    if (endp == buf) return false;
    s += k;
    return true;
}
}  // namespace

void read_matrix_market(const std::string &path_in, HostCoo &out)
{
    std::string path = path_in;
    struct stat sb;
    if (stat(path.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) {
        std::string alt = path + ".mtx";  // the reference's mains pass the name with and without the suffix
        if (stat(alt.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode))
            fail(BMSP_ERR_IO, "cannot open MatrixMarket file '%s' (also tried '%s')", path.c_str(), alt.c_str());
        path = alt;
    }
    MappedFile mf;
    mf.fd = open(path.c_str(), O_RDONLY);
    if (mf.fd < 0) fail(BMSP_ERR_IO, "cannot open '%s': %s", path.c_str(), strerror(errno));
    mf.n = (size_t)sb.st_size;
    if (mf.n == 0) fail(BMSP_ERR_IO, "'%s' is empty", path.c_str());
    void *mp = mmap(nullptr, mf.n, PROT_READ, MAP_PRIVATE, mf.fd, 0);
    if (mp == MAP_FAILED) { mf.n = 0; fail(BMSP_ERR_IO, "mmap('%s') failed: %s", path.c_str(), strerror(errno)); }
    mf.p = (const char *)mp;
    const char *s = mf.p, *e = mf.p + mf.n;

    // banner
    const char *l0 = s;
    s = skip_line(s, e);
    std::string banner(l0, (size_t)(s - l0));
    bool pattern = false, cplx = false;
    int sym = 0;  // 0 general, 1 symmetric/hermitian, -1 skew-symmetric
    if (banner.rfind("%%MatrixMarket", 0) == 0) {
        char t1[64] = "", t2[64] = "", t3[64] = "", t4[64] = "";
        sscanf(banner.c_str() + 14, "%63s %63s %63s %63s", t1, t2, t3, t4);
        for (char *t : {t1, t2, t3, t4})
            for (char *q = t; *q; q++) *q = (char)tolower(*q);
        if (strcmp(t1, "matrix") != 0) fail(BMSP_ERR_IO, "'%s': not a MatrixMarket matrix", path.c_str());
        if (strcmp(t2, "coordinate") != 0)
            fail(BMSP_ERR_UNSUPPORTED, "'%s': only coordinate storage is supported (got '%s')", path.c_str(), t2);
        pattern = !strcmp(t3, "pattern");
        cplx = !strcmp(t3, "complex");
        if (!pattern && !cplx && strcmp(t3, "real") && strcmp(t3, "integer"))
            fail(BMSP_ERR_IO, "'%s': unknown MatrixMarket field '%s'", path.c_str(), t3);
        if (!strcmp(t4, "general")) sym = 0;
        else if (!strcmp(t4, "symmetric") || !strcmp(t4, "hermitian")) sym = 1;
        else if (!strcmp(t4, "skew-symmetric")) sym = -1;
        else fail(BMSP_ERR_IO, "'%s': unknown MatrixMarket symmetry '%s'", path.c_str(), t4);
    } else {
        // no banner: the reference still consumes the first line and mirrors when it mentions "symmetric"
        sym = banner.find("symmetric") != std::string::npos ? 1 : 0;
    }
    // comment lines
    for (;;) {
        const char *t = skip_ws(s, e);
        if (t < e && *t == '%') { s = skip_line(t, e); continue; }
        s = t;
        break;
    }
    long long nr, nc, nz;
    if (!parse_i64(s, e, nr) || !parse_i64(s, e, nc) || !parse_i64(s, e, nz) || nr < 0 || nc < 0 || nz < 0)
        fail(BMSP_ERR_IO, "'%s': bad size line", path.c_str());
    if (nr > 0x7fffffffLL || nc > 0x7fffffffLL) fail(BMSP_ERR_LIMIT, "'%s': dimensions exceed int32", path.c_str());
    out.num_rows = (int)nr;
    out.num_cols = (int)nc;
    // body: nz entry lines, parsed by several host threads on line-aligned slices of the mapped file
    const char *body = s;
    unsigned hw = std::thread::hardware_concurrency();
    try { hw = std::min<unsigned>(hw ? hw : 1u, (unsigned)std::max<long>(1, sysconf(_SC_NPROCESSORS_ONLN))); } catch (...) {}
    size_t nthreads = std::min<size_t>({(size_t)(hw ? hw : 1), (size_t)16, (size_t)((e - body) >> 20) + 1});
    if (const char *env = getenv("BMSP_PARSE_THREADS")) nthreads = std::max(1, atoi(env));
    struct Part { std::vector<int> r, c; std::vector<double> v; long long lines = 0; std::string err; };
    std::vector<Part> parts(nthreads);
    std::vector<const char *> cut(nthreads + 1);
    cut[0] = body; cut[nthreads] = e;
    for (size_t t = 1; t < nthreads; t++) {
        const char *p = body + (size_t)(e - body) * t / nthreads;
        while (p < e && *p != '\n') p++;
        cut[t] = p < e ? p + 1 : e;
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    auto work = [&](size_t t) {
        Part &P = parts[t];
        const char *q = cut[t], *qe = cut[t + 1];
        size_t guess = (size_t)(qe - q) / 12 + 16;
        P.r.reserve(sym ? 2 * guess : guess); P.c.reserve(sym ? 2 * guess : guess); P.v.reserve(sym ? 2 * guess : guess);
        char msg[160];
        for (;;) {
            q = skip_ws(q, qe);
            if (q >= qe) break;
            long long r, c;
            double v = 1.0, im;
            if (!parse_i64(q, qe, r) || !parse_i64(q, qe, c)) { P.err = "malformed entry line"; return; }
            if (!pattern && !parse_f64(q, qe, v)) { P.err = "missing value"; return; }
            if (cplx && !parse_f64(q, qe, im)) { P.err = "missing imaginary part"; return; }
            if (r < 1 || c < 1 || r > nr || c > nc) {
                snprintf(msg, sizeof msg, "index (%lld,%lld) outside %lldx%lld", r, c, nr, nc);
                P.err = msg;
                return;
            }
            P.lines++;
            P.r.push_back((int)(r - 1)); P.c.push_back((int)(c - 1)); P.v.push_back(v);
            if (sym && r != c) {  // mirror off-diagonal entries (src/bmSpMatrix.cu:142-147)
                P.r.push_back((int)(c - 1)); P.c.push_back((int)(r - 1)); P.v.push_back(sym < 0 ? -v : v);
            }
        }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (size_t t = 0; t < nthreads; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    long long lines = 0;
    size_t total = 0;
    for (auto &P : parts) {
        if (!P.err.empty()) fail(BMSP_ERR_IO, "'%s': %s", path.c_str(), P.err.c_str());
        lines += P.lines;
        total += P.r.size();
    }
    if (lines < nz) fail(BMSP_ERR_IO, "'%s': unexpected end of file: %lld of %lld entries", path.c_str(), lines, nz);
    if (lines > nz) fail(BMSP_ERR_IO, "'%s': %lld entry lines but the size line announces %lld", path.c_str(), lines, nz);
    out.rows.resize(total); out.cols.resize(total); out.vals.resize(total);
    size_t at = 0;
    for (auto &P : parts) {
        std::copy(P.r.begin(), P.r.end(), out.rows.begin() + at);
        std::copy(P.c.begin(), P.c.end(), out.cols.begin() + at);
        std::copy(P.v.begin(), P.v.end(), out.vals.begin() + at);
        at += P.r.size();
    }
}

// ------------------------------------------------------------------------------------------------
// device builder
// ------------------------------------------------------------------------------------------------
namespace {

struct MakeSortKey {
    const int *rows, *cols;
    uint64_t *keys;
    uint32_t *perm;
    int cbits;
    int transposed;
    __device__ void operator()(uint64_t i) const
    {
        uint32_t r = (uint32_t)rows[i], c = (uint32_t)cols[i];
        // intra-tile position: coord_to_bmp (src/bmSpMatrix.cu:85-98)
        uint32_t pos = transposed ? ((c & 7u) << 3) | (r & 7u) : ((r & 7u) << 3) | (c & 7u);
        keys[i] = ((uint64_t)(r >> 3) << (cbits + 6)) | ((uint64_t)(c >> 3) << 6) | pos;
        perm[i] = (uint32_t)i;
    }
};

// packed head flags: high word = first element of a block, low word = first element of a coordinate
struct HeadFlags {
    const uint64_t *sk;
    uint64_t n;
    __device__ uint64_t operator()(uint64_t i) const
    {
        if (i >= n) return 0;
        if (i == 0) return (1ull << 32) | 1ull;
        uint64_t a = sk[i - 1], b = sk[i];
        return ((uint64_t)((a >> 6) != (b >> 6)) << 32) | (uint64_t)(a != b);
    }
};

struct CountOut {
    uint64_t n;
    uint64_t *totals;  // [0] = packed (blocks<<32 | elements)
    __device__ void operator()(uint64_t i, uint64_t ex) const
    {
        if (i == n) totals[0] = ex;
    }
};

template <typename T>
struct ValCast;
template <>
struct ValCast<float> {
    static __device__ float from(double d) { return (float)d; }
};
template <>
struct ValCast<double> {
    static __device__ double from(double d) { return d; }
};
template <>
struct ValCast<_Float16> {
    static __device__ _Float16 from(double d)
    {
        uint16_t b = f64_to_f16_bits(d);  // single rounding, like (half)double (src/bmSpMatrix.cu:141)
        return __builtin_bit_cast(_Float16, b);
    }
};

template <typename T>
struct EmitBlocks {
    const uint64_t *sk;
    const uint32_t *perm;
    const double *vals_in;
    uint64_t n;
    int cbits;
    uint64_t *keys, *bmps, *offsets;
    T *values;
    __device__ void operator()(uint64_t i, uint64_t ex) const
    {
        uint32_t b = (uint32_t)(ex >> 32), el = (uint32_t)ex;
        if (i == n) {
            offsets[b] = el;  // terminal offset = nnz
            return;
        }
        uint64_t k = sk[i];
        bool ehead = i == 0 || sk[i - 1] != k;
        if (!ehead) return;
        // value of this coordinate: duplicates summed in input order in the matrix's own precision
        T acc = ValCast<T>::from(vals_in[perm[i]]);
        for (uint64_t j = i + 1; j < n && sk[j] == k; j++) acc = acc + ValCast<T>::from(vals_in[perm[j]]);
        values[el] = acc;
        bool bhead = i == 0 || (sk[i - 1] >> 6) != (k >> 6);
        if (!bhead) return;
        uint64_t blk = k >> 6;
        uint64_t bmp = 0;
        for (uint64_t j = i; j < n && (sk[j] >> 6) == blk; j++) bmp |= 1ull << (63 - (uint32_t)(sk[j] & 63u));
        uint32_t bcol = (uint32_t)(blk & ((1ull << cbits) - 1ull));
        uint32_t brow = (uint32_t)(blk >> cbits);
        keys[b] = key_make(brow, bcol);  // coord_to_key (src/bmSpMatrix.cu:76-83)
        bmps[b] = bmp;
        offsets[b] = el;
    }
};

template <typename T>
void emit_blocks(const uint64_t *sk, const uint32_t *perm, const double *d_vals, uint64_t n, int cbits, bmsp_matrix_s *m,
                 hipStream_t st)
{
    EmitBlocks<T> out{sk, perm, d_vals, n, cbits, m->keys, m->bmps, m->offsets, (T *)m->values};
    device_exclusive_scan<uint64_t>(HeadFlags{sk, n}, out, n + 1, st);
}

struct RowPtrSearch {
    const uint64_t *keys;
    uint32_t nb;
    uint32_t *rowptr;
    __device__ void operator()(uint64_t r) const
    {
        // first block whose block-row is >= r
        uint32_t lo = 0, hi = nb;
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if ((keys[mid] >> 32) < r) lo = mid + 1;
            else hi = mid;
        }
        rowptr[r] = lo;
    }
};

}  // namespace

void free_matrix(bmsp_matrix_s *m)
{
    if (!m) return;
    if (m->ownership == 1) {
        pool_free(m->keys); pool_free(m->bmps); pool_free(m->offsets); pool_free(m->values);
    }
    pool_free(m->rowptr);
    pool_free(m->spmv_chunks);
    pool_free(m->spmv_pos);
    pool_free(m->block_meta);
    pool_free(m->sym_recs);
    pool_free(m->col_index); pool_free(m->col_index_row); pool_free(m->col_mass);
    pool_free(m->dense_tiles);
    pool_free(m->lane_tiles);
    pool_free(m->csr_rowptr); pool_free(m->csr_ent);
    pool_free(m->sp_tasks); pool_free(m->sp_task_begin); pool_free(m->sp_c_of_wave);
    free_matrix(m->shard_view);
    delete m;
}

// Drops what the operators derived from the arrays and cached on the handle (bmsp_matrix_invalidate).  Value-derived: the dense tile
// copies of the block-MAC kernels.  Structure-derived: block-row pointer and row maxima, SpMV plan and position cache, block records,
// the sharded SpMV's panel view.  Everything is rebuilt lazily by the next operator call.
void invalidate_matrix(bmsp_matrix_s *m, int structure_changed)
{
    if (!m) return;
    BMSP_HIP(hipDeviceSynchronize());  // no kernel may still be reading what is about to go back to the pool
    pool_free(m->dense_tiles); m->dense_tiles = nullptr;
    pool_free(m->lane_tiles); m->lane_tiles = nullptr;
    pool_free(m->csr_rowptr); pool_free(m->csr_ent);
    m->csr_rowptr = nullptr; m->csr_ent = nullptr;
    m->values_finite = -1;
    if (!structure_changed) return;
    pool_free(m->rowptr); m->rowptr = nullptr; m->rowptr_rows = 0; m->max_row_blocks = -1;
    m->struct_hash = 0; m->sp_a_hash = 0; m->sp_b_hash = 0;
    m->rm_partner_uid = 0; m->rm_partner_blocks = 0; m->rm_partner_mode = 0; m->rm_partner_cw_hash = 0;
    m->uid = next_matrix_uid();  // what other matrices remembered about this one's old structure no longer applies
    pool_free(m->sp_tasks); pool_free(m->sp_task_begin); pool_free(m->sp_c_of_wave);
    m->sp_tasks = nullptr; m->sp_task_begin = nullptr; m->sp_c_of_wave = nullptr; m->sp_n_tasks = 0;
    pool_free(m->spmv_chunks); m->spmv_chunks = nullptr; m->spmv_num_chunks = 0; m->spmv_plan_long = 0; m->spmv_full_tiles = 0;
    pool_free(m->spmv_pos); m->spmv_pos = nullptr; m->spmv_tinfo = nullptr; m->spmv_eoff = nullptr; m->spmv_pos_base = 0; m->spmv_pos_count = 0; m->spmv_pos_tried = 0;
    pool_free(m->block_meta); m->block_meta = nullptr;
    pool_free(m->sym_recs); m->sym_recs = nullptr;
    pool_free(m->col_index); pool_free(m->col_index_row); m->col_index = nullptr; m->col_index_row = nullptr; m->col_index_tried = 0;
    pool_free(m->col_mass); m->col_mass = nullptr;
    free_matrix(m->shard_view); m->shard_view = nullptr; m->shard_world = 0; m->shard_rank = 0; m->shard_bounds.clear();
}

void ensure_rowptr(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->rowptr) return;
    int64_t nbr = m->num_block_rows();
    if (m->block_num >= (1ll << 32)) fail(BMSP_ERR_LIMIT, "more than 2^32 blocks");
    m->rowptr = (uint32_t *)pool_alloc(sizeof(uint32_t) * (size_t)(nbr + 1));
    m->rowptr_rows = nbr;
    device_for_each(RowPtrSearch{m->keys, (uint32_t)m->block_num, m->rowptr}, (uint64_t)nbr + 1, st);
}

namespace {
struct RowBlocksIn {
    const uint32_t *rowptr;
    __device__ uint64_t operator()(uint64_t r) const { return (uint64_t)(rowptr[r + 1] - rowptr[r]); }
};
}  // namespace

// most blocks in one block-row, once per matrix (one small reduction and a read-back the first time)
void ensure_row_stats(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->max_row_blocks >= 0) return;
    ensure_rowptr(m, st);
    const int64_t nbr = m->num_block_rows();
    if (nbr == 0) { m->max_row_blocks = 0; return; }
    DevBuf<unsigned long long> mx(1);
    BMSP_HIP(hipMemsetAsync(mx.p, 0, 8, st));
    device_max_sum(RowBlocksIn{m->rowptr}, (uint64_t)nbr, mx.p, (unsigned long long *)nullptr, st);
    m->max_row_blocks = (int64_t)read_back(mx.p, st);
}

namespace {
struct StructMix {
    const uint64_t *keys, *bmps;
    __device__ uint64_t operator()(uint64_t b) const
    {
        uint64_t z = keys[b] * 0x9E3779B97F4A7C15ull ^ bmps[b];
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
};
}  // namespace

// Fingerprint of a matrix's STRUCTURE (dimensions, layout, keys, bitmaps): one reduction the first time, cached on the handle, dropped by
// bmsp_matrix_invalidate(m, 1).  A product stamps C with its operands' fingerprints (spgemm); bmsp_spgemm_numeric compares them.
uint64_t ensure_struct_hash(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->struct_hash) return m->struct_hash;
    unsigned long long sum = 0;
    if (m->block_num) {
        DevBuf<unsigned long long> acc(1);
        BMSP_HIP(hipMemsetAsync(acc.p, 0, 8, st));
        device_max_sum(StructMix{m->keys, m->bmps}, (uint64_t)m->block_num, (unsigned long long *)nullptr, acc.p, st);
        sum = read_back(acc.p, st);
    }
    uint64_t h = sum ^ ((uint64_t)(uint32_t)m->num_rows << 32 | (uint64_t)(uint32_t)m->num_cols) * 0xD6E8FEB86659FD93ull ^ (uint64_t)m->block_num * 0xA24BAED4963EE407ull ^
                 (uint64_t)(m->transposed ? 0x5851F42D4C957F2Dull : 0);
    if (h == 0) h = 1;
    m->struct_hash = h;
    return h;
}

namespace {
struct PackBlockMeta {
    const uint64_t *bmps, *offsets;
    uint32_t *meta;
    __device__ void operator()(uint64_t b) const
    {
        const uint64_t bm = bmps[b];
        uint32_t *r = meta + 4 * b;
        r[0] = (uint32_t)bm; r[1] = (uint32_t)(bm >> 32); r[2] = (uint32_t)offsets[b]; r[3] = 0u;
    }
};
}  // namespace

void ensure_block_meta(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->block_meta) return;
    if ((uint64_t)m->values_extent() >= (1ull << 32)) fail(BMSP_ERR_LIMIT, "packed block records hold 32-bit value offsets");
    m->block_meta = (uint32_t *)pool_alloc(16 * (size_t)(m->block_num ? m->block_num : 1));
    if (m->block_num) device_for_each(PackBlockMeta{m->bmps, m->offsets, m->block_meta}, (uint64_t)m->block_num, st);
}

namespace {
struct PackSymRecs {
    const uint64_t *keys, *bmps;
    uint32_t *out;
    __device__ void operator()(uint64_t b) const
    {
        const uint64_t bm = bmps[b], rm = tile_transpose(bm);
        uint32_t *r = out + 4 * b;
        r[0] = (uint32_t)rm; r[1] = (uint32_t)(rm >> 32); r[2] = key_col(keys[b]); r[3] = tile_or_bytes(bm);
    }
};
// tile_product_rm(a, transpose(bt)) against tile_product_bmp(a, bt) on bitmaps of every density
__global__ __launch_bounds__(kThreads) void tile_product_selftest_kernel(uint32_t n, uint32_t *bad)
{
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n) return;
    auto mix = [](uint64_t z) {
        z += 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    uint64_t a = mix(2ull * i), b = mix(2ull * i + 1);
    for (uint32_t k = 0; k < i % 5; k++) a &= mix(a + k);
    for (uint32_t k = 0; k < (i / 5) % 5; k++) b &= mix(b + k);
    if (i % 97 == 0) a = ~0ull;
    if (i % 89 == 0) b = ~0ull;
    if (tile_product_rm(a, tile_transpose(b)) != tile_product_bmp(a, b)) atomicAdd(bad, 1u);
}
}  // namespace

// {row-major bitmap, block column, rows in use} of every tile of a right operand (stored column-major): built once per matrix, like the
// block records
void ensure_sym_recs(bmsp_matrix_s *m, hipStream_t st)
{
    if (m->sym_recs) return;
    m->sym_recs = (uint32_t *)pool_alloc(16 * (size_t)(m->block_num ? m->block_num : 1));
    if (m->block_num) device_for_each(PackSymRecs{m->keys, m->bmps, m->sym_recs}, (uint64_t)m->block_num, st);
}

namespace {
struct IdxEntriesIn {
    const uint32_t *rowptr;
    uint64_t rows;
    uint32_t per_row;
    __device__ uint64_t operator()(uint64_t k) const { return k < rows && rowptr[k + 1] - rowptr[k] > kIdxMinLen ? (uint64_t)per_row : 0ull; }
};
struct IdxRowOut {
    const uint32_t *rowptr;
    uint32_t *idx_row;
    uint64_t rows;
    uint64_t *total;
    __device__ void operator()(uint64_t k, uint64_t ex) const
    {
        if (k == rows) { *total = ex; return; }
        idx_row[k] = rowptr[k + 1] - rowptr[k] > kIdxMinLen ? (uint32_t)ex : ~0u;
    }
};
// one thread per (block-row, grid column): first tile of the block-row at or beyond the column
struct FillColIndex {
    const uint64_t *keys;
    const uint32_t *rowptr, *idx_row;
    uint32_t *idx;
    uint32_t per_row, gran;
    __device__ void operator()(uint64_t i) const
    {
        const uint32_t k = (uint32_t)(i / per_row), q = (uint32_t)(i % per_row);
        const uint32_t off = idx_row[k];
        if (off == ~0u) return;
        uint32_t lo = rowptr[k], hi = rowptr[k + 1];
        const uint64_t col = (uint64_t)q * gran;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if ((uint64_t)key_col(keys[mid]) < col) lo = mid + 1;
            else hi = mid;
        }
        idx[off + q] = lo;
    }
};
}  // namespace

// Column index of the block-rows longer than kIdxMinLen tiles, on a grid of `gran` block columns: built once per matrix (right operands of
// the column-window passes: every window of every block-row of A that meets a long block-row of B cuts it at two grid columns).  Left
// unbuilt (col_index_row stays null, the passes search) when it would exceed 256 MB.
void ensure_col_index(bmsp_matrix_s *m, uint32_t gran, hipStream_t st)
{
    if (m->col_index_tried && m->col_index_gran == gran) return;
    pool_free(m->col_index); pool_free(m->col_index_row);
    m->col_index = nullptr; m->col_index_row = nullptr;
    m->col_index_tried = 1; m->col_index_gran = gran;
    ensure_rowptr(m, st);
    const uint64_t rows = (uint64_t)m->num_block_rows();
    const uint64_t per_row = ((uint64_t)m->num_block_cols() + gran - 1) / gran + 1;
    if (rows == 0 || per_row >= (1ull << 20) || rows * per_row >= (1ull << 40)) return;
    DevBuf<uint32_t> idx_row(rows + 1);
    HostScalar<uint64_t> total_h;
    device_exclusive_scan<uint64_t>(IdxEntriesIn{m->rowptr, rows, (uint32_t)per_row}, IdxRowOut{m->rowptr, idx_row.p, rows, total_h.dev()}, rows + 1, st);
    const uint64_t total = total_h.wait(st);
    if (total * 4 > (256ull << 20)) return;
    m->col_index = (uint32_t *)pool_alloc(4 * (size_t)(total ? total : 1));
    m->col_index_row = idx_row.take();
    if (total) device_for_each(FillColIndex{m->keys, m->rowptr, m->col_index_row, m->col_index, (uint32_t)per_row, gran}, rows * per_row, st);
}

namespace {
// tiles per granule of `gran` block columns: a histogram per workgroup in LDS first (power-law operands send most tiles to the first
// granules: one global atomic per tile took 1.1 ms on R-MAT 2^16), then one atomic per non-empty bin and workgroup
constexpr uint32_t kMassBins = 8192;
__global__ __launch_bounds__(kThreads) void mass_count_kernel(const uint64_t *__restrict__ keys, uint64_t n, uint32_t gran, uint32_t G, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t h[kMassBins];
    const bool lds = G <= kMassBins;
    if (lds) {
        for (uint32_t b = threadIdx.x; b < G; b += kThreads) h[b] = 0u;
        __syncthreads();
    }
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        const uint32_t b = key_col(keys[i]) / gran;
        if (lds) atomicAdd(&h[b], 1u);
        else atomicAdd(&hist[b], 1u);
    }
    if (lds) {
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < G; b += kThreads)
            if (h[b]) atomicAdd(&hist[b], h[b]);
    }
}
}  // namespace

// prefix sums of the tiles per granule of `gran` block columns (G + 1 entries): built once per matrix
void ensure_col_mass(bmsp_matrix_s *m, uint32_t gran, hipStream_t st)
{
    if (m->col_mass) return;
    const uint64_t G = ((uint64_t)m->num_block_cols() + gran - 1) / gran;
    DevBuf<uint32_t> hist(G + 1);
    BMSP_HIP(hipMemsetAsync(hist.p, 0, 4 * (G + 1), st));
    if (m->block_num) {
        const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)m->block_num + kThreads - 1) / kThreads, 1024);
        hipLaunchKernelGGL(mass_count_kernel, dim3(grid), dim3(kThreads), 0, st, m->keys, (uint64_t)m->block_num, gran, (uint32_t)G, hist.p);
        BMSP_CHECK_LAUNCH();
    }
    m->col_mass = (uint32_t *)pool_alloc(4 * (size_t)(G + 1));
    device_exclusive_scan<uint32_t>(PtrIn<uint32_t>{hist.p}, PtrOut<uint32_t>{m->col_mass}, G + 1, st);
}

int tile_product_selftest(hipStream_t st)
{
    DevBuf<uint32_t> bad(1);
    BMSP_HIP(hipMemsetAsync(bad.p, 0, 4, st));
    const uint32_t n = 1u << 20;
    hipLaunchKernelGGL(tile_product_selftest_kernel, dim3(n / kThreads), dim3(kThreads), 0, st, n, bad.p);
    BMSP_CHECK_LAUNCH();
    return (int)read_back(bad.p, st);
}

void prepare_spgemm_operand(bmsp_matrix_s *m, hipStream_t st)
{
    if ((uint64_t)m->values_extent() >= (1ull << 32) || m->block_num >= (1ll << 28)) return;  // pointer-based block-MAC: no records
    ensure_block_meta(m, st);
    // K = 32 MFMA block-MAC: A (normal layout) is always read from its dense copy, B from the dense copy or from the records
    if (m->dtype == BMSP_F16 && m->block_num < (1ll << 25) && (!m->transposed || mac_mfma32_b_dense(m))) ensure_dense_tiles(m, st);
    // V15 block-MAC (tc_version 5): fp32 operands with tiles at least a quarter full are staged from a dense copy too (256 B per block)
    if (m->dtype == BMSP_F32 && m->nnz >= 16 * m->block_num && (uint64_t)m->block_num * 256 <= (4ull << 30)) ensure_dense_tiles(m, st);
    // fp32 operands of nearly empty tiles: the row-sparse block-MAC's CSR copy (8 B per value) -- what such a matrix is multiplied from;
    // otherwise the fp32 strip block-MAC's (and the opt-in fp32 MFMA task-list kernel's) tiles in MFMA lane order (256 B per block).
    // (fp16 operands take the row-sparse kernel only under tc_version 5: their copy is made by the first such product)
    const char *rse = getenv("BMSP_MAC_ROWSPARSE");
    const bool sparse_tiles = m->nnz <= 16 * m->block_num && (uint64_t)m->nnz < (1ull << 29) && !m->view_values_end && m->ownership != 2 && !(rse && rse[0] == '0');
    if (m->dtype == BMSP_F32 && sparse_tiles) ensure_csr32(m, st);
    else if (m->dtype == BMSP_F32 && m->block_num < (1ll << 24) && mac_f32_mfma_usable(st)) ensure_lane_tiles(m, st);
    // what decides whether a product takes the row-merge path: block-row pointer and maxima, "every stored value is finite"
    ensure_row_stats(m, st);
    if (m->ownership != 2 || !m->view_block_begin) ensure_struct_hash(m, st);
    if (m->dtype != BMSP_F64) ensure_finite_flag(m, st);
}

bmsp_matrix_s *build_from_device_coo(int num_rows, int num_cols, int64_t nnz, const int *d_rows, const int *d_cols,
                                     const double *d_vals, int transposed, bmsp_dtype dtype, hipStream_t st)
{
    if (num_rows < 0 || num_cols < 0 || nnz < 0) fail(BMSP_ERR_INVALID, "negative dimension");
    if (nnz >= (1ll << 32)) fail(BMSP_ERR_LIMIT, "nnz %lld exceeds the 32-bit element range", (long long)nnz);
    std::unique_ptr<bmsp_matrix_s, void (*)(bmsp_matrix_s *)> m(new bmsp_matrix_s(), free_matrix);
    m->num_rows = num_rows; m->num_cols = num_cols; m->dtype = dtype; m->transposed = transposed ? 1 : 0;
    uint64_t n = (uint64_t)nnz;
    int rbits = ceil_log2_u64((uint64_t)m->num_block_rows());
    int cbits = ceil_log2_u64((uint64_t)m->num_block_cols());

    DevBuf<uint64_t> k0(n), k1(n);
    DevBuf<uint32_t> p0(n), p1(n);
    device_for_each(MakeSortKey{d_rows, d_cols, k0.p, p0.p, cbits, m->transposed}, n, st);
    PingPong<uint64_t> kk{k0.p, k1.p};
    PingPong<uint32_t> pp{p0.p, p1.p};
    device_radix_sort_pairs<uint32_t>(kk, pp, n, 0, rbits + cbits + 6, st);

    DevBuf<uint64_t> totals(1);
    device_exclusive_scan<uint64_t>(HeadFlags{kk.cur, n}, CountOut{n, totals.p}, n + 1, st);
    uint64_t packed = n ? read_back(totals.p, st) : 0;
    m->block_num = (int64_t)(packed >> 32);
    m->nnz = (int64_t)(packed & 0xffffffffull);
    size_t nb = (size_t)m->block_num;
    m->keys = (uint64_t *)pool_alloc(8 * (nb ? nb : 1));
    m->bmps = (uint64_t *)pool_alloc(8 * (nb ? nb : 1));
    m->offsets = (uint64_t *)pool_alloc(8 * (nb + 1));
    m->values = pool_alloc(dtype_size(dtype) * (size_t)(m->nnz ? m->nnz : 1));
    if (n == 0) {
        BMSP_HIP(hipMemsetAsync(m->offsets, 0, 8, st));
    } else if (dtype == BMSP_F32) emit_blocks<float>(kk.cur, pp.cur, d_vals, n, cbits, m.get(), st);
    else if (dtype == BMSP_F16) emit_blocks<_Float16>(kk.cur, pp.cur, d_vals, n, cbits, m.get(), st);
    else emit_blocks<double>(kk.cur, pp.cur, d_vals, n, cbits, m.get(), st);
    ensure_rowptr(m.get(), st);
    BMSP_HIP(hipStreamSynchronize(st));  // temporaries go back to the pool below
    return m.release();
}

// ------------------------------------------------------------------------------------------------
// bmSparse -> COO sorted by (row, col)   (generate_coo, src/bmSpMatrix.cu:320-363)
// ------------------------------------------------------------------------------------------------
namespace {
template <typename T>
struct ExpandBlocks {
    const uint64_t *keys, *bmps, *offsets;
    const T *values;
    int transposed;
    uint64_t *rc;   // (row << 32) | col per stored value
    double *vals;
    __device__ void operator()(uint64_t b) const
    {
        // a row-panel view keeps offsets absolute into its parent's value array: values are read at the absolute offset, the
        // outputs (sized by the view's own nnz) are written relative to the view's first value
        const uint64_t base = offsets[0];
        uint64_t bmp = bmps[b], off = offsets[b];
        uint32_t brow = key_row(keys[b]), bcol = key_col(keys[b]);
        uint32_t k = 0;
        while (bmp) {
            int p = __clzll((long long)bmp);
            bmp &= ~(1ull << (63 - p));
            uint32_t hi = (uint32_t)p >> 3, lo = (uint32_t)p & 7u;
            uint32_t r = brow * 8 + (transposed ? lo : hi), c = bcol * 8 + (transposed ? hi : lo);
            rc[off - base + k] = ((uint64_t)r << 32) | c;
            vals[off - base + k] = (double)values[off + k];
            k++;
        }
    }
};
struct GatherD {
    const double *in;
    const uint32_t *perm;
    double *out;
    __device__ void operator()(uint64_t i) const { out[i] = in[perm[i]]; }
};
struct Iota32 {
    uint32_t *p;
    __device__ void operator()(uint64_t i) const { p[i] = (uint32_t)i; }
};
}  // namespace

// every stored value as ((row << 32) | col, value), sorted by (row, col); both outputs hold nnz entries
void matrix_to_coo_device(bmsp_matrix_s *m, uint64_t *d_rc, double *d_vals, hipStream_t st)
{
    uint64_t n = (uint64_t)m->nnz;
    if (n == 0) return;
    DevBuf<uint64_t> k0(n), k1(n);
    DevBuf<uint32_t> p0(n), p1(n);
    DevBuf<double> v0(n);
    uint64_t nb = (uint64_t)m->block_num;
    if (m->dtype == BMSP_F32)
        device_for_each(ExpandBlocks<float>{m->keys, m->bmps, m->offsets, (const float *)m->values, m->transposed, k0.p, v0.p}, nb, st);
    else if (m->dtype == BMSP_F16)
        device_for_each(ExpandBlocks<_Float16>{m->keys, m->bmps, m->offsets, (const _Float16 *)m->values, m->transposed, k0.p, v0.p}, nb, st);
    else
        device_for_each(ExpandBlocks<double>{m->keys, m->bmps, m->offsets, (const double *)m->values, m->transposed, k0.p, v0.p}, nb, st);
    device_for_each(Iota32{p0.p}, n, st);
    PingPong<uint64_t> kk{k0.p, k1.p};
    PingPong<uint32_t> pp{p0.p, p1.p};
    int cb = ceil_log2_u64((uint64_t)m->num_cols), rb = ceil_log2_u64((uint64_t)m->num_rows);
    // sort by column bits, then by row bits (stable LSD): row-major order
    device_radix_sort_pairs<uint32_t>(kk, pp, n, 0, cb, st);
    device_radix_sort_pairs<uint32_t>(kk, pp, n, 32, 32 + rb, st);
    device_for_each(GatherD{v0.p, pp.cur, d_vals}, n, st);
    BMSP_HIP(hipMemcpyAsync(d_rc, kk.cur, 8 * n, hipMemcpyDeviceToDevice, st));
    BMSP_HIP(hipStreamSynchronize(st));
}

void matrix_to_coo_host(bmsp_matrix_s *m, int *rows, int *cols, double *vals, hipStream_t st)
{
    uint64_t n = (uint64_t)m->nnz;
    if (n == 0) return;
    DevBuf<uint64_t> rc(n);
    DevBuf<double> v(n);
    matrix_to_coo_device(m, rc.p, v.p, st);
    std::vector<uint64_t> hk(n);
    BMSP_HIP(hipStreamSynchronize(st));
    copy_d2h_staged(hk.data(), rc.p, 8 * n);
    copy_d2h_staged(vals, v.p, 8 * n);
    for (uint64_t i = 0; i < n; i++) {
        rows[i] = (int)(hk[i] >> 32);
        cols[i] = (int)(hk[i] & 0xffffffffull);
    }
}

namespace {
struct SplitRowCol {
    const uint64_t *rc;
    int *rows, *cols;
    __device__ void operator()(uint64_t i) const
    {
        if (rows) rows[i] = (int)(rc[i] >> 32);
        cols[i] = (int)(rc[i] & 0xffffffffull);
    }
};
// row_offsets[r] = index of the first entry whose row is >= r, from the (row, col)-sorted list
struct RowOffsetsFromSorted {
    const uint64_t *rc;
    uint64_t n;
    uint32_t num_rows;
    int *row_offsets;
    __device__ void operator()(uint64_t i) const
    {
        const uint32_t hi = i < n ? (uint32_t)(rc[i] >> 32) : num_rows;
        const uint32_t lo = i ? (uint32_t)(rc[i - 1] >> 32) + 1u : 0u;
        for (uint32_t r = lo; r <= hi; r++) row_offsets[r] = (int)i;
    }
};
struct RowOfEntry {
    const int *row_offsets;
    uint32_t num_rows;
    int *rows;
    __device__ void operator()(uint64_t i) const
    {
        uint32_t lo = 0, hi = num_rows;  // last r with row_offsets[r] <= i
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)row_offsets[mid] <= i) lo = mid; else hi = mid;
        }
        rows[i] = (int)lo;
    }
};
}  // namespace

namespace {
struct PackRowCol {
    const int *rows, *cols;
    uint64_t *keys;
    uint32_t *idx;
    __device__ void operator()(uint64_t i) const
    {
        keys[i] = ((uint64_t)(uint32_t)rows[i] << 32) | (uint32_t)cols[i];
        idx[i] = (uint32_t)i;
    }
};
// bmSpMatrix<T>::compare (src/bmSpMatrix.cu:381-432), one thread per stored value of m: find the comparand entry with the same
// coordinates (first of equal ones, as the stable host sort picks), add the relative error, count the entries m holds alone
__global__ __launch_bounds__(kThreads) void compare_kernel(const uint64_t *__restrict__ m_rc, const double *__restrict__ m_vals, uint64_t n,
                                                           const uint64_t *__restrict__ c_keys, const uint32_t *__restrict__ c_idx,
                                                           const double *__restrict__ c_vals, uint64_t nc, double *__restrict__ err_sum,
                                                           unsigned long long *__restrict__ missing)
{
    __shared__ double s_err[4];
    __shared__ unsigned long long s_miss[4];
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    double term = 0.0;
    unsigned long long miss = 0;
    if (i < n) {
        const uint64_t key = m_rc[i];
        uint64_t lo = 0, hi = nc;
        while (lo < hi) {
            const uint64_t mid = lo + ((hi - lo) >> 1);
            if (c_keys[mid] < key) lo = mid + 1; else hi = mid;
        }
        if (lo >= nc || c_keys[lo] != key) {
            miss = 1;
        } else {
            const double eps = 1e-8;  // :403
            const double cv = c_vals[c_idx[lo]], mv = m_vals[i];
            const double e = fabs(cv) < eps ? 0.0 : cv, r = fabs(mv) < eps ? 0.0 : mv;
            term = fabs(e - r) / fmax(fabs(e), eps);  // :418
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        term += __shfl_xor(term, d, kWave);
        miss += __shfl_xor(miss, d, kWave);
    }
    if (lane_id() == 0) { s_err[wave_id()] = term; s_miss[wave_id()] = miss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(err_sum, s_err[0] + s_err[1] + s_err[2] + s_err[3]);
        atomicAdd(missing, s_miss[0] + s_miss[1] + s_miss[2] + s_miss[3]);
    }
}
}  // namespace

void matrix_compare_device(bmsp_matrix_s *m, int64_t nnz, const int *d_rows, const int *d_cols, const double *d_vals, double *mean_rel_err,
                           int64_t *missing, hipStream_t st)
{
    const uint64_t n = (uint64_t)m->nnz, nc = (uint64_t)nnz;
    if (nc >= (1ull << 32)) fail(BMSP_ERR_LIMIT, "comparand with 2^32 or more entries");
    *mean_rel_err = 0.0;
    if (missing) *missing = 0;
    if (n == 0) return;
    DevBuf<uint64_t> rc(n), k0(nc), k1(nc);
    DevBuf<double> mv(n);
    DevBuf<uint32_t> i0(nc), i1(nc);
    matrix_to_coo_device(m, rc.p, mv.p, st);
    PingPong<uint64_t> kk{k0.p, k1.p};
    PingPong<uint32_t> ii{i0.p, i1.p};
    if (nc) {
        device_for_each(PackRowCol{d_rows, d_cols, k0.p, i0.p}, nc, st);
        device_radix_sort_pairs<uint32_t>(kk, ii, nc, 0, 64, st);
    }
    DevBuf<double> err(1);
    DevBuf<unsigned long long> miss(1);
    BMSP_HIP(hipMemsetAsync(err.p, 0, 8, st));
    BMSP_HIP(hipMemsetAsync(miss.p, 0, 8, st));
    hipLaunchKernelGGL(compare_kernel, grid_for(n), dim3(kThreads), 0, st, rc.p, mv.p, n, kk.cur, ii.cur, d_vals, nc, err.p, miss.p);
    BMSP_CHECK_LAUNCH();
    const double sum = read_back(err.p, st);
    const unsigned long long ms = read_back(miss.p, st);
    *mean_rel_err = sum / (double)n;  // "Final:" (:429)
    if (missing) *missing = (int64_t)ms;
}

void matrix_to_coo_device_split(bmsp_matrix_s *m, int *d_rows, int *d_cols, double *d_vals, hipStream_t st)
{
    const uint64_t n = (uint64_t)m->nnz;
    if (!n) return;
    DevBuf<uint64_t> rc(n);
    matrix_to_coo_device(m, rc.p, d_vals, st);
    device_for_each(SplitRowCol{rc.p, d_rows, d_cols}, n, st);
    BMSP_HIP(hipStreamSynchronize(st));
}

void matrix_to_csr_device(bmsp_matrix_s *m, int *d_row_offsets, int *d_cols, double *d_vals, hipStream_t st)
{
    const uint64_t n = (uint64_t)m->nnz;
    DevBuf<uint64_t> rc(n);
    matrix_to_coo_device(m, rc.p, d_vals, st);
    device_for_each(RowOffsetsFromSorted{rc.p, n, (uint32_t)m->num_rows, d_row_offsets}, n + 1, st);
    if (n) device_for_each(SplitRowCol{rc.p, nullptr, d_cols}, n, st);
    BMSP_HIP(hipStreamSynchronize(st));
}

bmsp_matrix_s *build_from_device_csr(int num_rows, int num_cols, int64_t nnz, const int *d_row_offsets, const int *d_cols, const double *d_vals,
                                     int transposed, bmsp_dtype dtype, hipStream_t st)
{
    DevBuf<int> rows((size_t)nnz);
    if (nnz) device_for_each(RowOfEntry{d_row_offsets, (uint32_t)num_rows, rows.p}, (uint64_t)nnz, st);
    return build_from_device_coo(num_rows, num_cols, nnz, rows.p, d_cols, d_vals, transposed, dtype, st);
}

}  // namespace bmsp

BMSP_DEFINE_WARM(builder)
