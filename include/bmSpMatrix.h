/*
 * bmSpMatrix.h -- the reference's container and operator names on top of the MI355X C ABI (bmsp.h).
 *
 * Keeps the public surface of the reference header include/bmSpMatrix.h:20-40 (class bmSpMatrix<T> with public
 * keys / bmps / offsets / values / num_rows / num_cols / nnz / block_num, the three constructors, compare, print,
 * generate_coo) and of the two operator templates bmSparse_SpMV (src/bmSparse_SPMV.cu:191-192) and bmSparse_mult
 * (src/bmSparse_SPGEMM.cu:827-828), so the reference's mains compile against it after swapping
 * thrust::device_vector / cusp::coo_matrix for the two small types below.  Everything is a call into libbmsp.so;
 * this header contains no compute.
 */
#ifndef BMSPMATRIX_H_
#define BMSPMATRIX_H_

#include "bmsp.h"
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#define BLOCK_WIDTH 8
#define BLOCK_HEIGHT 8
#define BMSP_BLOCK_SIZE (BLOCK_WIDTH * BLOCK_HEIGHT)

#ifndef BMSP_NO_HALF_TYPE
/* 16-bit storage type for bmSpMatrix<half> (the reference uses CUDA's __half, src/bmSpMatrix.cu:20,436). */
struct half {
    uint16_t bits;
};
#endif

namespace bmsp {

inline void check(int status)
{
    if (status != BMSP_OK) throw std::runtime_error(std::string("bmsp: ") + bmsp_last_error());
}

template <class T> struct dtype_of;
template <> struct dtype_of<float> { static const bmsp_dtype value = BMSP_F32; };
template <> struct dtype_of<double> { static const bmsp_dtype value = BMSP_F64; };
#ifndef BMSP_NO_HALF_TYPE
template <> struct dtype_of<half> { static const bmsp_dtype value = BMSP_F16; };
#endif

/* The subset of thrust::device_vector the reference's code uses: size/data/begin/end/swap/clear/shrink_to_fit,
 * construction from a host vector, copy back to the host.  Owns pool memory unless it is a view. */
template <class T> class device_vector {
    T *p_ = nullptr;
    size_t n_ = 0;
    bool own_ = true;

public:
    typedef T value_type;
    device_vector() {}
    explicit device_vector(size_t n) { resize(n); }
    explicit device_vector(const std::vector<T> &h)
    {
        resize(h.size());
        if (n_) check(bmsp_memcpy_h2d(p_, h.data(), n_ * sizeof(T)));
    }
    device_vector(const device_vector &) = delete;
    device_vector &operator=(const device_vector &) = delete;
    device_vector(device_vector &&o) noexcept { swap(o); }
    device_vector &operator=(device_vector &&o) noexcept
    {
        if (this != &o) { clear(); swap(o); }
        return *this;
    }
    ~device_vector() { clear(); }
    static device_vector view(T *p, size_t n)
    {
        device_vector v;
        v.p_ = p; v.n_ = n; v.own_ = false;
        return v;
    }
    void resize(size_t n)
    {
        clear();
        void *q = nullptr;
        check(bmsp_malloc(&q, (n ? n : 1) * sizeof(T)));
        p_ = static_cast<T *>(q); n_ = n; own_ = true;
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    T *begin() { return p_; }
    T *end() { return p_ + n_; }
    void swap(device_vector &o)
    {
        std::swap(p_, o.p_); std::swap(n_, o.n_); std::swap(own_, o.own_);
    }
    void clear()
    {
        if (p_ && own_) bmsp_free(p_);
        p_ = nullptr; n_ = 0; own_ = true;
    }
    void shrink_to_fit() {}
    T *release()
    {
        T *q = p_;
        p_ = nullptr; n_ = 0;
        return q;
    }
    std::vector<T> to_host() const
    {
        std::vector<T> h(n_);
        if (n_) check(bmsp_memcpy_d2h(h.data(), p_, n_ * sizeof(T)));
        return h;
    }
};

/* host COO with the member names of cusp::coo_matrix (cusp/coo_matrix.h:155-163, detail/matrix_base.h:38-40) */
template <class T> struct coo_matrix {
    size_t num_rows = 0, num_cols = 0, num_entries = 0;
    std::vector<int> row_indices, column_indices;
    std::vector<T> values;
};

}  // namespace bmsp

template <class valueType> class bmSpMatrix {
private:
    bmsp_matrix_t h_ = nullptr;
    bmsp::coo_matrix<double> coo;  // filled by generate_coo()
    void refresh()
    {
        int64_t nz = 0, nb = 0;
        bmsp::check(bmsp_matrix_info(h_, &num_rows, &num_cols, &nz, &nb, nullptr, nullptr));
        nnz = (int)nz; block_num = (int)nb;
        uint64_t *k, *b, *o; void *v;
        bmsp::check(bmsp_matrix_arrays(h_, &k, &b, &o, &v));
        keys = bmsp::device_vector<uint64_t>::view(k, (size_t)nb);
        bmps = bmsp::device_vector<uint64_t>::view(b, (size_t)nb);
        offsets = bmsp::device_vector<uint64_t>::view(o, (size_t)nb + 1);
        values = bmsp::device_vector<valueType>::view(static_cast<valueType *>(v), (size_t)nz);
    }

public:
    bmsp::device_vector<uint64_t> keys;
    bmsp::device_vector<uint64_t> bmps;
    bmsp::device_vector<uint64_t> offsets;
    bmsp::device_vector<valueType> values;
    int num_rows = 0, num_cols = 0, nnz = 0, block_num = 0;

    bmSpMatrix() {}
    /* src/bmSpMatrix.cu:111-219 */
    bmSpMatrix(std::string path, bool transpose)
    {
        bmsp::check(bmsp_matrix_from_mtx(path.c_str(), transpose ? 1 : 0, bmsp::dtype_of<valueType>::value, &h_));
        refresh();
    }
    /* src/bmSpMatrix.cu:30-43: adopts the four device vectors (the reference swaps them in) */
    bmSpMatrix(int num_rows_, int num_cols_, int block_num_, bmsp::device_vector<uint64_t> &keys_,
               bmsp::device_vector<uint64_t> &bmps_, bmsp::device_vector<uint64_t> &offsets_,
               bmsp::device_vector<valueType> &values_)
    {
        size_t nz = values_.size();
        bmsp::check(bmsp_matrix_from_arrays(num_rows_, num_cols_, block_num_, (int64_t)nz, keys_.data(), bmps_.data(), offsets_.data(),
                                            values_.data(), bmsp::dtype_of<valueType>::value, 0, 1, &h_));
        keys_.release(); bmps_.release(); offsets_.release(); values_.release();
        refresh();
    }
    bmSpMatrix(const bmSpMatrix &) = delete;
    bmSpMatrix &operator=(const bmSpMatrix &) = delete;
    bmSpMatrix(bmSpMatrix &&o) noexcept { *this = std::move(o); }
    bmSpMatrix &operator=(bmSpMatrix &&o) noexcept
    {
        if (this != &o) {
            reset(nullptr);
            h_ = o.h_; o.h_ = nullptr;
            if (h_) refresh();
            o.reset(nullptr);
        }
        return *this;
    }
    ~bmSpMatrix() { reset(nullptr); }

    /* takes ownership of a C-ABI handle (used by bmSparse_mult for C) */
    void reset(bmsp_matrix_t h)
    {
        keys.clear(); bmps.clear(); offsets.clear(); values.clear();
        if (h_) bmsp_matrix_free(h_);
        h_ = h;
        num_rows = num_cols = nnz = block_num = 0;
        coo = bmsp::coo_matrix<double>();
        if (h_) refresh();
    }
    bmsp_matrix_t handle() const { return h_; }

    /* src/bmSpMatrix.cu:320-363 */
    void generate_coo()
    {
        coo.num_rows = (size_t)num_rows; coo.num_cols = (size_t)num_cols; coo.num_entries = (size_t)nnz;
        coo.row_indices.resize((size_t)nnz); coo.column_indices.resize((size_t)nnz); coo.values.resize((size_t)nnz);
        if (h_) bmsp::check(bmsp_matrix_to_coo_host(h_, coo.row_indices.data(), coo.column_indices.data(), coo.values.data()));
    }
    const bmsp::coo_matrix<double> &host_coo()
    {
        if (coo.num_entries == 0) generate_coo();
        return coo;
    }
    /* src/bmSpMatrix.cu:381-432: prints "Final: <mean relative error>" and returns true */
    template <class T> bool compare(const bmsp::coo_matrix<T> &other)
    {
        std::vector<double> v(other.values.begin(), other.values.end());
        double err = 0; int64_t missing = 0;
        bmsp::check(bmsp_matrix_compare(h_, (int64_t)other.num_entries, other.row_indices.data(), other.column_indices.data(), v.data(),
                                        &err, &missing));
        std::printf("Final: %g", err);
        return true;
    }
    void print()
    {
        const bmsp::coo_matrix<double> &c = host_coo();
        std::printf("sparse matrix <%d, %d> with %d entries\n", num_rows, num_cols, nnz);
        for (size_t i = 0; i < c.num_entries; i++) std::printf(" %d %d %g\n", c.row_indices[i], c.column_indices[i], c.values[i]);
    }
};

/* src/bmSparse_SPMV.cu:191-230.  v, u are device pointers; synchronous like the reference (cudaDeviceSynchronize, :223). */
template <class ValueIn, class ValueOut> inline void bmSparse_SpMV(bmSpMatrix<ValueIn> &A, ValueIn *v, ValueOut *u, bool batched)
{
    static_assert(std::is_same<ValueOut, float>::value || std::is_same<ValueOut, double>::value, "u is float (double for double input)");
    bmsp::check(bmsp_spmv(A.handle(), v, u, batched ? BMSP_SPMV_BATCHED : BMSP_SPMV_DEFAULT, nullptr));
    bmsp::check(bmsp_synchronize());
}

/* Multi-vector form: U = A * V for k vectors, V row-major num_cols x k, U row-major num_rows x k. */
template <class ValueIn, class ValueOut> inline void bmSparse_SpMM(bmSpMatrix<ValueIn> &A, ValueIn *V, ValueOut *U, int k)
{
    static_assert(std::is_same<ValueOut, float>::value || std::is_same<ValueOut, double>::value, "U is float (double for double input)");
    bmsp::check(bmsp_spmm(A.handle(), V, k, U, k, k, nullptr));
    bmsp::check(bmsp_synchronize());
}

/* src/bmSparse_SPGEMM.cu:827-1223.  `mode` is the reference's `segmented` flag (declared bool there). */
template <class valueIn, class valueOut>
inline void bmSparse_mult(bmSpMatrix<valueIn> &A, bmSpMatrix<valueIn> &B, bmSpMatrix<valueOut> &C, bool mode, bool VERBOSE, long tc_version,
                          bmsp_spgemm_stats *stats = nullptr)
{
    bmsp_matrix_t c = nullptr;
    bmsp_spgemm_stats st;
    bmsp::check(bmsp_spgemm(A.handle(), B.handle(), &c, mode ? BMSP_SORT_SEGMENTED : BMSP_SORT_AUTO, (int)tc_version, VERBOSE ? 1 : 0, nullptr, &st));
    C.reset(c);
    if (stats) *stats = st;
    std::printf("Toda F: %lld \xce\xbcs \n", (long long)(st.t_us[0] + 0.5)); /* :1220 */
}

/* The two halves of bmSparse_mult for repeated products on one sparsity pattern (bmsp_spgemm_symbolic / bmsp_spgemm_numeric; the reference
 * runs both halves in every call, src/bmSparse_SPGEMM.cu:849-1158): _symbolic leaves C with the product's structure and zero values,
 * _numeric overwrites the values of a C of that structure with those of A x B. */
template <class valueIn, class valueOut>
inline void bmSparse_mult_symbolic(bmSpMatrix<valueIn> &A, bmSpMatrix<valueIn> &B, bmSpMatrix<valueOut> &C, bool mode, long tc_version,
                                   bmsp_spgemm_stats *stats = nullptr)
{
    bmsp_matrix_t c = nullptr;
    bmsp::check(bmsp_spgemm_symbolic(A.handle(), B.handle(), &c, mode ? BMSP_SORT_SEGMENTED : BMSP_SORT_AUTO, (int)tc_version, nullptr, stats));
    C.reset(c);
}
template <class valueIn, class valueOut>
inline void bmSparse_mult_numeric(bmSpMatrix<valueIn> &A, bmSpMatrix<valueIn> &B, bmSpMatrix<valueOut> &C, long tc_version, bmsp_spgemm_stats *stats = nullptr)
{
    bmsp::check(bmsp_spgemm_numeric(A.handle(), B.handle(), C.handle(), (int)tc_version, nullptr, stats));
}

/* The same product sharded over one process per GPU (bmsp_spgemm_sharded): every rank passes the same A and B, multiplies
 * its block-row panel of A and returns the whole C. */
template <class valueIn, class valueOut>
inline void bmSparse_mult_sharded(bmsp_comm_t comm, bmSpMatrix<valueIn> &A, bmSpMatrix<valueIn> &B, bmSpMatrix<valueOut> &C, bool mode, bool VERBOSE,
                                  long tc_version, bmsp_spgemm_stats *stats = nullptr, bmsp_shard_stats *shard = nullptr)
{
    bmsp_matrix_t c = nullptr;
    bmsp_spgemm_stats st;
    bmsp::check(bmsp_spgemm_sharded(comm, A.handle(), B.handle(), &c, mode ? BMSP_SORT_SEGMENTED : BMSP_SORT_AUTO, (int)tc_version, VERBOSE ? 1 : 0,
                                    nullptr, &st, shard));
    C.reset(c);
    if (stats) *stats = st;
    std::printf("Toda F: %lld \xce\xbcs \n", (long long)(st.t_us[0] + 0.5));
}

#endif /* BMSPMATRIX_H_ */
