/*
 * CSRMatrix.h -- host CSR container with the reference's class name and members
 * (reference include/CSRMatrix.h:13-21: CSRMatrix(std::string), CSRMatrix(cusp::csr_matrix*), multiply).
 * The reference only declares this class; its backing store is cusp::csr_matrix<int,float,host_memory> and its
 * multiply is cusp::multiply.  Here the arrays stay on the host and multiply runs on the GPU through the bmSparse
 * SpGEMM (bmsp_csr_multiply): numeric zeros are dropped like cusp's host SpGEMM, columns come back ascending.
 */
#ifndef CSRMATRIX_H_
#define CSRMATRIX_H_

#include "bmsp.h"
#include <stdexcept>
#include <string>
#include <vector>

/* shape of cusp::csr_matrix (cusp/csr_matrix.h:150-158, detail/matrix_base.h:38-40) for the pointer constructor */
template <class IndexType, class ValueType> struct bmsp_host_csr {
    size_t num_rows = 0, num_cols = 0, num_entries = 0;
    std::vector<IndexType> row_offsets, column_indices;
    std::vector<ValueType> values;
};

class CSRMatrix {
public:
    explicit CSRMatrix(std::string path) { check(bmsp_csr_from_mtx(path.c_str(), &h_)); }
    explicit CSRMatrix(bmsp_host_csr<int, float> *m)
    {
        check(bmsp_csr_from_arrays((int)m->num_rows, (int)m->num_cols, (int64_t)m->num_entries, m->row_offsets.data(),
                                   m->column_indices.data(), m->values.data(), &h_));
    }
    CSRMatrix(const CSRMatrix &o)
    {
        int nr, nc; int64_t nnz; const int *ro, *ci; const float *v;
        check(bmsp_csr_info(o.h_, &nr, &nc, &nnz));
        check(bmsp_csr_arrays(o.h_, &ro, &ci, &v));
        check(bmsp_csr_from_arrays(nr, nc, nnz, ro, ci, v, &h_));
    }
    CSRMatrix &operator=(const CSRMatrix &) = delete;
    ~CSRMatrix() { if (h_) bmsp_csr_free(h_); }

    CSRMatrix multiply(CSRMatrix matrix)
    {
        bmsp_csr_t c = nullptr;
        check(bmsp_csr_multiply(h_, matrix.h_, &c));
        return CSRMatrix(c);
    }
    std::vector<float> multiply(const std::vector<float> &x)
    {
        int nr, nc; int64_t nnz;
        check(bmsp_csr_info(h_, &nr, &nc, &nnz));
        if ((int)x.size() != nc) throw std::runtime_error("CSRMatrix::multiply: vector length mismatch");
        std::vector<float> y((size_t)nr);
        check(bmsp_csr_spmv(h_, x.data(), y.data()));
        return y;
    }
    /* the reference's own path: cusp::multiply on the host container (no GPU involved); `threads` = 0 uses every core */
    CSRMatrix multiply_host(CSRMatrix matrix, int threads = 0)
    {
        bmsp_csr_t c = nullptr;
        check(bmsp_csr_multiply_host(h_, matrix.h_, &c, threads));
        return CSRMatrix(c);
    }
    std::vector<float> multiply_host(const std::vector<float> &x, int threads = 0)
    {
        int nr, nc; int64_t nnz;
        check(bmsp_csr_info(h_, &nr, &nc, &nnz));
        if ((int)x.size() != nc) throw std::runtime_error("CSRMatrix::multiply_host: vector length mismatch");
        std::vector<float> y((size_t)nr);
        check(bmsp_csr_spmv_host(h_, x.data(), y.data(), threads));
        return y;
    }
    bmsp_host_csr<int, float> host() const
    {
        bmsp_host_csr<int, float> m;
        int nr, nc; int64_t nnz; const int *ro, *ci; const float *v;
        check(bmsp_csr_info(h_, &nr, &nc, &nnz));
        check(bmsp_csr_arrays(h_, &ro, &ci, &v));
        m.num_rows = (size_t)nr; m.num_cols = (size_t)nc; m.num_entries = (size_t)nnz;
        m.row_offsets.assign(ro, ro + nr + 1); m.column_indices.assign(ci, ci + nnz); m.values.assign(v, v + nnz);
        return m;
    }

private:
    explicit CSRMatrix(bmsp_csr_t h) : h_(h) {}
    static void check(int st)
    {
        if (st != BMSP_OK) throw std::runtime_error(std::string("bmsp: ") + bmsp_last_error());
    }
    bmsp_csr_t h_ = nullptr; /* the reference's private member is `matrix_repr` */
};

#endif /* CSRMATRIX_H_ */
