// hisparse/data_loader.h — CSR/CSC containers and .npz ingest for the SpMV host path.
//
// Same public surface (names, argument meaning, namespace spmv::io) as the reference's
// sw/data_loader.h so a driver written against it compiles unchanged:
//   CSRMatrix<T> :19-30 · create_csr_matrix :35-47 · load_csr_matrix_from_float_npz :51-70 ·
//   csr_matrix_convert_from_float<T> :76-84 · CSCMatrix<T> :93-104 · csr2csc :109-144 ·
//   csc_matrix_convert_from_float<T> :149-157.
// Differences, all deliberate:
//   * the archive is read by hisparse/npz.h instead of cnpy (absent from the reference checkout);
//   * `shape`, `indices`, `indptr` are decoded by their declared dtype (i4 or i8) instead of the
//     reference's "read u32 words 0 and 2" trick (:55-56), which only works for int64 shapes;
//   * malformed input throws std::runtime_error instead of reading out of bounds.
#ifndef HISPARSE_DATA_LOADER_H_
#define HISPARSE_DATA_LOADER_H_

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

#include "npz.h"
#include "worker_pool.h"

namespace spmv {
namespace io {

template <typename data_type>
struct CSRMatrix {
    uint32_t num_rows = 0;
    uint32_t num_cols = 0;
    std::vector<data_type> adj_data;     // non-zero values, row-major
    std::vector<uint32_t> adj_indices;   // column of each non-zero
    std::vector<uint32_t> adj_indptr;    // num_rows + 1 offsets into the two arrays above
};

template <typename data_type>
CSRMatrix<data_type> create_csr_matrix(uint32_t num_rows, uint32_t num_cols,
                                       std::vector<data_type> const& adj_data,
                                       std::vector<uint32_t> const& adj_indices,
                                       std::vector<uint32_t> const& adj_indptr) {
    CSRMatrix<data_type> m;
    m.num_rows = num_rows;
    m.num_cols = num_cols;
    m.adj_data = adj_data;
    m.adj_indices = adj_indices;
    m.adj_indptr = adj_indptr;
    return m;
}

// scipy.sparse.save_npz layout: members shape[2], data[nnz], indices[nnz], indptr[rows+1], format.
inline CSRMatrix<float> load_csr_matrix_from_float_npz(std::string csr_float_npz_path) {
    auto members = hisparse::npz::load(csr_float_npz_path);
    for (const char* key : {"shape", "data", "indices", "indptr"})
        if (!members.count(key)) throw std::runtime_error(std::string("npz: missing member ") + key);
    const auto& shape = members["shape"];
    const auto& data = members["data"];
    const auto& indices = members["indices"];
    const auto& indptr = members["indptr"];
    if (shape.count() != 2) throw std::runtime_error("npz: shape must have 2 entries");
    if (shape.kind() == 'f' || indices.kind() == 'f' || indptr.kind() == 'f') throw std::runtime_error("npz: shape / indices / indptr must be integer arrays");
    const int64_t rows64 = shape.as_int(0), cols64 = shape.as_int(1);
    if (rows64 < 0 || cols64 < 0 || rows64 > 0xffffffffll || cols64 > 0xffffffffll) throw std::runtime_error("npz: shape outside the 32-bit index range");
    CSRMatrix<float> m;
    m.num_rows = uint32_t(rows64);
    m.num_cols = uint32_t(cols64);
    const uint64_t nnz = data.count();
    if (nnz > 0xffffffffull) throw std::runtime_error("npz: more than 2^32-1 non-zeros (32-bit indptr, sw/data_loader.h:24)");
    if (indices.count() != nnz || indptr.count() != uint64_t(m.num_rows) + 1)
        throw std::runtime_error("npz: inconsistent CSR array lengths");
    m.adj_data.resize(nnz);
    m.adj_indices.resize(nnz);
    m.adj_indptr.resize(indptr.count());
    for (uint64_t i = 0; i < nnz; ++i) m.adj_data[i] = data.as_float(i);
    // The formatter indexes per-partition tables by column and per-row tables by indptr (util_convert_csr_to_dds): a
    // corrupt archive must be refused here, with the same rules hsf_csr_from_arrays applies to caller-provided arrays.
    for (uint64_t i = 0; i < nnz; ++i) {
        const int64_t c = indices.as_int(i);
        if (c < 0 || c >= cols64) throw std::runtime_error("npz: column index out of range");
        m.adj_indices[i] = uint32_t(c);
    }
    int64_t prev = 0;
    for (uint64_t i = 0; i < indptr.count(); ++i) {
        const int64_t v = indptr.as_int(i);
        if (v < prev || uint64_t(v) > nnz || (i == 0 && v != 0)) throw std::runtime_error("npz: indptr must start at 0, be non-decreasing and stay within nnz");
        m.adj_indptr[i] = uint32_t(v);
        prev = v;
    }
    if (m.adj_indptr.back() != nnz) throw std::runtime_error("npz: indptr does not end at nnz");
    return m;
}

// Element-wise float -> data_type using data_type's converting constructor (for hisparse::q8_24
// that is round-half-up + saturate, negatives -> 0).
template <typename data_type>
CSRMatrix<data_type> csr_matrix_convert_from_float(CSRMatrix<float> const& in) {
    CSRMatrix<data_type> out;
    out.num_rows = in.num_rows;
    out.num_cols = in.num_cols;
    // (in parallel: 42 M conversions are 0.3 s of one core; HISPARSE_FORMAT_THREADS caps the workers like in csr2cpsr)
    const size_t n = in.adj_data.size();
    out.adj_data.resize(n);
    unsigned workers = std::thread::hardware_concurrency();
    if (const char* env = std::getenv("HISPARSE_FORMAT_THREADS")) workers = unsigned(std::max(1, std::atoi(env)));
    workers = unsigned(std::min<size_t>(std::max(1u, workers), n / 65536 + 1));
    auto convert = [&](size_t t) { for (size_t i = n * t / workers; i < n * (t + 1) / workers; ++i) out.adj_data[i] = data_type(in.adj_data[i]); };
    hisparse::pooled_for(workers, workers, convert);
    out.adj_indices = in.adj_indices;
    out.adj_indptr = in.adj_indptr;
    return out;
}

template <typename data_type>
struct CSCMatrix {
    uint32_t num_rows = 0;
    uint32_t num_cols = 0;
    std::vector<data_type> adj_data;
    std::vector<uint32_t> adj_indices;   // row of each non-zero
    std::vector<uint32_t> adj_indptr;    // num_cols + 1
};

// Counting-sort transpose; rows stay ascending inside each column (stable), like the reference.
template <typename data_type>
CSCMatrix<data_type> csr2csc(CSRMatrix<data_type> const& csr) {
    CSCMatrix<data_type> csc;
    csc.num_rows = csr.num_rows;
    csc.num_cols = csr.num_cols;
    const size_t nnz = csr.adj_indptr.empty() ? 0 : csr.adj_indptr[csr.num_rows];
    csc.adj_data.resize(nnz);
    csc.adj_indices.resize(nnz);
    csc.adj_indptr.assign(size_t(csr.num_cols) + 1, 0);
    for (size_t n = 0; n < nnz; ++n) csc.adj_indptr[csr.adj_indices[n] + 1]++;
    for (size_t c = 0; c < csr.num_cols; ++c) csc.adj_indptr[c + 1] += csc.adj_indptr[c];
    std::vector<uint32_t> cursor(csc.adj_indptr.begin(), csc.adj_indptr.end() - 1);
    for (uint32_t r = 0; r < csr.num_rows; ++r) {
        for (uint32_t p = csr.adj_indptr[r]; p < csr.adj_indptr[r + 1]; ++p) {
            uint32_t dst = cursor[csr.adj_indices[p]]++;
            csc.adj_indices[dst] = r;
            csc.adj_data[dst] = csr.adj_data[p];
        }
    }
    return csc;
}

template <typename data_type>
CSCMatrix<data_type> csc_matrix_convert_from_float(CSCMatrix<float> const& in) {
    CSCMatrix<data_type> out;
    out.num_rows = in.num_rows;
    out.num_cols = in.num_cols;
    // (in parallel: 42 M conversions are 0.3 s of one core; HISPARSE_FORMAT_THREADS caps the workers like in csr2cpsr)
    const size_t n = in.adj_data.size();
    out.adj_data.resize(n);
    unsigned workers = std::thread::hardware_concurrency();
    if (const char* env = std::getenv("HISPARSE_FORMAT_THREADS")) workers = unsigned(std::max(1, std::atoi(env)));
    workers = unsigned(std::min<size_t>(std::max(1u, workers), n / 65536 + 1));
    auto convert = [&](size_t t) { for (size_t i = n * t / workers; i < n * (t + 1) / workers; ++i) out.adj_data[i] = data_type(in.adj_data[i]); };
    hisparse::pooled_for(workers, workers, convert);
    out.adj_indices = in.adj_indices;
    out.adj_indptr = in.adj_indptr;
    return out;
}

}  // namespace io
}  // namespace spmv

#endif  // HISPARSE_DATA_LOADER_H_
