// hisparse/row_sharding.h — one matrix across the GPUs of a node, by contiguous row slabs (host side, C++).
//
// The reference is a single-device design; what it does have is row partitioning with no cross-partition state
// (sw/data_formatter.h:494,500-511; the launch loop sw/benchmark.cpp:318-338 runs the row partitions one after another and
// nothing flows between them).  That is the axis used for multi-GPU: every device gets a contiguous slab of rows whose
// interior boundaries are multiples of the row padding granule (128 * interleave), formats its slab with the ordinary
// host pipeline (a slab is a complete CPSR matrix of its own), and holds a full copy of x.  The only exchange is the
// all-gather of the y slabs.  Same rules as hisparse_amd/sharding.py (the Python face used by bench.py and the tests).
#ifndef HISPARSE_ROW_SHARDING_H_
#define HISPARSE_ROW_SHARDING_H_

#include <algorithm>
#include <cstdint>
#include <vector>

#include "data_loader.h"

namespace hisparse {

// Row boundaries b_0 = 0 <= b_1 <= ... <= b_parts = rows balancing non-zeros; interior ones are multiples of `granule`.
// Every slab gets at least one granule of rows when the matrix has that many; otherwise trailing slabs are empty.
inline std::vector<uint32_t> split_rows_by_nnz(const std::vector<uint32_t>& indptr, uint32_t parts, uint32_t granule) {
    const uint32_t rows = uint32_t(indptr.size() - 1);
    const uint64_t nnz = indptr.back();
    const uint64_t granules = (uint64_t(rows) + granule - 1) / granule;
    std::vector<uint32_t> bounds{0};
    for (uint32_t p = 1; p < parts; ++p) {
        const double target = double(nnz) * p / parts;
        const uint64_t r = uint64_t(std::lower_bound(indptr.begin(), indptr.end(), target, [](uint32_t v, double t) { return double(v) < t; }) -
                                    indptr.begin());
        uint64_t g = (2 * r + granule) / (2 * uint64_t(granule));   // round(r / granule)
        const uint64_t lo_g = bounds.back() / granule + 1;
        const int64_t hi_g = int64_t(granules) - int64_t(parts - p);
        if (hi_g >= int64_t(lo_g)) g = std::min<uint64_t>(std::max(g, lo_g), uint64_t(hi_g));
        else g = std::min<uint64_t>(lo_g, granules);
        bounds.push_back(uint32_t(std::min<uint64_t>(g * granule, rows)));
    }
    bounds.push_back(rows);
    return bounds;
}

// CSR of rows [lo, hi) with all columns.
inline spmv::io::CSRMatrix<float> row_slab(const spmv::io::CSRMatrix<float>& m, uint32_t lo, uint32_t hi) {
    spmv::io::CSRMatrix<float> s;
    s.num_rows = hi - lo;
    s.num_cols = m.num_cols;
    const uint32_t a = m.adj_indptr[lo], b = m.adj_indptr[hi];
    s.adj_indptr.resize(size_t(hi - lo) + 1);
    for (uint32_t r = lo; r <= hi; ++r) s.adj_indptr[r - lo] = m.adj_indptr[r] - a;
    s.adj_indices.assign(m.adj_indices.begin() + a, m.adj_indices.begin() + b);
    s.adj_data.assign(m.adj_data.begin() + a, m.adj_data.begin() + b);
    return s;
}

inline uint32_t padded_rows(uint32_t rows, uint32_t granule) { return (rows + granule - 1) / granule * granule; }

}  // namespace hisparse

#endif  // HISPARSE_ROW_SHARDING_H_
