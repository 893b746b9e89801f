// hisparse/data_formatter.h — CSR -> CPSR (cyclic packed streams of rows) pre-processing.
//
// Public surface and semantics follow the reference's sw/data_formatter.h (namespace spmv::io) so
// drivers and tests written against it read the same; every routine is re-implemented here:
//   util_round_csr_matrix_dim                :15-29    pad rows (append empty rows) / cols (count only)
//   util_normalize_csr_matrix_by_outdegree   :33-47
//   util_pad_marker_end_of_row[_*]           :51-187   append {IDX_MARKER, n} after rows
//   CPSRMatrix + get_packed_{data,indices,indptr} :196-238
//   util_convert_csr_to_dds                  :256-313  column partitioning, partition-local col ids
//   util_reorder_rows_ascending_nnz          :338-368  (unused by SpMV, kept for surface parity)
//   util_pack_rows                           :384-446  deal rows round-robin to (channel, lane) streams
//   csr2cpsr                                 :468-544
// Output is byte-identical to the reference's for the same input (checked against the goldens of
// unit_tests/test_io.cpp and against oracle/cpsr_format.py).  What differs is the implementation:
// counts are computed before anything is materialised, the skip-count of a marker comes from one
// reverse scan per residue class instead of nested forward walks, column partitions of a row
// partition are formatted concurrently (HISPARSE_FORMAT_THREADS, default = hardware threads), and
// argument errors throw std::invalid_argument instead of calling exit().
#ifndef HISPARSE_DATA_FORMATTER_H_
#define HISPARSE_DATA_FORMATTER_H_

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "common.h"
#include "data_loader.h"
#include "worker_pool.h"
#include "q8_24.h"

namespace spmv {
namespace io {

namespace detail {

// The value word that accompanies an end-of-row marker carries the row advance n.
// float streams: the raw integer bit pattern of n placed in the float (data_formatter.h:69-71,154-156);
// any other value type: n converted by the type's integer constructor (:73,158) — for Q8.24 that is
// n.0, saturating at n >= 256 (the decoder reads the top 8 bits, spmv_cluster.h:81-82).
template <typename DataT>
inline DataT marker_word(uint32_t n) {
    if constexpr (std::is_same<DataT, float>::value) {
        float f;
        std::memcpy(&f, &n, 4);
        return f;
    } else {
        return DataT(n);
    }
}

// Largest row advance ONE marker can carry.  float streams hold the raw integer; a Q8.24 word holds it in its 8 integer
// bits, which is all the decoder reads (spmv_cluster.h:81-82): n >= 256 would saturate to 0xffffffff and decode as 255,
// silently landing every later row of that lane stream too early (SURVEY.md Appendix B.7 -- the reference has no check).
template <typename DataT>
constexpr uint32_t marker_limit() {
    return std::is_same<DataT, float>::value ? 0xffffffffu : 255u;
}

inline unsigned format_threads() {
    if (const char* e = std::getenv("HISPARSE_FORMAT_THREADS")) {
        int v = std::atoi(e);
        if (v > 0) return unsigned(v);
    }
    unsigned hw = std::thread::hardware_concurrency();
    return hw ? hw : 1u;
}

// run fn(i) for i in [0, n) on up to `threads` threads (worker_pool.h: parked workers, not a thread start per loop)
template <typename Fn>
inline void parallel_for(size_t n, unsigned threads, Fn fn) {
    hisparse::pooled_for(n, threads, fn);
}

}  // namespace detail

// Pad the row count up to a multiple of row_divisor by appending empty rows, and the column COUNT up
// to a multiple of col_divisor.  Mutates the caller's matrix, as the reference does (callers size x
// by the padded num_cols afterwards, sw/benchmark.cpp:110,205).
template <typename DataT>
void util_round_csr_matrix_dim(CSRMatrix<DataT>& csr_matrix, uint32_t row_divisor, uint32_t col_divisor) {
    uint32_t row_rem = csr_matrix.num_rows % row_divisor;
    if (row_rem != 0) {
        uint32_t extra = row_divisor - row_rem;
        csr_matrix.adj_indptr.resize(size_t(csr_matrix.num_rows) + 1 + extra, csr_matrix.adj_indptr[csr_matrix.num_rows]);
        csr_matrix.num_rows += extra;
    }
    uint32_t col_rem = csr_matrix.num_cols % col_divisor;
    if (col_rem != 0) csr_matrix.num_cols += col_divisor - col_rem;
}

// value := 1 / (number of non-zeros in the value's column)
template <typename DataT>
void util_normalize_csr_matrix_by_outdegree(CSRMatrix<DataT>& csr_matrix) {
    std::vector<uint32_t> col_population(csr_matrix.num_cols, 0);
    for (uint32_t c : csr_matrix.adj_indices) col_population[c]++;
    for (size_t p = 0; p < csr_matrix.adj_indices.size(); ++p)
        csr_matrix.adj_data[p] = 1.0 / col_population[csr_matrix.adj_indices[p]];
}

// One marker {idx_marker, 1} after EVERY row, empty or not.
template <typename DataT>
void util_pad_marker_end_of_row_no_skip_empty_rows(std::vector<DataT>& adj_data, std::vector<uint32_t>& adj_indices,
                                                   std::vector<uint32_t>& adj_indptr, uint32_t idx_marker) {
    const size_t rows = adj_indptr.size() - 1;
    std::vector<DataT> data_out;
    std::vector<uint32_t> idx_out;
    data_out.reserve(adj_data.size() + rows);
    idx_out.reserve(adj_indices.size() + rows);
    const DataT one = detail::marker_word<DataT>(1);
    for (size_t r = 0; r < rows; ++r) {
        data_out.insert(data_out.end(), adj_data.begin() + adj_indptr[r], adj_data.begin() + adj_indptr[r + 1]);
        idx_out.insert(idx_out.end(), adj_indices.begin() + adj_indptr[r], adj_indices.begin() + adj_indptr[r + 1]);
        data_out.push_back(one);
        idx_out.push_back(idx_marker);
    }
    for (size_t r = 1; r <= rows; ++r) adj_indptr[r] += uint32_t(r);
    adj_data.swap(data_out);
    adj_indices.swap(idx_out);
}

// Markers only after non-empty rows, except that the first `interleave_stride` rows (the head of
// every lane stream) always get one.  The marker's count n = 1 + number of empty rows that directly
// follow in the same residue class (row, row+stride, row+2*stride, ...), so the decoder can jump.
// An advance beyond what one marker word can carry (detail::marker_limit: 255 in fixed point) is emitted as a CHAIN of
// markers (255, 255, ..., remainder): every decoder -- the reference's included -- adds consecutive markers up, so the
// image stays valid where the reference's own formatter writes a saturated count and corrupts the row numbering.
// Byte-identical to the reference wherever the reference is itself correct (advance <= 255).
template <typename DataT>
void util_pad_marker_end_of_row_skip_empty_rows(std::vector<DataT>& adj_data, std::vector<uint32_t>& adj_indices,
                                                std::vector<uint32_t>& adj_indptr, uint32_t idx_marker,
                                                uint32_t interleave_stride) {
    const size_t rows = adj_indptr.size() - 1;
    if (interleave_stride == 0 || rows % interleave_stride != 0)
        throw std::invalid_argument("util_pad_marker_end_of_row: rows must be a multiple of the interleave stride");
    // gets_marker[r]: r keeps a marker.  trailing_empties[r]: empty rows after r in its class before the next kept row.
    std::vector<uint8_t> gets_marker(rows);
    std::vector<uint32_t> advance(rows, 0);
    for (size_t r = 0; r < rows; ++r) gets_marker[r] = (r < interleave_stride) || (adj_indptr[r + 1] != adj_indptr[r]);
    for (size_t tail = rows; tail-- > rows - interleave_stride;) {  // last row of each residue class
        uint32_t gap = 0;                                           // empties seen since the last kept row (scanning upward)
        for (size_t r = tail;; r -= interleave_stride) {
            if (gets_marker[r]) { advance[r] = 1 + gap; gap = 0; } else { ++gap; }
            if (r < interleave_stride) break;
        }
    }
    constexpr uint32_t limit = detail::marker_limit<DataT>();
    size_t kept = 0;
    for (size_t r = 0; r < rows; ++r) kept += gets_marker[r] ? (size_t(advance[r]) + limit - 1) / limit : 0;
    std::vector<DataT> data_out;
    std::vector<uint32_t> idx_out;
    data_out.reserve(adj_data.size() + kept);
    idx_out.reserve(adj_indices.size() + kept);
    std::vector<uint32_t> new_indptr(rows + 1, 0);
    for (size_t r = 0; r < rows; ++r) {
        if (gets_marker[r]) {
            data_out.insert(data_out.end(), adj_data.begin() + adj_indptr[r], adj_data.begin() + adj_indptr[r + 1]);
            idx_out.insert(idx_out.end(), adj_indices.begin() + adj_indptr[r], adj_indices.begin() + adj_indptr[r + 1]);
            for (uint32_t left = advance[r]; left != 0;) {
                const uint32_t n = std::min(left, limit);
                data_out.push_back(detail::marker_word<DataT>(n));
                idx_out.push_back(idx_marker);
                left -= n;
            }
        }
        new_indptr[r + 1] = uint32_t(data_out.size());
    }
    adj_data.swap(data_out);
    adj_indices.swap(idx_out);
    adj_indptr.swap(new_indptr);
}

template <typename DataT>
void util_pad_marker_end_of_row(std::vector<DataT>& adj_data, std::vector<uint32_t>& adj_indices,
                                std::vector<uint32_t>& adj_indptr, uint32_t idx_marker, uint32_t interleave_stride,
                                bool skip_empty_rows = false) {
    if (skip_empty_rows)
        util_pad_marker_end_of_row_skip_empty_rows(adj_data, adj_indices, adj_indptr, idx_marker, interleave_stride);
    else
        util_pad_marker_end_of_row_no_skip_empty_rows(adj_data, adj_indices, adj_indptr, idx_marker);
}

// CPSR container: one (data, indices, indptr) triple per (row partition, column partition, channel).
// `num_hbm_channels` counts VIRTUAL channels (physical channels x interleave factor).
template <typename packed_val_t, typename packed_idx_t, uint32_t pack_size>
struct CPSRMatrix {
    uint32_t num_row_partitions = 0;
    uint32_t num_col_partitions = 0;
    uint32_t num_hbm_channels = 0;
    bool skip_empty_rows = false;

    std::vector<std::vector<packed_val_t> > formatted_adj_data;
    std::vector<std::vector<packed_idx_t> > formatted_adj_indices;
    std::vector<std::vector<packed_idx_t> > formatted_adj_indptr;

    size_t slot(uint32_t row_partition_idx, uint32_t col_partition_idx, uint32_t hbm_channel_idx) const {
        return (size_t(row_partition_idx) * num_col_partitions + col_partition_idx) * num_hbm_channels + hbm_channel_idx;
    }
    // Same names as the reference's accessors (:212-237); these return references instead of copies.
    const std::vector<packed_val_t>& get_packed_data(uint32_t rp, uint32_t cp, uint32_t ch) const { return formatted_adj_data[slot(rp, cp, ch)]; }
    const std::vector<packed_idx_t>& get_packed_indices(uint32_t rp, uint32_t cp, uint32_t ch) const { return formatted_adj_indices[slot(rp, cp, ch)]; }
    const std::vector<packed_idx_t>& get_packed_indptr(uint32_t rp, uint32_t cp, uint32_t ch) const { return formatted_adj_indptr[slot(rp, cp, ch)]; }
};

// Split the columns into partitions of num_cols_per_partition ("dense-dense-sparse"); column ids
// become partition-local; order inside a row is preserved.  Output arrays are indexed by partition.
template <typename DataT>
void util_convert_csr_to_dds(uint32_t num_rows, uint32_t num_cols, const DataT* adj_data, const uint32_t* adj_indices,
                             const uint32_t* adj_indptr, uint32_t num_cols_per_partition,
                             std::vector<DataT> partitioned_adj_data[], std::vector<uint32_t> partitioned_adj_indices[],
                             std::vector<uint32_t> partitioned_adj_indptr[]) {
    const uint32_t parts = (num_cols + num_cols_per_partition - 1) / num_cols_per_partition;
    const unsigned threads = detail::format_threads();
    detail::parallel_for(parts, threads, [&](size_t p) { partitioned_adj_indptr[p].assign(size_t(num_rows) + 1, 0); });
    // rows in chunks of about equal non-zero count, one task each (the rows of different chunks are disjoint, so are their
    // stretches of every partition's arrays): per-row population of every partition, a prefix sum down the rows of every
    // partition, then the scatter with every chunk starting at its first row's offsets
    const uint32_t nnz = adj_indptr[num_rows] - adj_indptr[0];
    const uint32_t chunks = std::max<uint32_t>(1, std::min<uint32_t>(threads * 4, num_rows / 256 + 1));
    std::vector<uint32_t> first(chunks + 1, num_rows);
    first[0] = 0;
    for (uint32_t c = 1, r = 0; c < chunks; ++c) {
        const uint64_t target = uint64_t(nnz) * c / chunks + adj_indptr[0];
        while (r < num_rows && adj_indptr[r] < target) ++r;
        first[c] = r;
    }
    detail::parallel_for(chunks, threads, [&](size_t c) {
        for (uint32_t r = first[c]; r < first[c + 1]; ++r)
            for (uint32_t e = adj_indptr[r]; e < adj_indptr[r + 1]; ++e)
                partitioned_adj_indptr[adj_indices[e] / num_cols_per_partition][r + 1]++;
    });
    detail::parallel_for(parts, threads, [&](size_t p) {
        auto& ptr = partitioned_adj_indptr[p];
        for (uint32_t r = 0; r < num_rows; ++r) ptr[r + 1] += ptr[r];
        partitioned_adj_data[p].resize(ptr[num_rows]);
        partitioned_adj_indices[p].resize(ptr[num_rows]);
    });
    detail::parallel_for(chunks, threads, [&](size_t c) {
        std::vector<uint32_t> cursor(parts);
        for (uint32_t p = 0; p < parts; ++p) cursor[p] = partitioned_adj_indptr[p][first[c]];
        for (uint32_t r = first[c]; r < first[c + 1]; ++r) {
            for (uint32_t e = adj_indptr[r]; e < adj_indptr[r + 1]; ++e) {
                const uint32_t col = adj_indices[e];
                const uint32_t p = col / num_cols_per_partition;
                const uint32_t at = cursor[p]++;
                partitioned_adj_data[p][at] = adj_data[e];
                partitioned_adj_indices[p][at] = col - p * num_cols_per_partition;
            }
        }
    });
}

// Stable sort of the rows by non-zero count (unused by the SpMV path; kept for surface parity).
template <typename DataT>
void util_reorder_rows_ascending_nnz(std::vector<DataT> const& adj_data, std::vector<uint32_t> const& adj_indices,
                                     std::vector<uint32_t> const& adj_indptr, std::vector<DataT>& reordered_adj_data,
                                     std::vector<uint32_t>& reordered_adj_indices, std::vector<uint32_t>& reordered_adj_indptr) {
    const size_t rows = adj_indptr.size() - 1;
    std::vector<uint32_t> order(rows);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return adj_indptr[a + 1] - adj_indptr[a] < adj_indptr[b + 1] - adj_indptr[b];
    });
    reordered_adj_indptr.push_back(0);
    for (uint32_t r : order) {
        reordered_adj_data.insert(reordered_adj_data.end(), adj_data.begin() + adj_indptr[r], adj_data.begin() + adj_indptr[r + 1]);
        reordered_adj_indices.insert(reordered_adj_indices.end(), adj_indices.begin() + adj_indptr[r], adj_indices.begin() + adj_indptr[r + 1]);
        reordered_adj_indptr.push_back(reordered_adj_indptr.back() + (adj_indptr[r + 1] - adj_indptr[r]));
    }
}

// Deal rows to streams: row i*(channels*pack) + c*pack + j belongs to channel c, lane j, round i.
// Lane streams are the concatenation of their rows; arrays are sized to the longest lane of the
// channel and zero filled.  packed_adj_indptr[c][i] = per-lane element count before round i.
namespace detail {
// one channel of util_pack_rows (the channels only read the shared CSR arrays: csr2cpsr runs them as separate tasks)
template <typename DataT, typename packed_val_t, typename packed_idx_t>
void pack_rows_of_channel(std::vector<DataT> const& adj_data, std::vector<uint32_t> const& adj_indices, std::vector<uint32_t> const& adj_indptr,
                          uint32_t num_hbm_channels, uint32_t pack_size, uint32_t c, std::vector<packed_val_t>& packed_adj_data,
                          std::vector<packed_idx_t>& packed_adj_indices, std::vector<packed_idx_t>& packed_adj_indptr) {
    const size_t rows = adj_indptr.size() - 1;
    const size_t rows_per_round = size_t(num_hbm_channels) * pack_size;
    const size_t rounds = (rows + rows_per_round - 1) / rows_per_round;
    packed_idx_t running;
    for (uint32_t j = 0; j < pack_size; ++j) running.data[j] = 0;
    auto& ptr = packed_adj_indptr;
    ptr.reserve(ptr.size() + rounds + 1);
    ptr.push_back(running);
    for (size_t i = 0; i < rounds; ++i) {
        for (uint32_t j = 0; j < pack_size; ++j) {
            const size_t r = i * rows_per_round + size_t(c) * pack_size + j;
            if (r < rows) running.data[j] += adj_indptr[r + 1] - adj_indptr[r];
        }
        ptr.push_back(running);
    }
    uint32_t longest = 0;
    for (uint32_t j = 0; j < pack_size; ++j) longest = std::max(longest, running.data[j]);
    packed_adj_data.resize(longest);
    packed_adj_indices.resize(longest);
    for (uint32_t j = 0; j < pack_size; ++j) {
        uint32_t at = 0;
        for (size_t r = size_t(c) * pack_size + j; r < rows; r += rows_per_round)
            for (uint32_t e = adj_indptr[r]; e < adj_indptr[r + 1]; ++e, ++at) {
                packed_adj_data[at].data[j] = adj_data[e];
                packed_adj_indices[at].data[j] = adj_indices[e];
            }
    }
}
}  // namespace detail

template <typename DataT, typename packed_val_t, typename packed_idx_t>
void util_pack_rows(std::vector<DataT> const& adj_data, std::vector<uint32_t> const& adj_indices,
                    std::vector<uint32_t> const& adj_indptr, uint32_t num_hbm_channels, uint32_t pack_size,
                    std::vector<packed_val_t> packed_adj_data[], std::vector<packed_idx_t> packed_adj_indices[],
                    std::vector<packed_idx_t> packed_adj_indptr[]) {
    for (uint32_t c = 0; c < num_hbm_channels; ++c)
        detail::pack_rows_of_channel<DataT, packed_val_t, packed_idx_t>(adj_data, adj_indices, adj_indptr, num_hbm_channels, pack_size, c,
                                                                       packed_adj_data[c], packed_adj_indices[c], packed_adj_indptr[c]);
}

// CSR -> CPSR.  out_buf_len rows per row partition, vec_buf_len columns per column partition,
// num_hbm_channels VIRTUAL channels.  The matrix must already be padded (util_round_csr_matrix_dim).
template <typename packed_val_t, typename packed_idx_t, typename DataT, typename IndexT, uint32_t pack_size>
CPSRMatrix<packed_val_t, packed_idx_t, pack_size> csr2cpsr(CSRMatrix<DataT> const& csr_matrix, uint32_t idx_marker,
                                                        uint32_t out_buf_len, uint32_t vec_buf_len,
                                                        uint32_t num_hbm_channels, bool skip_empty_rows) {
    const uint32_t stride = pack_size * num_hbm_channels;
    if (csr_matrix.num_rows % stride != 0)
        throw std::invalid_argument("csr2cpsr: number of rows must be a multiple of " + std::to_string(stride) +
                                    " (use spmv::io::util_round_csr_matrix_dim)");
    if (csr_matrix.num_cols % pack_size != 0)
        throw std::invalid_argument("csr2cpsr: number of columns must be a multiple of " + std::to_string(pack_size) +
                                    " (use spmv::io::util_round_csr_matrix_dim)");
    if (out_buf_len == 0 || out_buf_len % stride != 0) throw std::invalid_argument("csr2cpsr: out_buf_len must be a positive multiple of pack_size*num_hbm_channels");
    if (vec_buf_len == 0 || vec_buf_len % pack_size != 0) throw std::invalid_argument("csr2cpsr: vec_buf_len must be a positive multiple of pack_size");

    CPSRMatrix<packed_val_t, packed_idx_t, pack_size> out;
    out.skip_empty_rows = skip_empty_rows;
    out.num_hbm_channels = num_hbm_channels;
    out.num_row_partitions = (csr_matrix.num_rows + out_buf_len - 1) / out_buf_len;
    out.num_col_partitions = (csr_matrix.num_cols + vec_buf_len - 1) / vec_buf_len;
    const size_t slots = size_t(out.num_row_partitions) * out.num_col_partitions * num_hbm_channels;
    out.formatted_adj_data.resize(slots);
    out.formatted_adj_indices.resize(slots);
    out.formatted_adj_indptr.resize(slots);
    const unsigned threads = detail::format_threads();
    const bool debug = std::getenv("HISPARSE_FORMAT_DEBUG") != nullptr;
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (debug) std::fprintf(stderr, "  csr2cpsr %-22s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
        t_lap = now;
    };
    lap("slot vectors");

    for (uint32_t rp = 0; rp < out.num_row_partitions; ++rp) {
        const uint32_t first_row = rp * out_buf_len;
        const uint32_t rows_here = std::min<uint32_t>(out_buf_len, csr_matrix.num_rows - first_row);
        const IndexT base = csr_matrix.adj_indptr[first_row];
        std::vector<IndexT> local_indptr(size_t(rows_here) + 1);
        for (uint32_t r = 0; r <= rows_here; ++r) local_indptr[r] = csr_matrix.adj_indptr[first_row + r] - base;

        std::vector<std::vector<DataT> > part_data(out.num_col_partitions);
        std::vector<std::vector<IndexT> > part_indices(out.num_col_partitions);
        std::vector<std::vector<IndexT> > part_indptr(out.num_col_partitions);
        lap("local indptr");
        util_convert_csr_to_dds<DataT>(rows_here, csr_matrix.num_cols, csr_matrix.adj_data.data() + base,
                                       csr_matrix.adj_indices.data() + base, local_indptr.data(), vec_buf_len,
                                       part_data.data(), part_indices.data(), part_indptr.data());
        lap("dds split");
        detail::parallel_for(out.num_col_partitions, threads, [&](size_t cp) {
            util_pad_marker_end_of_row<DataT>(part_data[cp], part_indices[cp], part_indptr[cp], idx_marker, stride, skip_empty_rows);
        });
        lap("markers");
        // util_pack_rows, one task per (column partition, channel): a matrix with two column partitions (mouse_gene) still
        // keeps 32 workers busy
        detail::parallel_for(size_t(out.num_col_partitions) * num_hbm_channels, threads, [&](size_t task) {
            const uint32_t cp = uint32_t(task / num_hbm_channels), c = uint32_t(task % num_hbm_channels);
            const size_t s = out.slot(rp, cp, c);
            detail::pack_rows_of_channel<DataT, packed_val_t, packed_idx_t>(part_data[cp], part_indices[cp], part_indptr[cp], num_hbm_channels, pack_size, c,
                                                                           out.formatted_adj_data[s], out.formatted_adj_indices[s],
                                                                           out.formatted_adj_indptr[s]);
        });
        lap("pack rows");
    }
    return out;
}

}  // namespace io
}  // namespace spmv

#endif  // HISPARSE_DATA_FORMATTER_H_
