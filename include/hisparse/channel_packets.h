// hisparse/channel_packets.h — assembly of the 16 per-channel packet buffers, packed x and packed y.
//
// Every reference driver carries an inline copy of this block (sw/benchmark.cpp:127-195,
// spmv_csim/csim.cpp:229-297, sw/host.cpp:163-231); here it is one named function.  The buffers it
// produces are the drop-in boundary: byte for byte what the reference hands to the xclbin kernels /
// to csim's top_wrapper (layout restated in SURVEY.md Appendix A.2):
//
//   packet (1+F)*pid            : indices[0] = F * start_pid        (payload offset of partition pid)
//   packet (1+F)*pid + 1 + f    : indices[k] = length of lane k of virtual channel pc + 16 f
//   packet (1+F)*P + F*(start_pid + p) + f : element p of the 8 lanes of virtual channel pc + 16 f
//
// with pid = row_partition * num_col_partitions + col_partition, P = number of partitions,
// start_{pid+1} = start_pid + max over (f, k) of the lane lengths, everything else zero.
#ifndef HISPARSE_CHANNEL_PACKETS_H_
#define HISPARSE_CHANNEL_PACKETS_H_

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include "common.h"
#include "data_formatter.h"
#include "data_loader.h"
#include "q8_24.h"

namespace hisparse {

template <typename T>
struct PackedVal {  // PACKED_VAL_T
    T data[PACK_SIZE];
};
struct PackedIdx {  // PACKED_IDX_T
    uint32_t data[PACK_SIZE];
};

inline uint32_t to_bits(q8_24 v) { return v.raw; }
inline uint32_t to_bits(float v) { return f32_bits(v); }

template <typename DataT>
using Cpsr = spmv::io::CPSRMatrix<PackedVal<DataT>, PackedIdx, PACK_SIZE>;

// Everything that crosses the boundary for one matrix.
struct ChannelPackets {
    Geometry geom;
    uint32_t num_rows = 0, num_cols = 0;  // padded
    uint32_t num_row_partitions = 0, num_col_partitions = 0;
    uint64_t nnz = 0;  // true non-zeros (adj_data.size()), the metric numerator
    bool skip_empty_rows = false;
    std::vector<MatPkt> channel[NUM_HBM_CHANNELS];
    uint32_t num_partitions() const { return num_row_partitions * num_col_partitions; }
    // rows per cluster in row partition j — the `part_len` kernel argument (sw/benchmark.cpp:301-322)
    uint32_t part_len(uint32_t row_partition) const {
        uint64_t rows = geom.logical_ob;
        if (row_partition + 1 == num_row_partitions && num_rows % geom.logical_ob != 0) rows = num_rows % geom.logical_ob;
        return uint32_t(rows / NUM_HBM_CHANNELS);
    }
    uint64_t streamed_bytes() const {
        uint64_t b = 0;
        for (const auto& c : channel) b += c.size() * sizeof(MatPkt);
        return b;
    }
};

template <typename DataT>
void assemble_channel_packets(const Cpsr<DataT>& cpsr, const Geometry& g, ChannelPackets& out) {
    const unsigned F = g.interleave;
    if (cpsr.num_hbm_channels != NUM_HBM_CHANNELS * F) throw std::invalid_argument("assemble_channel_packets: CPSR channel count != 16 * interleave");
    const uint32_t parts = cpsr.num_row_partitions * cpsr.num_col_partitions;
    spmv::io::detail::parallel_for(NUM_HBM_CHANNELS, spmv::io::detail::format_threads(), [&](size_t pc) {
        // pass 1: payload length (in per-virtual-channel packets) and start of every partition
        std::vector<uint32_t> span(parts), start(parts);
        uint64_t total = 0;
        for (uint32_t pid = 0; pid < parts; ++pid) {
            uint32_t longest = 0;
            for (unsigned f = 0; f < F; ++f) {
                const auto& ptr = cpsr.formatted_adj_indptr[size_t(pid) * cpsr.num_hbm_channels + pc + f * NUM_HBM_CHANNELS];
                for (unsigned k = 0; k < PACK_SIZE; ++k) longest = std::max(longest, ptr.back().data[k]);
            }
            span[pid] = longest;
            start[pid] = uint32_t(total);
            total += longest;
        }
        const uint64_t header = uint64_t(parts) * (1 + F);
        std::vector<MatPkt>& buf = out.channel[pc];
        buf.assign(header + total * F, MatPkt{});
        // pass 2: headers and interleaved payload
        for (uint32_t pid = 0; pid < parts; ++pid) {
            buf[uint64_t(pid) * (1 + F)].indices.data[0] = start[pid] * F;
            for (unsigned f = 0; f < F; ++f) {
                const size_t slot = size_t(pid) * cpsr.num_hbm_channels + pc + f * NUM_HBM_CHANNELS;
                const auto& lens = cpsr.formatted_adj_indptr[slot].back();
                for (unsigned k = 0; k < PACK_SIZE; ++k) buf[uint64_t(pid) * (1 + F) + 1 + f].indices.data[k] = lens.data[k];
                const auto& idx = cpsr.formatted_adj_indices[slot];
                const auto& val = cpsr.formatted_adj_data[slot];
                MatPkt* dst = buf.data() + header + uint64_t(start[pid]) * F + f;
                for (size_t p = 0; p < idx.size(); ++p, dst += F) {
                    for (unsigned k = 0; k < PACK_SIZE; ++k) {
                        dst->indices.data[k] = idx[p].data[k];
                        dst->vals.data[k] = to_bits(val[p].data[k]);
                    }
                }
            }
        }
    });
}

// The whole host pre-processing of one matrix, in the order every reference driver performs it
// (sw/benchmark.cpp:110-195): round dims (mutates ext_matrix), convert values, csr2cpsr, assemble.
template <typename DataT>
ChannelPackets format_matrix_as(spmv::io::CSRMatrix<float>& ext_matrix, const Geometry& g, bool skip_empty_rows) {
    using namespace spmv::io;
    if (g.logical_ob > 0xffffffffull || g.logical_vb > 0xffffffffull) throw std::invalid_argument("bank sizes too large");
    const bool debug = std::getenv("HISPARSE_FORMAT_DEBUG") != nullptr;     // wall time of the phases
    auto t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto now = std::chrono::steady_clock::now();
        if (debug) std::fprintf(stderr, "format %-24s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    };
    util_round_csr_matrix_dim<float>(ext_matrix, g.row_divisor, PACK_SIZE);
    // csr_matrix_convert_from_float without the copies of the index arrays (0.17 GB on ogbl-ppa): the converted matrix BORROWS
    // them from the caller's for the duration of csr2cpsr (handed back on every path)
    CSRMatrix<DataT> mat;
    mat.num_rows = ext_matrix.num_rows;
    mat.num_cols = ext_matrix.num_cols;
    {
        CSRMatrix<float> values_only;
        values_only.adj_data.swap(ext_matrix.adj_data);
        struct GiveBack {
            std::vector<float>&a, &b;
            ~GiveBack() { a.swap(b); }
        } give_back_values{values_only.adj_data, ext_matrix.adj_data};
        mat.adj_data = std::move(csr_matrix_convert_from_float<DataT>(values_only).adj_data);
    }
    struct Borrowed {
        CSRMatrix<DataT>& to;
        CSRMatrix<float>& from;
        Borrowed(CSRMatrix<DataT>& t, CSRMatrix<float>& f) : to(t), from(f) { to.adj_indices.swap(from.adj_indices); to.adj_indptr.swap(from.adj_indptr); }
        ~Borrowed() { to.adj_indices.swap(from.adj_indices); to.adj_indptr.swap(from.adj_indptr); }
    };
    lap("round + convert");
    Cpsr<DataT> cpsr;
    uint64_t nnz = 0;
    {
        Borrowed borrowed(mat, ext_matrix);
        nnz = mat.adj_data.size();
        cpsr = csr2cpsr<PackedVal<DataT>, PackedIdx, DataT, uint32_t, PACK_SIZE>(mat, IDX_MARKER, uint32_t(g.logical_ob), uint32_t(g.logical_vb),
                                                                              g.virtual_channels, skip_empty_rows);
    }
    lap("csr2cpsr");
    ChannelPackets out;
    out.geom = g;
    out.num_rows = mat.num_rows;
    out.num_cols = mat.num_cols;
    out.num_row_partitions = cpsr.num_row_partitions;
    out.num_col_partitions = cpsr.num_col_partitions;
    out.nnz = nnz;
    out.skip_empty_rows = skip_empty_rows;
    assemble_channel_packets<DataT>(cpsr, g, out);
    lap("assemble channel packets");
    return out;
}

inline ChannelPackets format_matrix(spmv::io::CSRMatrix<float>& ext_matrix, const Geometry& g, bool skip_empty_rows) {
    return impl_is_float(g.impl) ? format_matrix_as<float>(ext_matrix, g, skip_empty_rows)
                                 : format_matrix_as<q8_24>(ext_matrix, g, skip_empty_rows);
}

// x: float -> value words, natural order, 8 per packet (sw/benchmark.cpp:207-212).
inline void pack_vector(int impl, const float* x, size_t n, uint32_t* words) {
    if (impl_is_float(impl)) for (size_t i = 0; i < n; ++i) words[i] = f32_bits(x[i]);
    else for (size_t i = 0; i < n; ++i) words[i] = q8_24_raw_from_double(x[i]);
}
// y: value words -> float (spmv_csim/csim.cpp:172,186-196).
inline void unpack_result(int impl, const uint32_t* words, size_t n, float* y) {
    if (impl_is_float(impl)) for (size_t i = 0; i < n; ++i) y[i] = bits_f32(words[i]);
    else for (size_t i = 0; i < n; ++i) y[i] = q8_24_raw_to_float(words[i]);
}

}  // namespace hisparse

#endif  // HISPARSE_CHANNEL_PACKETS_H_
