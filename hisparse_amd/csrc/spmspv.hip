// spmspv.hip — SpMSpV on gfx950: y = A x for a SPARSE x over a CSC matrix.  EXTENSION.
//
// The reference only stubs this operator -- the packet / pair types SPMSPV_MAT_PKT_T and IDX_VAL_T
// (spmv/libfpga/common.h:52-54) and the CSC conversion csr2csc (sw/data_loader.h:109-144); the paper (section 7) names it as the
// natural next kernel on the same datapath.  The work is proportional to the non-zeros of the SELECTED columns only, which is the
// point of the operator.
//
// Round 3: the SpMV kernels' row-owner scheme instead of a device-scope atomic per product (memory-side atomics run at ~24 G/s on
// this chip):
//   1. lengths of the selected columns -> exclusive scan (hipCUB) -> every column's place in an element list, and the total;
//   2. EXPAND: one wavefront per stored x entry streams that entry's matrix column (row index + value word, both contiguous in
//      CSC: two coalesced loads per 64 non-zeros), multiplies with the PE arithmetic of the numeric mode and writes
//      (row, product word) to its place in the list;
//   3. BIN: one radix-sort pass structure over the row's HIGH bits only (hipcub::DeviceRadixSort on bits [13, log2 rows)): the list
//      ordered by row block of 8192 rows, unordered inside a block;
//   4. ACCUMULATE: one workgroup per row block finds its stretch of the list by binary search, adds the products into 64-bit LDS
//      accumulators (ds_add_u64 / ds_add_f64: exact integer sums, double sums of the fp32 products) and writes ITS rows of y --
//      every row exactly once, so y needs no zeroing pass and no finish pass.
// No global atomics anywhere on this path.  It pays from a few million products on matrices with at least 32 row blocks (measured
// crossovers below); smaller jobs go through the round-2 scatter kernel (a device-scope atomic per product into a zeroed accumulator
// vector), which is launch-bound at that size either way.  HISPARSE_SPMSPV=atomic forces the scatter (A/B runs).
//   fixed: products rounded / saturated one by one (q8_24_mul), summed exactly in 64 bits, clamped once -- bit-identical to the
//          saturating PE sum, in any order;
//   float: one fp32 multiply per product; the sum order is not fixed (LDS atomics, or memory-side ones on the direct path): tolerance
//          parity like every float path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "spmv_device.h"
#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr uint32_t kBlockBits = 13;                       // a row block = 8192 rows: 64 KiB of 8-byte LDS accumulators
constexpr uint32_t kBlockRows = 1u << kBlockBits;
// The binned path costs ~200 us before the first product (scan, one stream sync, expand, sort, accumulate) and runs one workgroup per
// 8192-row block; the direct scatter costs ~50 us + 1 us per 20 K products.  Measured (tools/spmspv_probe.py, profiles/r03_spmspv.txt):
// ogbl-ppa, 4.3 M products: 205 vs 273 us, 21 M: 610 vs 1023 us; 0.4 M: 205 vs 86 us; mouse_gene (6 row blocks): binned slower at any size.
constexpr uint32_t kSpmspvDirectLimit = 1u << 21;         // fewer products than this: the direct scatter
constexpr uint32_t kSpmspvMinBlocks = 32;                 // fewer row blocks than this: too little parallelism in the accumulate pass

__global__ __launch_bounds__(256) void spmspv_lengths_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ x_index, uint32_t x_count,
                                                            uint32_t num_cols, uint32_t* __restrict__ len) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > x_count) return;
    uint32_t n = 0;
    if (k < x_count) {
        const uint32_t col = x_index[k];
        if (col < num_cols) n = indptr[col + 1] - indptr[col];      // (checked on the host as well; never read out of range)
    }
    len[k] = n;                                                     // len[x_count] = 0: the exclusive scan leaves the total there
}

template <bool kFloat>
__device__ __forceinline__ uint32_t product_word(uint32_t value_word, uint32_t x_word) {
    if (kFloat) return __float_as_uint(__uint_as_float(value_word) * __uint_as_float(x_word));
    return q8_24_mul(value_word, x_word);
}

template <bool kFloat>
__global__ __launch_bounds__(256) void spmspv_expand_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ row_indices,
                                                           const uint32_t* __restrict__ value_words, const uint32_t* __restrict__ x_index,
                                                           const uint32_t* __restrict__ x_words, uint32_t x_count, uint32_t num_cols,
                                                           const uint32_t* __restrict__ place, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const uint32_t lane = threadIdx.x & (kWaveLanes - 1);
    const uint32_t wave = blockIdx.x * (blockDim.x / kWaveLanes) + threadIdx.x / kWaveLanes;
    const uint32_t waves = gridDim.x * (blockDim.x / kWaveLanes);
    for (uint32_t k = wave; k < x_count; k += waves) {
        const uint32_t col = x_index[k];
        if (col >= num_cols) continue;
        const uint32_t xw = x_words[k];
        const uint32_t lo = indptr[col], hi = indptr[col + 1], at = place[k];
        for (uint32_t e = lo + lane; e < hi; e += kWaveLanes) {
            keys[at + (e - lo)] = row_indices[e];
            vals[at + (e - lo)] = product_word<kFloat>(value_words[e], xw);
        }
    }
}

// One workgroup per row block: its products sit in [first key with key >> 13 >= b, first key with key >> 13 > b) of the binned list.
template <bool kFloat>
__global__ __launch_bounds__(1024) void spmspv_accumulate_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, uint32_t total,
                                                                uint32_t* __restrict__ y, uint32_t num_rows) {
    using R = Rows<kFloat>;
    __shared__ typename R::acc_t acc[kBlockRows];
    const uint32_t b = blockIdx.x, row0 = b << kBlockBits;
    for (uint32_t i = threadIdx.x; i < kBlockRows; i += blockDim.x) acc[i] = 0;
    auto first_at_or_above = [&](uint32_t block) {          // rows are binned by block only: monotone in key >> kBlockBits
        uint32_t lo = 0, hi = total;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            if ((keys[mid] >> kBlockBits) < block) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const uint32_t begin = first_at_or_above(b), end = first_at_or_above(b + 1);
    __syncthreads();
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t local = keys[i] & (kBlockRows - 1u);
        if (kFloat) R::add(acc, local, __uint_as_float(vals[i]));                          // ds_add_f64 of the fp32 product
        else atomicAdd(acc + local, static_cast<unsigned long long>(vals[i]));             // ds_add_u64
    }
    // no-return LDS atomics can outlive s_waitcnt lgkmcnt(0) (spmv_kernels.hip): a RETURNING atomic per wavefront, awaited
    const typename R::acc_t flushed = atomicAdd(acc + (threadIdx.x / kWaveLanes), static_cast<typename R::acc_t>(0));
    asm volatile("" ::"v"(flushed));
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kBlockRows && row0 + i < num_rows; i += blockDim.x) y[row0 + i] = R::finish(acc[i]);
}

// ---- the direct path for a handful of products ------------------------------------------------------------------------------------
template <bool kFloat>
__global__ __launch_bounds__(256) void spmspv_scatter_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ row_indices,
                                                            const uint32_t* __restrict__ value_words, const uint32_t* __restrict__ x_index,
                                                            const uint32_t* __restrict__ x_words, uint32_t x_count, uint32_t num_cols,
                                                            unsigned long long* __restrict__ acc64, float* __restrict__ acc32) {
    const uint32_t lane = threadIdx.x & (kWaveLanes - 1);
    const uint32_t wave = blockIdx.x * (blockDim.x / kWaveLanes) + threadIdx.x / kWaveLanes;
    const uint32_t waves = gridDim.x * (blockDim.x / kWaveLanes);
    for (uint32_t k = wave; k < x_count; k += waves) {
        const uint32_t col = x_index[k];
        if (col >= num_cols) continue;
        const uint32_t xw = x_words[k];
        const uint32_t lo = indptr[col], hi = indptr[col + 1];
        for (uint32_t e = lo + lane; e < hi; e += kWaveLanes) {
            const uint32_t row = row_indices[e];
            if (kFloat) atomicAdd(acc32 + row, __uint_as_float(value_words[e]) * __uint_as_float(xw));
            else atomicAdd(acc64 + row, static_cast<unsigned long long>(q8_24_mul(value_words[e], xw)));
        }
    }
}

template <bool kFloat>
__global__ __launch_bounds__(256) void spmspv_finish_kernel(const unsigned long long* __restrict__ acc64, const float* __restrict__ acc32,
                                                           uint32_t* __restrict__ y, uint32_t num_rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_rows) return;
    if (kFloat) y[i] = __float_as_uint(acc32[i]);
    else y[i] = acc64[i] > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(acc64[i]);      // AP_SAT (pe.h:72)
}

uint32_t bits_for(uint32_t n) {       // smallest b with 2^b >= n
    uint32_t b = 0;
    while ((uint64_t(1) << b) < n) ++b;
    return b;
}

}  // namespace

size_t spmspv_sort_temp_bytes(uint64_t max_elements, uint32_t num_rows) {
    size_t bytes = 0;
    uint32_t* k = nullptr;
    const int end_bit = int(std::max(kBlockBits + 1, bits_for(num_rows)));
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k, k, k, k, size_t(std::max<uint64_t>(max_elements, 1)), int(kBlockBits), end_bit, nullptr);
    size_t scan = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan, k, k, size_t(1) << 24, nullptr);      // (x entries: far fewer than this)
    return std::max(bytes, scan) + 256;
}

hipError_t launch_spmspv(bool is_float, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, const uint32_t* x_index,
                         const uint32_t* x_words, uint32_t x_count, uint32_t num_rows, uint32_t num_cols, const SpmspvScratch& s, uint32_t* y,
                         hipStream_t stream, uint64_t* products_out, const char* force_path) {
    hipError_t e;
    uint32_t total = 0;
    if (x_count) {
        // 1. where every selected column's products go, and how many there are
        hipLaunchKernelGGL(spmspv_lengths_kernel, dim3((x_count + 256) / 256), dim3(256), 0, stream, indptr, x_index, x_count, num_cols, s.lengths);
        size_t temp = s.temp_bytes;
        if ((e = hipcub::DeviceScan::ExclusiveSum(s.temp, temp, s.lengths, s.place, size_t(x_count) + 1, stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(&total, s.place + x_count, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        if (total > s.capacity) return hipErrorInvalidValue;      // cannot happen: capacity = the matrix's non-zeros
    }
    if (products_out) *products_out = total;
    const dim3 expand_grid(std::min<uint32_t>((x_count + 3) / 4, 4096)), block(256);       // 4 wavefronts per workgroup, one x entry each
    const char* force = force_path;      // atomic | binned: force a path (hs_set_option "spmspv" / HISPARSE_SPMSPV; tests, A/B runs)
    const bool force_direct = force && std::string(force) == "atomic", force_binned = force && std::string(force) == "binned" && total > 0;
    if (!force_binned && (total < kSpmspvDirectLimit || (num_rows + kBlockRows - 1) / kBlockRows < kSpmspvMinBlocks || force_direct)) {
        // ---- a handful of products: zero, scatter with memory-side atomics, clamp / copy -----------------------------------------
        if ((e = hipMemsetAsync(s.accumulators, 0, size_t(num_rows) * (is_float ? 4 : 8), stream)) != hipSuccess) return e;
        if (total) {
            if (is_float)
                hipLaunchKernelGGL(spmspv_scatter_kernel<true>, expand_grid, block, 0, stream, indptr, row_indices, value_words, x_index, x_words, x_count,
                                   num_cols, nullptr, static_cast<float*>(s.accumulators));
            else
                hipLaunchKernelGGL(spmspv_scatter_kernel<false>, expand_grid, block, 0, stream, indptr, row_indices, value_words, x_index, x_words, x_count,
                                   num_cols, static_cast<unsigned long long*>(s.accumulators), nullptr);
        }
        const dim3 grid((num_rows + 255) / 256);
        if (is_float) hipLaunchKernelGGL(spmspv_finish_kernel<true>, grid, block, 0, stream, nullptr, static_cast<const float*>(s.accumulators), y, num_rows);
        else hipLaunchKernelGGL(spmspv_finish_kernel<false>, grid, block, 0, stream, static_cast<const unsigned long long*>(s.accumulators), nullptr, y, num_rows);
        return hipGetLastError();
    }
    // 2. expand
    if (is_float)
        hipLaunchKernelGGL(spmspv_expand_kernel<true>, expand_grid, block, 0, stream, indptr, row_indices, value_words, x_index, x_words, x_count, num_cols,
                           s.place, s.keys[0], s.vals[0]);
    else
        hipLaunchKernelGGL(spmspv_expand_kernel<false>, expand_grid, block, 0, stream, indptr, row_indices, value_words, x_index, x_words, x_count, num_cols,
                           s.place, s.keys[0], s.vals[0]);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // 3. bin by row block (the high bits of the row only; a matrix of at most one block needs no binning)
    const uint32_t* keys = s.keys[0];
    const uint32_t* vals = s.vals[0];
    const uint32_t row_bits = bits_for(num_rows);
    if (row_bits > kBlockBits) {
        size_t temp = s.temp_bytes;
        if ((e = hipcub::DeviceRadixSort::SortPairs(s.temp, temp, s.keys[0], s.keys[1], s.vals[0], s.vals[1], size_t(total), int(kBlockBits), int(row_bits),
                                                    stream)) != hipSuccess)
            return e;
        keys = s.keys[1];
        vals = s.vals[1];
    }
    // 4. accumulate per row block and write y
    const dim3 grid((num_rows + kBlockRows - 1) / kBlockRows);
    if (is_float) hipLaunchKernelGGL(spmspv_accumulate_kernel<true>, grid, dim3(1024), 0, stream, keys, vals, total, y, num_rows);
    else hipLaunchKernelGGL(spmspv_accumulate_kernel<false>, grid, dim3(1024), 0, stream, keys, vals, total, y, num_rows);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
