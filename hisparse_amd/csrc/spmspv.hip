// spmspv.hip — SpMSpV on gfx950: y = A x for a SPARSE x over a CSC matrix.  EXTENSION.
//
// The reference only stubs this operator -- the packet / pair types SPMSPV_MAT_PKT_T and IDX_VAL_T
// (spmv/libfpga/common.h:52-54) and the CSC conversion csr2csc (sw/data_loader.h:109-144); the paper (section 7) names it as the
// natural next kernel on the same datapath.  Here: one wavefront per stored x entry streams that entry's matrix column
// (row index + value word, both contiguous in CSC: two coalesced loads per 64 non-zeros), multiplies with the PE arithmetic of
// the numeric mode and adds the products to per-row accumulators in HBM.  The work is proportional to the non-zeros of the
// SELECTED columns only, which is the point of the operator; the accumulate is a device-scope atomic per product (the rows of
// different columns collide arbitrarily), so this path is for sparse x -- for a dense x the SpMV path is 10-100x faster.
//   fixed: products rounded / saturated one by one (q8_24_mul), summed in 64-bit integer accumulators, clamped once by the
//          finish pass -- bit-identical to the saturating PE sum, in any order;
//   float: one fp32 multiply per product, fp32 atomic adds (order = arrival order, like the FPGA's): tolerance parity.
#include <hip/hip_runtime.h>

#include "spmv_device.h"
#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

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
        if (col >= num_cols) continue;                       // checked on the host as well; never read out of range
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

}  // namespace

hipError_t launch_spmspv(bool is_float, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, const uint32_t* x_index,
                         const uint32_t* x_words, uint32_t x_count, uint32_t num_rows, uint32_t num_cols, void* accumulators, uint32_t* y,
                         hipStream_t stream) {
    hipError_t e = hipMemsetAsync(accumulators, 0, size_t(num_rows) * (is_float ? 4 : 8), stream);
    if (e != hipSuccess) return e;
    if (x_count) {
        const dim3 grid(std::min<uint32_t>((x_count + 3) / 4, 4096)), block(256);       // 4 wavefronts per workgroup, one x entry each
        if (is_float)
            hipLaunchKernelGGL(spmspv_scatter_kernel<true>, grid, block, 0, stream, indptr, row_indices, value_words, x_index, x_words, x_count,
                               num_cols, nullptr, static_cast<float*>(accumulators));
        else
            hipLaunchKernelGGL(spmspv_scatter_kernel<false>, grid, block, 0, stream, indptr, row_indices, value_words, x_index, x_words, x_count,
                               num_cols, static_cast<unsigned long long*>(accumulators), nullptr);
    }
    const dim3 grid((num_rows + 255) / 256), block(256);
    if (is_float) hipLaunchKernelGGL(spmspv_finish_kernel<true>, grid, block, 0, stream, nullptr, static_cast<const float*>(accumulators), y, num_rows);
    else hipLaunchKernelGGL(spmspv_finish_kernel<false>, grid, block, 0, stream, static_cast<const unsigned long long*>(accumulators), nullptr, y, num_rows);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
