// spmv_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the SpMV hot path.
//
// Replaces the FPGA dataflow  spmv_vector_loader -> spmv_sk0/1/2 -> spmv_result_drain
// (spmv/spmv_vector_loader.cpp:95-121, spmv/spmv_sk0.cpp:12-121, spmv/spmv_result_drain.cpp:10-126):
//
//   FPGA unit (reference)                                    here
//   -------------------------------------------------------  ----------------------------------------------
//   vector loader + 8 vector banks per cluster               one x tile (<= 160 KiB) per workgroup in LDS,
//     (vecbuf_access_unit.h:66-72,126-128)                     loaded once per column partition
//   CPSR_matrix_loader, 1 packet/cycle/channel               64-lane wavefronts, 16 B + 8 B coalesced loads
//     (spmv_cluster.h:34-107)                                  from re-tiled streams (stream_tiles.h)
//   shuffle 1 by col%8 + bank read (shuffle.h:380-468)       ds_read_b32 gather from the LDS tile
//   shuffle 2 by row%8 + PE accumulate (pe.h:62-81)          per-lane running sum, flushed at end-of-row
//   PE output banks, zeroed per row partition (pe.h:131-135) global accumulator, integer / fp32 atomics
//   result packer + drain (spmv_result_drain.cpp:104-113)    natural-order y written by the finalize pass
//
// Bandwidth-bound integer/fp32 gather work: no MFMA anywhere.
// Numerics: fixed point = ap_ufixed<32,8,AP_RND,AP_SAT> products summed in u64 and clamped once
// (bit-exact with the saturating PE because all terms are non-negative);
// float = separate fp32 multiply and add (no FMA contraction: -ffp-contract=off), association differs
// from the FPGA's arrival order, hence tolerance parity.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr int kThreads = kRunsPerWorkgroup;   // 1024 = 16 wavefronts: one workgroup fills a CU's LDS with the x tile
constexpr int kBatchGroups = kStepQuantum / kStepsPerGroup;  // groups (of 4 steps) loaded per prefetch batch

// mat_val * vec_val narrowed to Q8.24: exact 64-bit product, + half LSB, >> 24, saturate (pe.h:64).
__device__ __forceinline__ uint32_t q8_24_mul(uint32_t a, uint32_t b) {
    const uint64_t wide = static_cast<uint64_t>(a) * b;
    const uint64_t r = (wide >> 24) + ((wide >> 23) & 1u);
    return r > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(r);
}

template <bool kFloat>
struct RowSum;
template <>
struct RowSum<false> {  // fixed point: exact integer sum of saturated products
    using accum_t = unsigned long long;
    uint64_t v = 0;
    __device__ __forceinline__ void add(uint32_t mat, uint32_t vec, bool special) { v += q8_24_mul(special ? 0u : mat, vec); }
    __device__ __forceinline__ bool nonzero() const { return v != 0; }
    __device__ __forceinline__ void flush(accum_t* acc, uint32_t row) { atomicAdd(acc + row, static_cast<unsigned long long>(v)); }
    __device__ __forceinline__ void clear() { v = 0; }
};
template <>
struct RowSum<true> {  // float: multiply then add, as the float PEs do (pe-pob.h:63-65, pe-stall.h:52,138)
    using accum_t = float;
    float v = 0.0f;
    __device__ __forceinline__ void add(uint32_t mat, uint32_t vec, bool special) {
        const float prod = __uint_as_float(mat) * __uint_as_float(vec);
        v += special ? 0.0f : prod;
    }
    __device__ __forceinline__ bool nonzero() const { return v != 0.0f; }
    __device__ __forceinline__ void flush(accum_t* acc, uint32_t row) { atomicAdd(acc + row, v); }
    __device__ __forceinline__ void clear() { v = 0.0f; }
};

// One element: gather x from the LDS tile, accumulate, and on an end-of-row flag flush the row sum
// and step the row counter (flagged non-zero: +stride; flagged special element = ROWSET: absolute row).
// kAblate (profiling builds only, HISPARSE_ABLATE): bit 0 = no atomics, bit 1 = no LDS gather, bit 2 = no arithmetic.
template <bool kFloat, int kAblate>
__device__ __forceinline__ void consume(uint32_t col, uint32_t val, bool flagged, const uint32_t* xs, uint32_t col_clamp,
                                        uint32_t stride, RowSum<kFloat>& sum, uint32_t& row,
                                        typename RowSum<kFloat>::accum_t* accum) {
    const bool special = col == kSpecialCol;
    const uint32_t xv = (kAblate & 2) ? col : xs[min(col, col_clamp)];
    if (kAblate & 4) { asm volatile("" ::"v"(xv), "v"(val)); } else { sum.add(val, xv, special); }
    if (flagged) {
        if (kAblate & 1) { asm volatile("" ::"v"(row)); } else if (sum.nonzero()) sum.flush(accum, row);
        sum.clear();
        row = special ? val : row + stride;
    }
}

template <bool kFloat, int kAblate>
__global__ __launch_bounds__(kThreads) void spmv_stream_kernel(const uint8_t* __restrict__ image, const Piece* __restrict__ pieces,
                                                                const uint32_t* __restrict__ wg_first,
                                                                const uint32_t* __restrict__ x,
                                                                typename RowSum<kFloat>::accum_t* __restrict__ accum,
                                                                uint32_t num_cols, uint32_t tile_cols, uint32_t stride,
                                                                int32_t row_part_filter) {
    extern __shared__ __attribute__((aligned(16))) uint32_t xs[];
    const uint32_t lane = threadIdx.x & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWaveLanes);
    // Workgroup b runs on XCD b % 8 (observed dispatch order); give each XCD a contiguous range of
    // logical workgroups so the workgroups that share an x tile share an L2.  Speed only.
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    uint32_t resident_tile = 0xffffffffu;
    const uint32_t p_end = wg_first[wg + 1];
    for (uint32_t p = wg_first[wg]; p < p_end; ++p) {
        const Piece piece = pieces[p];
        if (row_part_filter >= 0 && piece.row_part != static_cast<uint32_t>(row_part_filter)) continue;
        const uint32_t cols_here = min(tile_cols, num_cols - piece.col_tile * tile_cols);
        if (piece.col_tile != resident_tile) {
            __syncthreads();  // everyone is done with the previous tile
            const uint4* src = reinterpret_cast<const uint4*>(x + static_cast<size_t>(piece.col_tile) * tile_cols);
            for (uint32_t i = threadIdx.x; i < cols_here / 4; i += kThreads) reinterpret_cast<uint4*>(xs)[i] = src[i];
            __syncthreads();
            resident_tile = piece.col_tile;
        }
        const uint32_t steps = piece.steps;
        const uint8_t* chunk = image + piece.offset + static_cast<uint64_t>(wave) * (kChunkHeaderBytes + static_cast<uint64_t>(steps) * 8 +
                                                                                    static_cast<uint64_t>(steps / kStepsPerGroup) * kGroupBytes);
        uint32_t row = reinterpret_cast<const uint32_t*>(chunk)[lane];
        const uint64_t* flags = reinterpret_cast<const uint64_t*>(chunk + kChunkHeaderBytes);
        const uint8_t* groups = chunk + kChunkHeaderBytes + static_cast<uint64_t>(steps) * 8;
        const uint32_t col_clamp = cols_here - 1;
        const uint32_t batches = steps / (kStepsPerGroup * kBatchGroups);

        uint2 cnext[kBatchGroups];
        uint4 vnext[kBatchGroups];
#pragma unroll
        for (int j = 0; j < kBatchGroups; ++j) {
            cnext[j] = reinterpret_cast<const uint2*>(groups + j * kGroupBytes)[lane];
            vnext[j] = reinterpret_cast<const uint4*>(groups + j * kGroupBytes + kGroupColsBytes)[lane];
        }
        RowSum<kFloat> sum;
        for (uint32_t b = 0; b < batches; ++b) {
            uint2 c[kBatchGroups];
            uint4 v[kBatchGroups];
#pragma unroll
            for (int j = 0; j < kBatchGroups; ++j) { c[j] = cnext[j]; v[j] = vnext[j]; }
            if (b + 1 < batches) {
                const uint8_t* nxt = groups + static_cast<uint64_t>(b + 1) * kBatchGroups * kGroupBytes;
#pragma unroll
                for (int j = 0; j < kBatchGroups; ++j) {
                    cnext[j] = reinterpret_cast<const uint2*>(nxt + j * kGroupBytes)[lane];
                    vnext[j] = reinterpret_cast<const uint4*>(nxt + j * kGroupBytes + kGroupColsBytes)[lane];
                }
            }
            const uint64_t* f = flags + static_cast<uint64_t>(b) * kBatchGroups * kStepsPerGroup;
#pragma unroll
            for (int j = 0; j < kBatchGroups; ++j) {
                const uint64_t f0 = f[j * 4 + 0], f1 = f[j * 4 + 1], f2 = f[j * 4 + 2], f3 = f[j * 4 + 3];
                consume<kFloat, kAblate>(c[j].x & 0xffffu, v[j].x, (f0 >> lane) & 1u, xs, col_clamp, stride, sum, row, accum);
                consume<kFloat, kAblate>(c[j].x >> 16, v[j].y, (f1 >> lane) & 1u, xs, col_clamp, stride, sum, row, accum);
                consume<kFloat, kAblate>(c[j].y & 0xffffu, v[j].z, (f2 >> lane) & 1u, xs, col_clamp, stride, sum, row, accum);
                consume<kFloat, kAblate>(c[j].y >> 16, v[j].w, (f3 >> lane) & 1u, xs, col_clamp, stride, sum, row, accum);
            }
        }
        if (!(kAblate & 1) && sum.nonzero()) sum.flush(accum, row);  // the run ended inside a row: hand the partial sum over
    }
}

__global__ __launch_bounds__(256) void finalize_fixed_kernel(uint64_t* __restrict__ accum, uint32_t* __restrict__ y, uint32_t row_lo,
                                                             uint32_t row_hi) {
    const uint32_t r = row_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= row_hi) return;
    const uint64_t s = accum[r];
    accum[r] = 0;  // ready for the next SpMV (the PEs zero their banks at the start of every launch, pe.h:131-135)
    y[r] = s > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(s);  // AP_SAT of the running sum (pe.h:72)
}

}  // namespace

template <bool kFloat, int kAblate>
hipError_t configure_one(uint32_t lds_bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_stream_kernel<kFloat, kAblate>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
}

// Profiling aid: HISPARSE_ABLATE=<bits> launches a variant with parts of the work removed (results are wrong).
int ablation() {
    static const int v = [] { const char* e = std::getenv("HISPARSE_ABLATE"); return e ? std::atoi(e) : 0; }();
    return v;
}

hipError_t configure_spmv_kernels(uint32_t lds_bytes) {
    hipError_t e;
    if ((e = configure_one<false, 0>(lds_bytes)) != hipSuccess) return e;
    if ((e = configure_one<true, 0>(lds_bytes)) != hipSuccess) return e;
    if ((e = configure_one<false, 1>(lds_bytes)) != hipSuccess) return e;
    if ((e = configure_one<false, 2>(lds_bytes)) != hipSuccess) return e;
    if ((e = configure_one<false, 3>(lds_bytes)) != hipSuccess) return e;
    if ((e = configure_one<false, 7>(lds_bytes)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_spmv_stream(bool is_float, const SpmvLaunch& a, hipStream_t stream) {
    if (a.num_workgroups == 0) return hipSuccess;
    const dim3 grid(a.num_workgroups), block(kThreads);
#define HS_LAUNCH(FLOAT, ABL, T)                                                                                          \
    hipLaunchKernelGGL((spmv_stream_kernel<FLOAT, ABL>), grid, block, a.lds_bytes, stream, a.image, a.pieces, a.wg_first, a.x, \
                       static_cast<T*>(a.accum), a.num_cols, a.tile_cols, a.row_stride, a.row_part_filter)
    if (is_float) {
        HS_LAUNCH(true, 0, float);
    } else {
        switch (ablation()) {
            case 1: HS_LAUNCH(false, 1, unsigned long long); break;
            case 2: HS_LAUNCH(false, 2, unsigned long long); break;
            case 3: HS_LAUNCH(false, 3, unsigned long long); break;
            case 7: HS_LAUNCH(false, 7, unsigned long long); break;
            default: HS_LAUNCH(false, 0, unsigned long long); break;
        }
    }
#undef HS_LAUNCH
    return hipGetLastError();
}

hipError_t launch_finalize_fixed(uint64_t* accum, uint32_t* y, uint32_t row_lo, uint32_t row_hi, hipStream_t stream) {
    if (row_hi <= row_lo) return hipSuccess;
    const uint32_t n = row_hi - row_lo;
    hipLaunchKernelGGL(finalize_fixed_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, y, row_lo, row_hi);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
