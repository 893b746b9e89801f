// spmv_device.h — device-side pieces shared by the gfx950 kernels (spmv_kernels.hip: PAIRS / DELTA element streams,
// spmv_bitmap.hip: BITMAP rows): the Q8.24 product, the per-mode accumulator types, the wavefront sum.
#ifndef HISPARSE_SPMV_DEVICE_H_
#define HISPARSE_SPMV_DEVICE_H_

#include <hip/hip_runtime.h>

#include "stream_tiles.h"

namespace hisparse {
namespace dev {
namespace {

// mat_val * vec_val narrowed to Q8.24: exact 64-bit product, + half LSB, >> 24, saturate (pe.h:64).
__device__ __forceinline__ uint32_t q8_24_mul(uint32_t a, uint32_t b) {
    // min((a*b + 2^23) >> 24, 2^32-1): the rounding constant rides in the multiply-add (a*b + 2^23 < 2^64), the shift
    // is one v_alignbit on the 64-bit register pair, and "result >= 2^32" is "top byte of the high word != 0".
    const uint64_t wide = static_cast<uint64_t>(a) * b + 0x800000ull;
    uint32_t hi = static_cast<uint32_t>(wide >> 32);
    const uint32_t lo = static_cast<uint32_t>(wide);
    asm("" : "+v"(hi));   // keeps hipcc from re-deriving the overflow test from a second, unrounded 64-bit multiply
    const uint32_t r = __builtin_amdgcn_alignbit(hi, lo, 24);
    return (hi >> 24) ? 0xffffffffu : r;
}

// Matrix descriptors (blocks, units) never change while the kernel runs: reading them through the CONSTANT address
// space makes every uniform read a scalar-cache load, whatever hipcc can or cannot prove after the inline asm below
// (through a plain pointer it fell back to VECTOR loads of the uniform address: a vmcnt(0) drain per unit boundary).
typedef const __attribute__((address_space(4))) Unit* UnitTable;
typedef const __attribute__((address_space(4))) Block* BlockTable;

template <bool kFloat>
struct Rows;
template <>
struct Rows<false> {
    using acc_t = unsigned long long;   // LDS accumulator
    using prod_t = unsigned long long;
    static __device__ __forceinline__ prod_t product(uint32_t mat, uint32_t vec) { return q8_24_mul(mat, vec); }
    static __device__ __forceinline__ void add(acc_t* ys, uint32_t row, prod_t p) { atomicAdd(ys + row, p); }   // ds_add_u64
    static __device__ __forceinline__ uint32_t finish(acc_t s) { return s > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(s); }  // AP_SAT (pe.h:72)
    using sum_t = unsigned long long;   // a lane's private sum over a row run
    static __device__ __forceinline__ sum_t widen(prod_t p) { return p; }
    static __device__ __forceinline__ void add_sum(acc_t* ys, uint32_t row, sum_t v) { atomicAdd(ys + row, v); }
};
template <>
struct Rows<true> {
    using acc_t = double;    // LDS accumulator: ds_add_f64 is ~9x faster than ds_add_f32 here (stream_tiles.h)
    using prod_t = float;
    // multiply in float like the float PEs (pe-pob.h:63-65, pe-stall.h:52,138); the products are then summed in double
    // and rounded to float once per row (and per column slice) -- closer to the exact sum than the PEs' float running sum
    static __device__ __forceinline__ prod_t product(uint32_t mat, uint32_t vec) { return __uint_as_float(mat) * __uint_as_float(vec); }
    static __device__ __forceinline__ void add(acc_t* ys, uint32_t row, prod_t p) { atomicAdd(ys + row, static_cast<double>(p)); }   // ds_add_f64
    static __device__ __forceinline__ uint32_t finish(acc_t s) { return __float_as_uint(static_cast<float>(s)); }
    using sum_t = double;               // a lane's private sum over a row run
    static __device__ __forceinline__ sum_t widen(prod_t p) { return static_cast<double>(p); }
    static __device__ __forceinline__ void add_sum(acc_t* ys, uint32_t row, sum_t v) { atomicAdd(ys + row, v); }
};

// Sum over the 64 lanes of a wavefront (result valid in every lane).
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWaveLanes);
    return v;
}

// Sum over the 64 lanes with DPP moves only (no LDS traffic: __shfl_* compile to ds_bpermute here, ~18 LDS instructions for a 64-bit
// value); the total arrives in lane 63 only.  row_shr:1,2,4,8 -> row_bcast:15 (rows 1,3) -> row_bcast:31 (rows 2,3): the inclusive scan
// LLVM's own atomic optimizer builds on gfx9.  Lanes without a source add 0 (old = identity, bound_ctrl off).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_from(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), kCtrl, kRowMask, 0xf, false));
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ unsigned long long dpp_from(unsigned long long v) {
    const uint32_t lo = dpp_from<kCtrl, kRowMask>(static_cast<uint32_t>(v)), hi = dpp_from<kCtrl, kRowMask>(static_cast<uint32_t>(v >> 32));
    return static_cast<unsigned long long>(hi) << 32 | lo;
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_from(double v) {      // identity 0.0 = all-zero bits
    return __longlong_as_double(static_cast<long long>(dpp_from<kCtrl, kRowMask>(static_cast<unsigned long long>(__double_as_longlong(v)))));
}
// (the same sequence leaves the INCLUSIVE prefix sum in every lane: wave_inclusive_scan)
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v);
template <typename T>
__device__ __forceinline__ T wave_total_in_lane63(T v) { return wave_inclusive_scan(v); }
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
    v += dpp_from<0x111, 0xf>(v);      // row_shr:1
    v += dpp_from<0x112, 0xf>(v);      // row_shr:2
    v += dpp_from<0x114, 0xf>(v);      // row_shr:4
    v += dpp_from<0x118, 0xf>(v);      // row_shr:8   -> lane 15 of every row of 16 holds its row's sum
    v += dpp_from<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3
    v += dpp_from<0x143, 0xc>(v);      // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}

// Segmented inclusive scan over the 64 lanes, DPP only: `row` never decreases from lane to lane (sorted elements), lanes of equal row form
// contiguous runs, and after the call the LAST lane of every run (is_last) holds the run's sum -- one conflict-free LDS add per row run
// instead of up to 64 lanes colliding on one accumulator (~2 clocks per extra lane: 12 of 17 us on a mouse_gene slab).  Kogge-Stone inside
// the rows of 16 lanes (row_shr 1, 2, 4, 8), then row_bcast:15 / row_bcast:31 across them; a lane adds what arrives only when it comes from
// its own row run (the source's row equals its own: with sorted rows everything in between then does too).  Lanes without a source see
// row 0xffffffff.
template <int kCtrl, int kRowMask, typename T>
__device__ __forceinline__ void segmented_step(T& v, uint32_t row) {
    const uint32_t from_row = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(row), kCtrl, kRowMask, 0xf, false));
    const T arriving = dpp_from<kCtrl, kRowMask>(v);
    if (from_row == row) v += arriving;
}
template <typename T>
__device__ __forceinline__ T segmented_run_sums(T v, uint32_t row, bool& is_last) {
    segmented_step<0x111, 0xf>(v, row);
    segmented_step<0x112, 0xf>(v, row);
    segmented_step<0x114, 0xf>(v, row);
    segmented_step<0x118, 0xf>(v, row);
    segmented_step<0x142, 0xa>(v, row);
    segmented_step<0x143, 0xc>(v, row);
    const uint32_t next_row = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(-1, static_cast<int>(row), 0x130, 0xf, 0xf, false));   // wave_shl:1 (lane 63: none)
    is_last = next_row != row;
    return v;
}

}  // namespace
}  // namespace dev
}  // namespace hisparse

#endif  // HISPARSE_SPMV_DEVICE_H_
