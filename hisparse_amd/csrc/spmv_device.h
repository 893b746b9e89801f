// spmv_device.h — device-side pieces shared by the gfx950 kernels (spmv_kernels.hip: PAIRS / DELTA element streams,
// spmv_bitmap.hip: BITMAP rows): the Q8.24 product, the per-mode accumulator types, the wavefront sum.
#ifndef HISPARSE_SPMV_DEVICE_H_
#define HISPARSE_SPMV_DEVICE_H_

#include <hip/hip_runtime.h>

#include <type_traits>

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
    using lane_t = unsigned long long;  // DELTA dense rows: a lane's sum over its run inside ONE unit (exact)
    static __device__ __forceinline__ void add_lane(acc_t* ys, uint32_t row, lane_t v) { atomicAdd(ys + row, v); }
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
    // DELTA dense rows: a lane's sum over its run inside ONE unit -- a hundred products at most -- is taken in fp32 like the float PEs'
    // running sums (pe-pob.h:62-71, pe-stall.h:137-141) and joins the row's DOUBLE accumulator when the row or the unit changes: one
    // full-rate v_add_f32 per element instead of a conversion and a half-rate v_add_f64
    using lane_t = float;
    static __device__ __forceinline__ void add_lane(acc_t* ys, uint32_t row, lane_t v) { atomicAdd(ys + row, static_cast<double>(v)); }
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


// ---- column-sliced plans: the combine pass of step k carried into the SpMV kernel of step k + 1 (round 5) ---------------------------
// A column-sliced step is SpMV kernel (per-slice partial rows) + combine_slices_kernel (y = their sum): a second launch of 2.5-5 us for
// 2-10 MB of traffic.  Four attempts to fold it into the SAME kernel lost to the cross-workgroup hand-over they need at the kernel's tail
// (DESIGN.md).  This one needs none: when hs_run follows hs_run, the partial rows of step k are complete and visible when the kernel of
// step k + 1 STARTS (a kernel boundary lies between them), so that kernel's workgroups -- each a stripe of the rows, by workgroup index --
// add them up and write y(k) first, while their own descriptor and first stream loads travel; step k + 1 writes the OTHER set of partial
// vectors.  The host launches the stand-alone combine only for the LAST step of such a run (hs_api.cpp: flush_combine).  Sums in slice
// order from 0.0f / saturating adds, exactly as combine_slices_kernel: the same words.
struct CarriedCombine {
    const uint32_t* partial = nullptr;      // the previous step's partial vectors (nullptr: nothing to carry)
    uint32_t* y = nullptr;                  // where that step's result goes
    uint32_t num_rows = 0;                  // stride between the partial vectors, and the rows to combine (a multiple of 128)
    uint32_t slices = 0;
};

template <typename Launch>
inline CarriedCombine carried(const Launch& a) {
    CarriedCombine c;
    if (a.carry_partial && a.carry_slices > 1 && a.row_part_filter < 0) {
        c.partial = a.carry_partial;
        c.y = a.carry_y;
        c.num_rows = a.carry_rows;
        c.slices = a.carry_slices;
    }
    return c;
}

template <bool kFloat, int kBlockThreads>
__device__ __forceinline__ void carried_combine(const CarriedCombine& c, uint32_t wg, uint32_t groups, uint32_t tid) {
    using Sum = typename std::conditional<kFloat, float, uint32_t>::type;
    const uint32_t quads = c.num_rows / 4u, per = (quads + groups - 1u) / groups;
    const uint32_t q_end = min(quads, (wg + 1u) * per);
    for (uint32_t q = wg * per + tid; q < q_end; q += kBlockThreads) {
        const uint4* p = reinterpret_cast<const uint4*>(c.partial) + q;
        Sum s[4] = {0, 0, 0, 0};
        for (uint32_t k0 = 0; k0 < c.slices; k0 += 4u) {      // at most four 16-byte loads in flight per thread (the row-block kernels' register budget)
            uint4 v[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) v[k] = p[static_cast<size_t>(min(k0 + k, c.slices - 1u)) * quads];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) {
                if (k0 + k >= c.slices) break;
                const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (kFloat) s[j] += __uint_as_float(w[j]);
                    else s[j] = __builtin_elementwise_add_sat(s[j], w[j]);      // saturating adds of non-negative terms: min(sum, 2^32 - 1) in any order
                }
            }
        }
        uint4 out;
        if constexpr (kFloat) out = make_uint4(__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3]));
        else out = make_uint4(s[0], s[1], s[2], s[3]);
        reinterpret_cast<uint4*>(c.y)[q] = out;
    }
}

}  // namespace
}  // namespace dev
}  // namespace hisparse

#endif  // HISPARSE_SPMV_DEVICE_H_
