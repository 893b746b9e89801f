// hisparse/q8_24.h — unsigned Q8.24 fixed point with round-half-up quantisation and saturation.
//
// The reference's fixed-point value type is `ap_ufixed<32, 8, AP_RND, AP_SAT>`
// (spmv/libfpga/common.h:35-38) from the Vitis HLS 2020.2 `ap_fixed.h` header, which is NOT part of
// the reference checkout.  This file restates the documented semantics of that type for exactly the
// operations the hot path performs:
//   * construction from float            sw/data_loader.h:80 (std::copy float -> VAL_T),
//                                        sw/benchmark.cpp:210 (VAL_T(vector_f[..]))
//   * construction from an integer       sw/data_formatter.h:73,158 (marker value = row count)
//   * a * b narrowed back to VAL_T       spmv/libfpga/pe.h:64
//   * q + incr narrowed back to VAL_T    spmv/libfpga/pe.h:72
//   * bit range (31, 32-IBITS)           spmv/libfpga/spmv_cluster.h:82 (marker decode)
//   * conversion to float                spmv_csim/csim.cpp:172
// Documented behaviour used: the product of two ap_ufixed<32,8> is the exact ap_ufixed<64,16>; the
// sum is the exact ap_ufixed<33,9>; assignment to the narrower type applies AP_RND (add half an LSB
// of the destination, then truncate: "round to plus infinity") and then AP_SAT (clamp to
// [0, 2^32-1] raw).  Negative inputs clamp to 0 because the type is unsigned.
// The reference's own tests never exercise rounding or saturation (all-ones matrices, x in {0,1}),
// so these two rules are pinned only by this restatement and the oracle's identical one
// (oracle/cpu_ref.c) — see DESIGN.md "parity pins".
#ifndef HISPARSE_Q8_24_H_
#define HISPARSE_Q8_24_H_

#include <cmath>
#include <cstdint>

namespace hisparse {

constexpr uint32_t Q8_24_MAX_RAW = 0xffffffffu;
constexpr uint32_t Q8_24_ONE_RAW = 1u << 24;

// float/double -> raw Q8.24.  d * 2^24 is exact in double for every finite float d < 256.
inline uint32_t q8_24_raw_from_double(double d) {
    if (!(d > 0.0)) return 0u;  // negatives, -0, +0 and NaN all land on 0
    double scaled = std::floor(d * 16777216.0 + 0.5);
    if (scaled >= 4294967296.0) return Q8_24_MAX_RAW;
    return static_cast<uint32_t>(scaled);
}
// integer n -> raw of n.0, saturating (n >= 256 gives 0xffffffff).
inline uint32_t q8_24_raw_from_uint(uint64_t n) {
    return n >= 256u ? Q8_24_MAX_RAW : static_cast<uint32_t>(n << 24);
}
// (a * b) narrowed: AP_RND then AP_SAT.
inline uint32_t q8_24_mul_raw(uint32_t a, uint32_t b) {
    uint64_t wide = static_cast<uint64_t>(a) * b;  // Q16.48, < 2^64
    uint64_t r = (wide >> 24) + ((wide >> 23) & 1u);  // == (wide + 2^23) >> 24 without overflow
    return r > Q8_24_MAX_RAW ? Q8_24_MAX_RAW : static_cast<uint32_t>(r);
}
// (a + b) narrowed: no fractional bits are dropped, only AP_SAT applies.
inline uint32_t q8_24_add_raw(uint32_t a, uint32_t b) {
    uint64_t s = static_cast<uint64_t>(a) + b;
    return s > Q8_24_MAX_RAW ? Q8_24_MAX_RAW : static_cast<uint32_t>(s);
}
inline float q8_24_raw_to_float(uint32_t raw) {
    return static_cast<float>(static_cast<double>(raw) / 16777216.0);
}

// Value-type wrapper so the formatter templates can be instantiated with it the way the
// reference instantiates them with VAL_T.
class q8_24 {
 public:
    uint32_t raw;
    q8_24() : raw(0) {}
    q8_24(float f) : raw(q8_24_raw_from_double(f)) {}
    q8_24(double d) : raw(q8_24_raw_from_double(d)) {}
    q8_24(uint32_t n) : raw(q8_24_raw_from_uint(n)) {}
    q8_24(int n) : raw(n <= 0 ? 0u : q8_24_raw_from_uint(static_cast<uint64_t>(n))) {}
    static q8_24 from_raw(uint32_t r) {
        q8_24 q;
        q.raw = r;
        return q;
    }
    explicit operator float() const { return q8_24_raw_to_float(raw); }
    // bits (31, 32-IBITS): the integer part, used by the marker decode.
    uint32_t int_bits() const { return raw >> 24; }
    friend q8_24 operator*(q8_24 a, q8_24 b) { return from_raw(q8_24_mul_raw(a.raw, b.raw)); }
    friend q8_24 operator+(q8_24 a, q8_24 b) { return from_raw(q8_24_add_raw(a.raw, b.raw)); }
    friend bool operator==(q8_24 a, q8_24 b) { return a.raw == b.raw; }
    friend bool operator!=(q8_24 a, q8_24 b) { return a.raw != b.raw; }
};
static_assert(sizeof(q8_24) == 4, "q8_24 must be a bare 32-bit word");

}  // namespace hisparse

#endif  // HISPARSE_Q8_24_H_
