// hisparse/common.h — overlay constants and wire types of the SpMV hot path.
//
// Restates (does not copy) the configuration block of the reference kernel library:
//   spmv/libfpga/common.h:30-50,162-179     (fixed-point build)
//   spmv-fp/libfpga/common.h:18-51,174-200  (float_pob / float_stall builds)
// In the reference these are compile-time constants selected by -DFP_POB / -DFP_STALL; here the
// numeric mode ("impl") and both bank sizes are RUNTIME values so one library serves all three.
#ifndef HISPARSE_COMMON_H_
#define HISPARSE_COMMON_H_

#include <cstdint>
#include <cstring>

#define IDX_MARKER 0xffffffffu  // end-of-row marker in the index slot (spmv/libfpga/common.h:8)

namespace hisparse {

// 8 PEs per cluster / 8 words per packet (common.h:30).
constexpr unsigned PACK_SIZE = 8;
// 4 + 6 + 6 clusters, one per matrix HBM channel (common.h:173-176).
constexpr unsigned SK0_CLUSTER = 4;
constexpr unsigned SK1_CLUSTER = 6;
constexpr unsigned SK2_CLUSTER = 6;
constexpr unsigned NUM_HBM_CHANNELS = SK0_CLUSTER + SK1_CLUSTER + SK2_CLUSTER;
// Q8.24 split of the fixed-point value type (common.h:35-38).
constexpr unsigned IBITS = 8;
constexpr unsigned FBITS = 32 - IBITS;

// Numeric mode == the reference's IMPL make variable (sw/Makefile:2-12).
enum Impl : int {
    IMPL_FIXED = 0,        // ap_ufixed<32,8,AP_RND,AP_SAT>, INTERLEAVE_FACTOR 1, OB bank 8192
    IMPL_FLOAT_POB = 1,    // fp32, partial output buffers,  INTERLEAVE_FACTOR 1, OB bank 1024
    IMPL_FLOAT_STALL = 2,  // fp32, stall queue + row interleaving, INTERLEAVE_FACTOR 8, OB bank 8192
};

inline bool impl_valid(int impl) { return impl >= IMPL_FIXED && impl <= IMPL_FLOAT_STALL; }
inline bool impl_is_float(int impl) { return impl != IMPL_FIXED; }
// INTERLEAVE_FACTOR (spmv/libfpga/common.h:169, spmv-fp/libfpga/common.h:181,187).
inline unsigned impl_interleave_factor(int impl) { return impl == IMPL_FLOAT_STALL ? 8u : 1u; }
// Default per-bank sizes in words: OB_BANK_SIZE / VB_BANK_SIZE of the shipped bitstreams
// (common.h:164-165; spmv-fp common.h:180,186,190; sw/bm.sh:21-27 passes v=4, o=8 or 1).
inline unsigned impl_default_ob_bank(int impl) { return impl == IMPL_FLOAT_POB ? 1024u : 8192u; }
inline unsigned impl_default_vb_bank(int) { return 4096u; }

// Wire packets, exactly as they cross the kernel-memory boundary (common.h:44-50).
// Values are carried as raw 32-bit patterns: Q8.24 words in fixed mode, IEEE-754 bits otherwise.
struct PackedWord {
    uint32_t data[PACK_SIZE];
};
struct MatPkt {  // SPMV_MAT_PKT_T: 8 column indices THEN 8 values = 64 bytes
    PackedWord indices;
    PackedWord vals;
};
static_assert(sizeof(PackedWord) == 32, "vector/result packet is 32 bytes");
static_assert(sizeof(MatPkt) == 64, "matrix packet is 64 bytes");

inline uint32_t f32_bits(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
inline float bits_f32(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// Geometry every driver derives the same way (sw/benchmark.cpp:112-117, spmv_csim/csim.cpp:216-219).
struct Geometry {
    int impl;
    unsigned interleave;        // F
    unsigned ob_bank, vb_bank;  // words per bank
    uint64_t logical_ob;        // rows per row partition    = ob_bank * P * C
    uint64_t logical_vb;        // columns per col partition = vb_bank * P
    unsigned virtual_channels;  // C * F
    unsigned row_divisor;       // P * C * F: rows are padded to a multiple of this
};
inline Geometry make_geometry(int impl, unsigned ob_bank, unsigned vb_bank) {
    Geometry g;
    g.impl = impl;
    g.interleave = impl_interleave_factor(impl);
    g.ob_bank = ob_bank;
    g.vb_bank = vb_bank;
    g.logical_ob = uint64_t(ob_bank) * PACK_SIZE * NUM_HBM_CHANNELS;
    g.logical_vb = uint64_t(vb_bank) * PACK_SIZE;
    g.virtual_channels = NUM_HBM_CHANNELS * g.interleave;
    g.row_divisor = PACK_SIZE * NUM_HBM_CHANNELS * g.interleave;
    return g;
}

}  // namespace hisparse

#endif  // HISPARSE_COMMON_H_
