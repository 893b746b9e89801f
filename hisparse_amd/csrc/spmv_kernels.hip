// spmv_kernels.hip — the gfx950 (MI355X, CDNA4) kernel of the SpMV hot path.
//
// Replaces the FPGA dataflow  spmv_vector_loader -> spmv_sk0/1/2 -> spmv_result_drain
// (spmv/spmv_vector_loader.cpp:95-121, spmv/spmv_sk0.cpp:12-121, spmv/spmv_result_drain.cpp:10-126):
//
//   FPGA unit (reference)                                     here, per 1024-thread workgroup
//   --------------------------------------------------------  --------------------------------------------------
//   spmv_cluster: owns rows, PE output banks zeroed per        owns a row block; 64-bit row accumulators in LDS (integer
//     launch, dumped at the end (pe.h:121-178)                   sums / double sums of the fp32 products), zeroed per
//                                                                block, written once
//   vector loader + vecbuf_access_unit: x partition in 8       2 LOADER wavefronts refill a ring of 2-4 x sub-tile buffers
//     banks, writer/reader alternate per partition               (32 KiB each) with LDS-DMA, up to 3 sub-tiles ahead, while ...
//     (vecbuf_access_unit.h:146-163)
//   CPSR_matrix_loader: one 64-byte packet per cycle           ... 14 CONSUMER wavefronts stream coalesced 8-byte (PAIRS) or
//     (spmv_cluster.h:73-98)                                      6-byte (DELTA) elements, 8 steps in flight per wavefront,
//                                                                parked in accumulator registers until a counted wait
//   shuffle 1 by col%8 + bank read (shuffle.h, vecbuf :126)    ds_read_b32 gather from the LDS sub-tile
//   shuffle 2 by row%8 + PE accumulate (pe.h:62-81)            ds_add_u64 / ds_add_f64 into the row accumulators
//   result packer + drain, natural row order                   coalesced y store, AP_SAT clamp / fp32 rounding applied once
//     (spmv_result_drain.cpp:104-113)
//
// One barrier per (row block, x sub-tile) hands the freshly filled buffer to the consumers.
// No global atomics: y is complete when the kernel ends, or -- for column-sliced matrices -- after the small
// combine_slices_kernel that adds the per-slice partial results.  Bandwidth-bound integer / fp32 gather work: no MFMA.
// Numerics: fixed point = ap_ufixed<32,8,AP_RND,AP_SAT> products summed exactly in 64 bits and clamped
// once (bit-exact with the saturating PE because every term is non-negative);
// float = one fp32 multiply per product (no FMA contraction: -ffp-contract=off), summed in double and rounded to fp32
// once per row and column slice; the order of the adds differs from the FPGA's arrival order anyway, hence tolerance parity.
// Data layout and the two stream formats: stream_tiles.h.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include "spmv_kernels.h"
#include "spmv_device.h"

namespace hisparse {
namespace dev {

namespace {

constexpr int kThreads = kWaveLanes * kWavesPerWorkgroup;       // 1024
constexpr int kLightThreads = 256;                               // the LIGHT kernel's workgroup (up to four per CU)
#ifndef HS_LIGHT_BATCH
#define HS_LIGHT_BATCH 4      // round 5 (profiles/r05_sweep_ring_depth.txt): 16: 9.4 us, 8: 5.9-6.1, 4: 5.65, 3: 5.7, 2: 5.8 on transformer-95; the same layer at 10 % density
#endif                        // (LIGHT forced) 11.8 / 8.5-9.1 / 7.35 / 7.6 / 8.0 -- fewer chunks in flight leave the gathered lines of x in the L1
constexpr int kLightBatch = HS_LIGHT_BATCH;                      // chunks in flight per wavefront
constexpr uint32_t kBufBytes = kSubTileCols * 4u;               // one x buffer of the LDS ring (32 KiB)

// Copy one x sub-tile into an LDS buffer with kStride cooperating threads (t = 0 .. kStride-1).
// Completely branch-free: out-of-range threads re-copy the last 16 bytes (same data to the same place), so all
// kIters loads are in flight before the first LDS write and the L2 latency is paid once per sub-tile.
template <int kIters, int kStride>
__device__ __forceinline__ void fill_x(uint32_t* dst, const uint32_t* __restrict__ x, uint32_t col0, uint32_t ncols, uint32_t t) {
    const uint4* src = reinterpret_cast<const uint4*>(x + col0);
    uint4* d = reinterpret_cast<uint4*>(dst);
    const uint32_t n4 = ncols / 4;   // >= 2: ncols is a positive multiple of 8
    uint4 r[kIters];
#pragma unroll
    for (int j = 0; j < kIters; ++j) r[j] = src[min(t + j * kStride, n4 - 1)];
#pragma unroll
    for (int j = 0; j < kIters; ++j) d[min(t + j * kStride, n4 - 1)] = r[j];
}

// The loader wavefronts' refill: LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR round trip).
// The LDS destination of one instruction is wave-uniform base + lane * 16, which is exactly the linear sub-tile.
// Always kDmaPerFill instructions per wavefront (columns past `ncols` re-read the sub-tile's last 16 bytes into LDS
// words nobody reads), so the landing of a given refill can be awaited with a counted s_waitcnt.
constexpr uint32_t kDmaCols = kWaveLanes * 4;                                  // 256 columns = 1 KiB per instruction
constexpr uint32_t kDmaPerFill = kSubTileCols / kDmaCols / kLoaderWaves;       // 8 instructions per wavefront per refill
__device__ __forceinline__ void dma_fill_x(uint32_t* dst, const uint32_t* __restrict__ x, uint32_t col0, uint32_t ncols, uint32_t w,
                                           uint32_t lane) {
#pragma unroll
    for (uint32_t j = 0; j < kDmaPerFill; ++j) {
        const uint32_t c = (w * kDmaPerFill + j) * kDmaCols;                   // wave-uniform first column of this instruction
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) uint32_t*)(x + col0 + min(c + lane * 4, ncols - 4)),
                                         (__attribute__((address_space(3))) uint32_t*)(dst + c), 16, 0, 0);
    }
}
// "at most kFills refills are still in flight" (vmcnt retires in order; the loader wavefronts issue nothing else).
// (Round 2 tried touching the NEXT sub-tile into this XCD's L2 ahead of its refill -- one dword per 128-byte line, parked in an accumulator
// register -- because the row blocks of an XCD walk the sub-tiles in the same order at about the same time and every refill misses L2 at the
// same moment: it shifted the loaders' wait into the barrier and left the total; removed.)
template <int kFills>
__device__ __forceinline__ void dma_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFills * kDmaPerFill) : "memory");
}

// The element stream is loaded with hand-placed instructions: hipcc's s_waitcnt placement degrades to vmcnt(0)
// (a full drain per element) in the unit/barrier control flow below, which would serialise the HBM stream.
// Rules kept here:
//   * exactly ONE issue<K>() per step and no other vector-memory instruction inside the consumer loop, so "the step
//     issued kDepth steps ago has landed" is exactly a counted s_waitcnt vmcnt (vmcnt retires in order);
//   * data in flight lives in ACCUMULATOR registers (a0..a31), which hipcc never allocates in this kernel, and only
//     becomes a compiler-visible value inside take<K>(), AFTER the wait.  With ordinary asm outputs the compiler is free
//     to copy a load's destination register before the wait (it did: a v_mov of not-yet-landed data, one record in a
//     few million wrong);
//   * the stream base is a loop-invariant SGPR pair, the per-step address a 32-bit vector offset (no 64-bit vector
//     address arithmetic per step); s_nop 4 keeps 5 wait states between any scalar write of the base and its use;
//   * `nt`: every element is read exactly once per SpMV, so it should not displace x in the L2 (measured: -8 %).
//   * round 6: ... unless the image FITS the 256 MiB Infinity Cache, i.e. stays there from one SpMV to the next, AND the blocks walk several (row range x
//     sub-tile) units: then `sc1` without `nt` (the policy the SWEEP kernel takes for such images, spmv_sweep.hip) shortens the drained pipeline's
//     refill at every unit border -- ogbl-ppa's 8-way slabs 16.8 -> 15.3 us, its 2-way slabs 34.8 -> 32.1, mouse_gene 33.8 -> 33.4, the sliced DELTA
//     plans of the pruned-NN layers -2.5 % -- while a pure stream keeps `nt`: hollywood (872 MB) 137.5 -> 155 us without it, ogbn-products 203 -> 208, the
//     headline matrix round-robin over three images 56 -> 65 us, and the one-unit-per-block slabs of mouse_gene 12.4 -> 12.9 / 20.0 -> 20.9
//     (profiles/r06_rowblock_stream_policy*.txt).  A plan-time choice (hs_api.cpp: stream_resident), a template parameter here: bit 2 of kRing.
#ifndef HS_ROWBLOCK_STREAM_POLICY
#define HS_ROWBLOCK_STREAM_POLICY "nt"      // the stream loads of an image that is streamed from HBM every time
#endif
#ifndef HS_ROWBLOCK_RESIDENT_POLICY
#define HS_ROWBLOCK_RESIDENT_POLICY "sc1"   // ... and of one that stays in the Infinity Cache (kRing & 4)
#endif
// one asm statement per cache policy, chosen at compile time (the policy is part of the instruction text)
#define HS_BY_POLICY(kRes, STATEMENT)                                 \
    do {                                                              \
        if constexpr (kRes) { STATEMENT(HS_ROWBLOCK_RESIDENT_POLICY); } \
        else { STATEMENT(HS_ROWBLOCK_STREAM_POLICY); }                \
    } while (0)
#define HS_RING_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
                      "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31"
constexpr int kMaxDepth = 16;
constexpr int kOwner24Depth = 3;     // OWNER24: records in flight per wavefront (5.25 KiB)

// kRing & 3: 0 = PAIRS / OWNER with 32-bit position words, 1 = DELTA, 2 = PAIRS / OWNER with 24-bit position words, 3 = OWNER24 records;
// kRing & 4: the image stays in the Infinity Cache (stream loads with HS_ROWBLOCK_RESIDENT_POLICY instead of `nt`)
template <int kFmt, bool kRes>
struct RingT;
template <int kRing>
using Ring = RingT<(kRing & 3), (kRing & 4) != 0>;
template <bool kRes>
struct RingT<0, kRes> {   // PAIRS: one dwordx2 per lane and step
    static constexpr uint32_t kLaneBytes = 8;
    template <int K>
    static __device__ __forceinline__ void issue(const uint8_t* base, uint32_t byte_off, uint32_t lane_off) {
#define HS_ISSUE(P) asm volatile("s_nop 4\n\tglobal_load_dwordx2 a[%0:%1], %2, %3 " P ::"n"(2 * K), "n"(2 * K + 1), "v"(byte_off + lane_off), "s"(base) : "memory", HS_RING_AGPRS)
        HS_BY_POLICY(kRes, HS_ISSUE);
#undef HS_ISSUE
    }
    template <int K, int kDepth>
    static __device__ __forceinline__ void take(uint32_t& value, uint32_t& where) {
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]"
                     : "=v"(value), "=v"(where) : "n"(2 * K), "n"(2 * K + 1), "n"(kDepth - 1) : "memory");
    }
    // steps K and K + 1 together (neither slot has been re-issued yet: kDepth - 2 younger loads may stay in flight)
    template <int K, int kDepth>
    static __device__ __forceinline__ void take2(uint32_t& value0, uint32_t& where0, uint32_t& value1, uint32_t& where1) {
        asm volatile("s_waitcnt vmcnt(%8)\n\tv_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
                     "v_accvgpr_read_b32 %3, a[%7]"
                     : "=v"(value0), "=v"(where0), "=v"(value1), "=v"(where1)
                     : "n"(2 * K), "n"(2 * K + 1), "n"(2 * K + 2), "n"(2 * K + 3), "n"(kDepth - 2) : "memory");
    }
};
template <bool kRes>
struct RingT<1, kRes> {    // DELTA: a record = two slots per lane: {value A, value B} as one dwordx2, {gap A, gap B} as one dword
    static constexpr uint32_t kLaneBytes = 8;
    template <int K>
    static __device__ __forceinline__ void issue(const uint8_t* base, uint32_t byte_off, uint32_t lane_off) {
        static_assert(2 * K + 1 < kMaxDepth && K + kMaxDepth < 2 * kMaxDepth, "values in a0..a15, gap words in a16..a23: ring depth <= 8");
#define HS_ISSUE(P)                                                                                                                                         \
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 a[%0:%1], %3, %5 " P "\n\tglobal_load_dword a[%2], %4, %5 offset:512 " P ::"n"(2 * K), "n"(2 * K + 1),     \
                 "n"(K + kMaxDepth), "v"(byte_off + lane_off), "v"(byte_off + lane_off / 2), "s"(base)                                                     \
                 : "memory", HS_RING_AGPRS)
        HS_BY_POLICY(kRes, HS_ISSUE);
#undef HS_ISSUE
    }
    // value A, value B, gaps (A in the low half, B in the high half)
    template <int K, int kDepth>
    static __device__ __forceinline__ void take_record(uint32_t& value_a, uint32_t& value_b, uint32_t& gaps) {
        asm volatile("s_waitcnt vmcnt(%6)\n\tv_accvgpr_read_b32 %0, a[%3]\n\tv_accvgpr_read_b32 %1, a[%4]\n\tv_accvgpr_read_b32 %2, a[%5]"
                     : "=v"(value_a), "=v"(value_b), "=v"(gaps) : "n"(2 * K), "n"(2 * K + 1), "n"(K + kMaxDepth), "n"(2 * (kDepth - 1)) : "memory");
    }
};
template <bool kRes>
struct RingT<2, kRes> {    // 24-bit position words: a 448-byte step = 64 value dwords, then 64 x 3 bytes (local_row << 13 | local_col)
    static constexpr uint32_t kLaneBytes = 4;
    template <int K>
    static __device__ __forceinline__ void issue(const uint8_t* base, uint32_t byte_off, uint32_t lane_off) {
        // the position word is read as an UNALIGNED dword at byte 256 + 3 * lane (its top byte belongs to the next lane)
#define HS_ISSUE(P)                                                                                                                       \
    asm volatile("s_nop 4\n\tglobal_load_dword a[%0], %2, %4 " P "\n\tglobal_load_dword a[%1], %3, %4 offset:256 " P ::"n"(K),           \
                 "n"(K + kMaxDepth), "v"(byte_off + lane_off), "v"(byte_off + lane_off - lane_off / 4), "s"(base)                        \
                 : "memory", HS_RING_AGPRS)
        HS_BY_POLICY(kRes, HS_ISSUE);
#undef HS_ISSUE
    }
    template <int K, int kDepth>
    static __device__ __forceinline__ void take(uint32_t& value, uint32_t& where) {
        asm volatile("s_waitcnt vmcnt(%4)\n\tv_accvgpr_read_b32 %0, a[%2]\n\tv_accvgpr_read_b32 %1, a[%3]"
                     : "=v"(value), "=v"(where) : "n"(K), "n"(K + kMaxDepth), "n"(2 * (kDepth - 1)) : "memory");
        where &= 0xffffffu;
    }
    template <int K, int kDepth>
    static __device__ __forceinline__ void take2(uint32_t& value0, uint32_t& where0, uint32_t& value1, uint32_t& where1) {
        asm volatile("s_waitcnt vmcnt(%8)\n\tv_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\t"
                     "v_accvgpr_read_b32 %3, a[%7]"
                     : "=v"(value0), "=v"(where0), "=v"(value1), "=v"(where1)
                     : "n"(K), "n"(K + kMaxDepth), "n"(K + 1), "n"(K + 1 + kMaxDepth), "n"(2 * (kDepth - 2)) : "memory");
        where0 &= 0xffffffu;
        where1 &= 0xffffffu;
    }
};
template <bool kRes>
struct RingT<3, kRes> {    // OWNER24: a record = FOUR steps per lane: the 4 value words as one dwordx4, the 4 x 24-bit position words as one dwordx3
    // ring slot K (K < 4 records in flight): values in a[4K : 4K+3], position words in a[16+4K : 16+4K+2] (even-aligned tuples)
    template <int K>
    static __device__ __forceinline__ void issue(const uint8_t* base, uint32_t byte_off, uint32_t lane16, uint32_t lane12) {
        static_assert(K < 4 && 16 + 4 * K + 2 < 2 * kMaxDepth, "a0..a15: values of four records, a16..a30: their position words");
#define HS_ISSUE(P)                                                                                                                                          \
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 a[%0:%1], %4, %6 " P "\n\tglobal_load_dwordx3 a[%2:%3], %5, %6 offset:1024 " P ::"n"(4 * K), "n"(4 * K + 3), \
                 "n"(16 + 4 * K), "n"(16 + 4 * K + 2), "v"(byte_off + lane16), "v"(byte_off + lane12), "s"(base)                                            \
                 : "memory", HS_RING_AGPRS)
        HS_BY_POLICY(kRes, HS_ISSUE);
#undef HS_ISSUE
    }
    template <int K, int kDepth>
    static __device__ __forceinline__ void take(uint32_t (&v)[4], uint32_t (&w)[3]) {
        asm volatile("s_waitcnt vmcnt(%14)\n\tv_accvgpr_read_b32 %0, a[%7]\n\tv_accvgpr_read_b32 %1, a[%8]\n\tv_accvgpr_read_b32 %2, a[%9]\n\t"
                     "v_accvgpr_read_b32 %3, a[%10]\n\tv_accvgpr_read_b32 %4, a[%11]\n\tv_accvgpr_read_b32 %5, a[%12]\n\tv_accvgpr_read_b32 %6, a[%13]"
                     : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(w[0]), "=v"(w[1]), "=v"(w[2])
                     : "n"(4 * K), "n"(4 * K + 1), "n"(4 * K + 2), "n"(4 * K + 3), "n"(16 + 4 * K), "n"(16 + 4 * K + 1), "n"(16 + 4 * K + 2), "n"(2 * (kDepth - 1))
                     : "memory");
    }
};
// a pointer the compiler must keep in a scalar register pair
__device__ __forceinline__ const uint8_t* scalar_pointer(const uint8_t* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));   // (the builtin returns int:
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));         //  no sign extension, please)
    return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// Workgroup barrier that orders LDS traffic only: the consumers' prefetched global loads stay in flight
// across it (a plain __syncthreads() carries a generic release fence and would drain them).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <bool kFloat> struct OwnerSum { using type = float; };
template <> struct OwnerSum<false> { using type = uint32_t; };
// ---- the consumer side of one row block: stream this wavefront's chunks / records through all sub-tiles of the block ----
template <bool kFloat>
struct Consumer {
    using acc_t = typename Rows<kFloat>::acc_t;
    using prod_t = typename Rows<kFloat>::prod_t;
    static constexpr uint32_t kNoRow = 0xffffffffu;
    const uint8_t* stream;     // scalar_pointer: first chunk / record of this wavefront in the block
    UnitTable unit;
    uint32_t U, wave, lane, lane_off, ring, nrows, last;
    const uint32_t* xs;
    acc_t* ys;
    uint32_t u = 0, slot = 0, end = 0, base = 0;
    uint32_t next_end = 0;     // end_step of unit u + 1, requested one unit ahead (a scalar load whose latency is off the unit boundary)
    const uint32_t* xb;
    uint32_t pos = 0;          // DELTA: this lane's position in the current sub-tile
    bool head = true;          // DELTA: the next record of this wavefront is a head record
    uint32_t run_row = kNoRow; // PAIRS dense rows: row whose products are being summed in registers ...
    prod_t run_sum = 0;        // ... and this lane's share of that sum
    uint32_t lane_row = 0;     // DELTA dense rows: the row this LANE is on (its run of consecutive elements rarely leaves it) ...
    typename Rows<kFloat>::lane_t lane_sum = 0;  // ... and the lane's private sum on it, flushed to LDS when the row or the unit changes
    typename OwnerSum<kFloat>::type own_sum = 0;   // OWNER: the lane's sum on lane_row (4-byte accumulators touched by this wavefront only: fp32, or saturating Q8.24)
    uint32_t spare = 0;        // OWNER: the wavefront's own spare accumulator (local row nrows + wave)
    uint32_t row_base = 0;     // OWNER24: first local row of the wavefront's share of the current unit (rows are stored relative to it)
    uint32_t lane_off12 = 0;   // OWNER24: lane * 12, the lane's offset among a record's position words
    uint64_t t_flush = 0, t_barrier = 0;   // OWNER profiling build (kAblate & 256): clocks spent in end-of-unit flushes / at unit barriers
};

// OWNER profiling build: where a wavefront's time goes (HISPARSE_ABLATE=256, tools/owner_profile.py); 8 u64 per wavefront
__device__ uint64_t* g_owner_profile = nullptr;
// Timeline build (HISPARSE_ABLATE=512, full work, correct results; tools/rowblock_timeline.py): 100 MHz timestamps of the phases of the
// first kTimelineBlocks blocks of every workgroup, taken by consumer wavefront 0 and loader wavefront 14:
//   [0] block entered  [1] prologue done (accumulators zeroed, first sub-tile copied)  [2] own main loop finished
//   [3] every wavefront's main loop finished (barrier)  [4] result stores issued
__device__ uint64_t* g_rowblock_timeline = nullptr;
constexpr uint32_t kTimelineBlocks = 4, kTimelineStamps = 8;
template <int kAblate>
__device__ __forceinline__ void timeline_stamp(uint32_t k, uint32_t wave, uint32_t lane, uint32_t i) {
    if (!(kAblate & 512)) return;
    if (lane != 0 || (wave != 0 && wave != kConsumerWaves) || k >= kTimelineBlocks || !g_rowblock_timeline) return;
    const uint64_t t = __builtin_amdgcn_s_memrealtime();
    g_rowblock_timeline[((static_cast<size_t>(blockIdx.x) * kTimelineBlocks + k) * 2 + (wave ? 1 : 0)) * kTimelineStamps + i] = t;
}

// One step (slot K of the ring).  Returns false when the block is finished.
// kDelta: DELTA format, otherwise PAIRS; kDense: the block's rows are long (Block::flags & kBlockDenseRows): products are
// summed in registers first (PAIRS: by the whole wavefront on one row; DELTA: by every lane along its own run).
template <bool kFloat, int kRing, int kAblate, int kDepth, bool kDense, int K>
__device__ __forceinline__ bool consume_step(Consumer<kFloat>& c) {
    using R = Rows<kFloat>;
    constexpr bool kDelta = (kRing & 3) == 1, k24 = (kRing & 3) == 2;      // (kRing & 4: the cache policy of the stream loads, Ring<>)
    constexpr uint32_t kStride = kDelta ? kRecordBytes : k24 ? kWaveStrideBytes24 : kWaveStrideBytes;
    constexpr uint32_t kColMask = k24 ? kSubTileCols - 1u : 0xffffu, kRowShift = k24 ? kOwnerColBits : 16u;
    const uint32_t s = c.base + K;
    while (s == c.end) {               // this wavefront finished sub-tile u (possibly with no work in it)
        if (!kDelta && kDense && c.u + 1 == c.U && c.run_row != Consumer<kFloat>::kNoRow) {   // last sub-tile: hand the register sum over
            const typename R::prod_t sum = wave_sum(c.run_sum);
            if (c.lane == 0) R::add(c.ys, c.run_row, sum);
            c.run_row = Consumer<kFloat>::kNoRow;
        }
        if (kDelta && kDense) {        // every lane hands its private row sum over before the sub-tile changes
            R::add_lane(c.ys, c.lane_row, c.lane_sum);
            c.lane_row = c.nrows;      // the spare accumulator: the next flush of an idle lane adds 0 there
            c.lane_sum = 0;
        }
        if (!(kAblate & 8)) lds_barrier();
        if (++c.u == c.U) return false;
        c.end = c.next_end;
        c.next_end = c.unit[min(c.u + 1, c.U - 1)].end_step[c.wave];
        c.slot = c.slot + 1 == c.ring ? 0 : c.slot + 1;
        c.xb = c.xs + c.slot * kSubTileCols;
        c.head = true;
    }
    uint32_t mat, aux;                 // value word; PAIRS: row << 16 | col
    if constexpr (kDelta) {
        // A lane owns a run of consecutive slots of the position-sorted unit.  The head slot of every (unit, wavefront) -- slot A of
        // its first record -- gives each lane its absolute start position (local_row * 8192 + local_col); every following slot carries
        // one value word and one 16-bit gap.  No per-lane branch: a bridge slot (gap 0xffff, value 0) advances 65535 and adds 0 at a
        // valid row of the block; fixed-point padding is (gap 0, value 0); float padding is a bridge whose row is clamped to the
        // spare accumulator ys[nrows] (0 * x could be NaN for a non-finite x, so float bridge slots add a literal 0).
        uint32_t value_a, value_b, gaps;
        Ring<kRing>::template take_record<K, kDepth>(value_a, value_b, gaps);
        auto slot = [&](uint32_t value, uint32_t gap) {
            c.pos += gap;
            uint32_t row = c.pos / kSubTileCols;
            const uint32_t col = c.pos % kSubTileCols;
            if (kAblate & 1) {
                asm volatile("" ::"v"(row), "v"(value));
                return;
            }
            const uint32_t xv = (kAblate & 2) ? col : c.xb[col];
            typename R::prod_t prod = R::product(value, xv);
            if (kFloat) {
                if (gap == kBridgeGap) prod = 0;
                row = min(row, c.nrows);
            }
            if (kDense) {
                // Dense rows: a lane's run of consecutive sorted elements stays on one row for many steps (and the lanes
                // of one instruction would collide on the few rows there are): sum in a register, touch LDS on row changes.
                if (row != c.lane_row) {       // per lane (exec-masked)
                    R::add_lane(c.ys, c.lane_row, c.lane_sum);
                    c.lane_row = row;
                    c.lane_sum = 0;
                }
                c.lane_sum += prod;
            } else {
                R::add(c.ys, row, prod);
            }
        };
        if (c.head) {                  // wave-uniform: slot A is the head
            c.pos = value_a;
            c.head = false;
        } else {
            slot(value_a, gaps & 0xffffu);
        }
        slot(value_b, gaps >> 16);
    } else {
        Ring<kRing>::template take<K, kDepth>(mat, aux);
        const uint32_t xv = (kAblate & 2) ? aux : c.xb[aux & kColMask];
        if (kAblate & 1) {
            asm volatile("" ::"v"(xv), "v"(mat));
        } else {
            const typename R::prod_t prod = R::product(mat, xv);
            const uint32_t row = aux >> kRowShift;
            if (kDense) {
                // Chunks are row-sorted, so the whole wavefront is usually on ONE row, and stays on it for many chunks:
                // keep a per-lane running sum in registers while the row does not change and touch the LDS accumulator
                // only when it does (one wave_sum + one ds_add per row per wavefront instead of a 64-way conflicting
                // atomic per chunk).
                const uint32_t row0 = __builtin_amdgcn_readfirstlane(row);
                const bool uniform = __ballot(row == row0) == ~0ull;
                if (uniform && row0 == c.run_row) {
                    c.run_sum += prod;
                } else {
                    if (c.run_row != Consumer<kFloat>::kNoRow) {
                        const typename R::prod_t sum = wave_sum(c.run_sum);
                        if (c.lane == 0) R::add(c.ys, c.run_row, sum);
                    }
                    if (uniform) { c.run_row = row0; c.run_sum = prod; }
                    else { c.run_row = Consumer<kFloat>::kNoRow; c.run_sum = 0; R::add(c.ys, row, prod); }   // chunk straddles rows
                }
            } else {
                R::add(c.ys, row, prod);
            }
        }
    }
    Ring<kRing>::template issue<K>(c.stream, min(s + kDepth, c.last) * kStride, c.lane_off);
    return true;
}

// ---- OWNER format (stream_tiles.h): float accumulators without atomics ----------------------------------------------------------
// The wavefront owns its rows: nobody else reads or writes ys32[row] for the rows in its stream, and LDS executes one wavefront's
// instructions in order, so a plain read-modify-write is safe as long as the lanes of ONE instruction hold distinct rows.  The
// builder deals a (unit, wavefront) share -- sorted by (row, column) -- to the lanes in consecutive runs, so rows are
// non-decreasing from lane to lane.  A lane sums while its row stays the same and writes when it changes: two lanes flushing in
// the same step flush different rows (the lower lane's old row is below its new row, which is at most the higher lane's first
// row).  Only at the end of a unit, when every lane hands its last row over, can neighbours hold the same row: one segmented
// wavefront reduction first.
// The arithmetic of an OWNER accumulator (4 bytes, touched by one wavefront only).  Float: fp32 product and sum, like the float PEs.
// Fixed point (OWNER24 only): Q8.24 products (rounded / saturated one by one) added with SATURATING unsigned adds -- min(a + b, 2^32-1)
// is associative and commutative on non-negative terms, so any order of such adds gives min(sum, 2^32-1) = the PE's saturating running
// sum (pe.h:72), bit for bit, in four bytes instead of the eight the atomic paths need for their exact 64-bit sums.
template <bool kFloat>
struct OwnerOps;
template <>
struct OwnerOps<true> {
    using val_t = float;
    static __device__ __forceinline__ val_t product(uint32_t mat, uint32_t xv) { return __uint_as_float(mat) * __uint_as_float(xv); }   // pe-stall.h:52
    static __device__ __forceinline__ val_t add(val_t a, val_t b) { return a + b; }
};
template <>
struct OwnerOps<false> {
    using val_t = uint32_t;
    static __device__ __forceinline__ val_t product(uint32_t mat, uint32_t xv) { return q8_24_mul(mat, xv); }
    static __device__ __forceinline__ val_t add(val_t a, val_t b) { return __builtin_elementwise_add_sat(a, b); }      // v_add_u32 ... clamp
};
template <bool kFloat>
__device__ __forceinline__ void owner_flush(typename OwnerOps<kFloat>::val_t* ys32, uint32_t row, typename OwnerOps<kFloat>::val_t sum) {
    ys32[row] = OwnerOps<kFloat>::add(ys32[row], sum);
}
template <bool kFloat>
__device__ __forceinline__ void owner_end_of_unit(Consumer<kFloat>& c) {
    using Ops = OwnerOps<kFloat>;
    using val_t = typename Ops::val_t;
    val_t* ys32 = reinterpret_cast<val_t*>(c.ys);
    const uint32_t row = c.lane_row;
    val_t sum = c.own_sum;
    const bool live = row != c.spare;                           // lanes that saw only padding have nothing to hand over
    const uint32_t above = __shfl_down(row, 1, kWaveLanes);
    const bool same_above = live && c.lane + 1 < kWaveLanes && above == row;
    if (__ballot(same_above) == 0) {                            // the usual case in a hyper-sparse unit: all rows distinct
        if (live) owner_flush<kFloat>(ys32, row, sum);
    } else {
        // equal rows are contiguous lanes: suffix sums inside every run of equal rows, the run's first lane writes
#pragma unroll
        for (uint32_t d = 1; d < kWaveLanes; d <<= 1) {
            const val_t s2 = __shfl_down(sum, d, kWaveLanes);
            const uint32_t r2 = __shfl_down(row, d, kWaveLanes);
            if (c.lane + d < kWaveLanes && r2 == row) sum = Ops::add(sum, s2);
        }
        const uint32_t below = __shfl_up(row, 1, kWaveLanes);
        if (live && (c.lane == 0 || below != row)) owner_flush<kFloat>(ys32, row, sum);
    }
    c.lane_row = c.spare;
    c.own_sum = 0;
}

template <int kRing, int kAblate, int kDepth, int K>
__device__ __forceinline__ bool consume_step_owner(Consumer<true>& c) {
    constexpr uint32_t kStep = kChunkBytes;
    static_assert(kRing == 0, "8-byte OWNER chunks");
    const uint32_t s = c.base + K;
    while (s == c.end) {               // this wavefront finished sub-tile u (possibly with no work in it)
        if (kAblate & 256) {
            const uint64_t t0 = __builtin_readcyclecounter();
            owner_end_of_unit(c);
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the flush has executed
            const uint64_t t1 = __builtin_readcyclecounter();
            lds_barrier();
            const uint64_t t2 = __builtin_readcyclecounter();
            c.t_flush += t1 - t0;
            c.t_barrier += t2 - t1;
        } else {
        if (!(kAblate & 1)) owner_end_of_unit(c);
        if (!(kAblate & 8)) lds_barrier();
        }
        if (++c.u == c.U) return false;
        c.end = c.next_end;
        c.next_end = c.unit[min(c.u + 1, c.U - 1)].end_step[c.wave];
        c.slot = c.slot + 1 == c.ring ? 0 : c.slot + 1;
        c.xb = c.xs + c.slot * kSubTileCols;
    }
    uint32_t mat, where;               // value word; local_row << 13 | local_col
    Ring<kRing>::template take<K, kDepth>(mat, where);
    const uint32_t row = where >> kOwnerColBits, col = where & (kSubTileCols - 1u);
    if (kAblate & 1) {
        const uint32_t xv = (kAblate & 2) ? where : c.xb[col];
        asm volatile("" ::"v"(xv), "v"(mat), "v"(row));
    } else {
        // The accumulator of the row the lane is LEAVING and the x word of the element it has just taken are independent:
        // both LDS reads go out back to back and their latencies overlap (one LDS round trip per step instead of two).
        float* ys32 = reinterpret_cast<float*>(c.ys);
        const bool leaving = row != c.lane_row;     // per lane; the very first step of a unit leaves the spare accumulator (adds 0)
        float old = 0.0f;
        if (leaving) old = ys32[c.lane_row];
        const uint32_t xv = (kAblate & 2) ? where : c.xb[col];
        const float prod = __uint_as_float(mat) * __uint_as_float(xv);     // one fp32 multiply, like the float PEs (pe-stall.h:52)
        if (leaving) {
            ys32[c.lane_row] = old + c.own_sum;
            c.lane_row = row;
            c.own_sum = 0;
        }
        c.own_sum += prod;
    }
    Ring<kRing>::template issue<K>(c.stream, min(s + kDepth, c.last) * kStep, c.lane_off);
    return true;
}
// Two steps of the same unit at once: the two x words and the accumulators of the (up to two) rows the lane leaves are four
// independent LDS reads -- one round trip for two steps.  (The row entered at step K can be left at step K + 1; its accumulator is
// not written by step K, whose write goes to the row left THERE, and no other lane's write of this pair can hit it: a higher lane
// leaves rows >= this lane's last row, which this lane never leaves, a lower lane leaves rows below this lane's first row.)
template <int kRing, int kAblate, int kDepth, int K>
__device__ __forceinline__ bool consume_pair_owner(Consumer<true>& c) {
    constexpr uint32_t kStep = kChunkBytes;
    static_assert(kRing == 0, "8-byte OWNER chunks");
    const uint32_t s = c.base + K;
    if ((kAblate & 1) || s == c.end || s + 1 == c.end) {     // a unit ends at or inside the pair: one step at a time
        if (!consume_step_owner<kRing, kAblate, kDepth, K>(c)) return false;
        return consume_step_owner<kRing, kAblate, kDepth, K + 1>(c);
    }
    uint32_t mat0, where0, mat1, where1;
    Ring<kRing>::template take2<K, kDepth>(mat0, where0, mat1, where1);
    const uint32_t row0 = where0 >> kOwnerColBits, col0 = where0 & (kSubTileCols - 1u);
    const uint32_t row1 = where1 >> kOwnerColBits, col1 = where1 & (kSubTileCols - 1u);
    float* ys32 = reinterpret_cast<float*>(c.ys);
    const bool leaving0 = row0 != c.lane_row, leaving1 = row1 != row0;
    float old0 = 0.0f, old1 = 0.0f;
    if (leaving0) old0 = ys32[c.lane_row];
    if (leaving1) old1 = ys32[row0];
    const uint32_t xv0 = (kAblate & 2) ? where0 : c.xb[col0];
    const uint32_t xv1 = (kAblate & 2) ? where1 : c.xb[col1];
    const float prod0 = __uint_as_float(mat0) * __uint_as_float(xv0), prod1 = __uint_as_float(mat1) * __uint_as_float(xv1);
    if (leaving0) {
        ys32[c.lane_row] = old0 + c.own_sum;
        c.own_sum = 0;
    }
    c.own_sum += prod0;
    if (leaving1) {
        ys32[row0] = old1 + c.own_sum;
        c.own_sum = 0;
    }
    c.own_sum += prod1;
    c.lane_row = row1;
    Ring<kRing>::template issue<K>(c.stream, min(s + kDepth, c.last) * kStep, c.lane_off);
    Ring<kRing>::template issue<K + 1>(c.stream, min(s + 1 + kDepth, c.last) * kStep, c.lane_off);
    return true;
}
template <int kRing, int kAblate, int kDepth, int... Ks>
__device__ __forceinline__ bool consume_round_owner(Consumer<true>& c, std::integer_sequence<int, Ks...>) {
    static_assert(kDepth % 2 == 0, "steps are taken in pairs");
    return ((Ks % 2 != 0 || consume_pair_owner<kRing, kAblate, kDepth, Ks>(c)) && ...);
}

// ---- OWNER24 (stream_tiles.h): the OWNER scheme over records of four steps with 24-bit position words --------------------------------
// Steps are numbered through the block; record r holds steps 4r .. 4r+3, which may belong to different units (the unit test sits in
// front of every step, as above).  A position word holds the row relative to the first row of the wavefront's share of the CURRENT
// unit (Consumer::row_base, from the high half of Unit::end_step), 2047 = the wavefront's spare accumulator.
template <bool kFloat, int kAblate>
__device__ __forceinline__ bool owner24_next_unit(Consumer<kFloat>& c, uint32_t s) {      // false: the block is finished
    while (s == c.end) {               // this wavefront finished sub-tile u (possibly with no work in it)
        if (!(kAblate & 1)) owner_end_of_unit<kFloat>(c);
        if (!(kAblate & 8)) lds_barrier();
        if (++c.u == c.U) return false;
        const uint32_t packed = c.next_end;
        c.end = packed & kOwnerStepMask;
        c.row_base = packed >> 16;
        c.next_end = c.unit[min(c.u + 1, c.U - 1)].end_step[c.wave];
        c.slot = c.slot + 1 == c.ring ? 0 : c.slot + 1;
        c.xb = c.xs + c.slot * kSubTileCols;
    }
    return true;
}
template <bool kFloat>
__device__ __forceinline__ uint32_t owner24_row(const Consumer<kFloat>& c, uint32_t where) {
    const uint32_t field = where >> kOwnerColBits;
    return field == kOwnerSpareField ? c.spare : c.row_base + field;
}
template <bool kFloat, int kAblate>
__device__ __forceinline__ void owner24_one(Consumer<kFloat>& c, uint32_t mat, uint32_t where) {
    using Ops = OwnerOps<kFloat>;
    using val_t = typename Ops::val_t;
    const uint32_t row = owner24_row<kFloat>(c, where), col = where & (kSubTileCols - 1u);
    if (kAblate & 1) {
        asm volatile("" ::"v"(mat), "v"(row), "v"(col));
        return;
    }
    val_t* ys32 = reinterpret_cast<val_t*>(c.ys);
    const bool leaving = row != c.lane_row;     // per lane; the very first step of a unit leaves the spare accumulator (adds 0)
    val_t old = 0;
    if (leaving) old = ys32[c.lane_row];
    const uint32_t xv = c.xb[col];
    const val_t prod = Ops::product(mat, xv);
    if (leaving) {
        ys32[c.lane_row] = Ops::add(old, c.own_sum);
        c.lane_row = row;
        c.own_sum = 0;
    }
    c.own_sum = Ops::add(c.own_sum, prod);
}
// steps s and s + 1 (see consume_pair_owner for why the four LDS reads may go out together)
template <bool kFloat, int kAblate>
__device__ __forceinline__ bool owner24_pair(Consumer<kFloat>& c, uint32_t s, uint32_t mat0, uint32_t where0, uint32_t mat1, uint32_t where1) {
    using Ops = OwnerOps<kFloat>;
    using val_t = typename Ops::val_t;
    if ((kAblate & 1) || s == c.end || s + 1 == c.end) {      // a unit ends at or inside the pair: one step at a time
        if (!owner24_next_unit<kFloat, kAblate>(c, s)) return false;
        owner24_one<kFloat, kAblate>(c, mat0, where0);
        if (!owner24_next_unit<kFloat, kAblate>(c, s + 1)) return false;
        owner24_one<kFloat, kAblate>(c, mat1, where1);
        return true;
    }
    const uint32_t row0 = owner24_row<kFloat>(c, where0), col0 = where0 & (kSubTileCols - 1u);
    const uint32_t row1 = owner24_row<kFloat>(c, where1), col1 = where1 & (kSubTileCols - 1u);
    val_t* ys32 = reinterpret_cast<val_t*>(c.ys);
    const bool leaving0 = row0 != c.lane_row, leaving1 = row1 != row0;
    val_t old0 = 0, old1 = 0;
    if (leaving0) old0 = ys32[c.lane_row];
    if (leaving1) old1 = ys32[row0];
    const uint32_t xv0 = c.xb[col0], xv1 = c.xb[col1];
    const val_t prod0 = Ops::product(mat0, xv0), prod1 = Ops::product(mat1, xv1);
    if (leaving0) {
        ys32[c.lane_row] = Ops::add(old0, c.own_sum);
        c.own_sum = 0;
    }
    c.own_sum = Ops::add(c.own_sum, prod0);
    if (leaving1) {
        ys32[row0] = Ops::add(old1, c.own_sum);
        c.own_sum = 0;
    }
    c.own_sum = Ops::add(c.own_sum, prod1);
    c.lane_row = row1;
    return true;
}
// one record (ring slot K): take it, put the record kDepth further on in flight in its place, then its four steps
template <bool kFloat, int kAblate, int kDepth, int K>
__device__ __forceinline__ bool consume_record_owner24(Consumer<kFloat>& c) {
    const uint32_t r = c.base + K;
    uint32_t v[4], w[3];
    Ring<3>::template take<K, kDepth>(v, w);
    Ring<3>::template issue<K>(c.stream, min(r + kDepth, c.last) * kOwnerRecordBytes, c.lane_off, c.lane_off12);
    const uint32_t where0 = w[0] & 0xffffffu, where1 = __builtin_amdgcn_alignbit(w[1], w[0], 24) & 0xffffffu;
    const uint32_t where2 = __builtin_amdgcn_alignbit(w[2], w[1], 16) & 0xffffffu, where3 = w[2] >> 8;
    if (!owner24_pair<kFloat, kAblate>(c, 4 * r, v[0], where0, v[1], where1)) return false;
    return owner24_pair<kFloat, kAblate>(c, 4 * r + 2, v[2], where2, v[3], where3);
}
template <bool kFloat, int kAblate, int kDepth, int... Ks>
__device__ __forceinline__ bool consume_round_owner24(Consumer<kFloat>& c, std::integer_sequence<int, Ks...>) {
    return (consume_record_owner24<kFloat, kAblate, kDepth, Ks>(c) && ...);
}
template <bool kFloat, int... Ks>
__device__ __forceinline__ void prime_ring_owner24(Consumer<kFloat>& c, std::integer_sequence<int, Ks...>) {
    (Ring<3>::template issue<Ks>(c.stream, min(static_cast<uint32_t>(Ks), c.last) * kOwnerRecordBytes, c.lane_off, c.lane_off12), ...);
}
template <bool kFloat, int kDepth>
__device__ __forceinline__ void consumer_begin_owner24(Consumer<kFloat>& c, const uint8_t* stream, UnitTable unit, uint32_t U, uint32_t wave, uint32_t lane,
                                                       const uint32_t* xs, uint32_t ring, typename Rows<kFloat>::acc_t* ys, uint32_t nrows, uint32_t total_steps,
                                                       uint32_t first_end) {
    static_assert(kDepth >= 1 && kDepth <= 4, "records in flight: a0..a30");
    c.stream = scalar_pointer(stream);
    c.unit = unit; c.U = U; c.wave = wave; c.lane = lane; c.ring = ring; c.nrows = nrows;
    c.lane_off = lane * 16u;
    c.lane_off12 = lane * 12u;
    const uint32_t records = (total_steps + kOwnerRecordSteps - 1) / kOwnerRecordSteps;
    c.last = records ? records - 1 : 0;          // prefetches past the end re-read the last record (no branch)
    c.xs = xs; c.xb = xs; c.ys = ys;
    c.end = first_end & kOwnerStepMask;
    c.row_base = first_end >> 16;
    c.next_end = U > 1 ? unit[1].end_step[wave] : first_end;
    c.lane_row = nrows + wave;
    c.spare = nrows + wave;
    prime_ring_owner24<kFloat>(c, std::make_integer_sequence<int, kDepth>());
}
template <bool kFloat, int kAblate, int kDepth>
__device__ __forceinline__ void consumer_run_owner24(Consumer<kFloat>& c) {
    __builtin_amdgcn_s_waitcnt(0x0f70);   // see consumer_run: clears hipcc's "LDS-DMA may be pending"
    for (;; c.base += kDepth)
        if (!consume_round_owner24<kFloat, kAblate, kDepth>(c, std::make_integer_sequence<int, kDepth>())) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", HS_RING_AGPRS);   // the clamped tail prefetches must land before the ring is reused
}

template <bool kFloat, int kRing, int kAblate, int kDepth, bool kDense, int... Ks>
__device__ __forceinline__ bool consume_round(Consumer<kFloat>& c, std::integer_sequence<int, Ks...>) {
    return (consume_step<kFloat, kRing, kAblate, kDepth, kDense, Ks>(c) && ...);
}
template <int kRing, int... Ks>
__device__ __forceinline__ void prime_ring(const uint8_t* stream, uint32_t last, uint32_t stride, uint32_t lane_off, std::integer_sequence<int, Ks...>) {
    (Ring<kRing>::template issue<Ks>(stream, min(static_cast<uint32_t>(Ks), last) * stride, lane_off), ...);
}

// Before the block's prologue: set the wavefront's consumer up and put the first kDepth loads in flight, so that the HBM
// latency of the stream overlaps the accumulator zeroing and the first x sub-tile copy.
template <bool kFloat, int kRing, int kDepth, bool kOwner = false>
__device__ __forceinline__ void consumer_begin(Consumer<kFloat>& c, const uint8_t* stream, UnitTable unit, uint32_t U, uint32_t wave,
                                               uint32_t lane, const uint32_t* xs, uint32_t ring, typename Rows<kFloat>::acc_t* ys, uint32_t nrows,
                                               uint32_t total, uint32_t first_end) {
    static_assert(kDepth <= kMaxDepth, "the ring lives in a0..a31");
    constexpr bool kDelta = (kRing & 3) == 1, k24 = (kRing & 3) == 2;      // (kRing & 4: the cache policy of the stream loads, Ring<>)
    constexpr uint32_t kStride = kDelta ? kRecordBytes : kOwner ? kChunkBytes : k24 ? kWaveStrideBytes24 : kWaveStrideBytes;
    c.stream = scalar_pointer(stream);
    c.unit = unit; c.U = U; c.wave = wave; c.lane = lane; c.ring = ring; c.nrows = nrows;
    c.lane_off = lane * Ring<kRing>::kLaneBytes;
    c.last = total ? total - 1 : 0;    // prefetches past the end re-read the last chunk / record (no branch)
    c.xs = xs; c.xb = xs; c.ys = ys;
    c.end = first_end;
    c.next_end = U > 1 ? unit[1].end_step[wave] : first_end;
    c.lane_row = kOwner ? nrows + wave : nrows;
    c.spare = nrows + wave;
    prime_ring<kRing>(c.stream, c.last, kStride, c.lane_off, std::make_integer_sequence<int, kDepth>());
}

template <bool kFloat, int kRing, int kAblate, int kDepth, bool kDense, bool kOwner = false>
__device__ __forceinline__ void consumer_run(Consumer<kFloat>& c) {
    // The loader branch of the kernel leaves "LDS-DMA may be pending" in hipcc's wait-count bookkeeping, and that state
    // reaches this loop around the block loop and through the shared prologue: every LDS store on a conditional path below
    // would then get its own s_waitcnt vmcnt(0) and drain the prefetch ring.  A consumer wavefront never has LDS-DMA in
    // flight, so say so once, here, where it costs nothing: the prologue's x copy has just waited for vmcnt(0) anyway.
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), lgkmcnt/expcnt untouched
    if (kAblate & 64) { c.u = c.U - 1; c.end = c.last + 1; }   // profiling: one unit per block
    const uint64_t t_begin = (kOwner && (kAblate & 256)) ? __builtin_readcyclecounter() : 0;
    for (;; c.base += kDepth) {
        if constexpr (kOwner) {
            if (!consume_round_owner<kRing, kAblate, kDepth>(c, std::make_integer_sequence<int, kDepth>())) break;
        } else {
            if (!consume_round<kFloat, kRing, kAblate, kDepth, kDense>(c, std::make_integer_sequence<int, kDepth>())) break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", HS_RING_AGPRS);   // the clamped tail prefetches must land before the ring is reused
    if (kOwner && (kAblate & 256) && c.lane == 0 && g_owner_profile) {
        uint64_t* p = g_owner_profile + (static_cast<size_t>(blockIdx.x) * kWavesPerWorkgroup + c.wave) * 8;
        p[0] += __builtin_readcyclecounter() - t_begin;   // the wavefront's whole consumer phase
        p[1] += c.t_flush;
        p[2] += c.t_barrier;
        p[3] += c.U;
        p[4] += c.last + 1;                               // steps
    }
}

// kDepth: element loads in flight per lane (kDepth x 512 B per wavefront).
// kAblate (profiling builds only, HISPARSE_ABLATE): bit 0 = no LDS accumulate, bit 1 = no LDS gather,
// bit 2 = no x sub-tile refill, bit 3 = no per-sub-tile barrier, bit 4 = no block prologue (zero + first sub-tile),
// bit 5 = no result store, bit 6 = ignore unit boundaries.  Any non-zero value gives wrong results.
// kRing: 1 = the image is in the DELTA stream format (stream_tiles.h), 0 = PAIRS / OWNER chunks with 32-bit position words, 2 = with 24-bit ones.
// kOwner: the image is in the OWNER format (float only): 4-byte float accumulators, nrows + 14 of them.
template <bool kFloat, int kRing, int kAblate, int kDepth, bool kOwner = false>
__global__ __launch_bounds__(kThreads) void spmv_rowblock_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks,
                                                                  const Unit* __restrict__ units, const uint32_t* __restrict__ x,
                                                                  uint32_t* __restrict__ out, int32_t row_part_filter, uint32_t ring,
                                                                  uint32_t x_base, const uint32_t* __restrict__ part_heads, CarriedCombine carry) {
    using acc_t = typename Rows<kFloat>::acc_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    acc_t* ys = reinterpret_cast<acc_t*>(lds);                    // [nrows + 1] at LDS address 0: row addresses need no base add
    uint32_t* xs = reinterpret_cast<uint32_t*>(lds + x_base);     // [ring][kSubTileCols]
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    const bool loader = wave >= kConsumerWaves;
    // Workgroup b runs on XCD b % 8 (observed dispatch order): neighbouring logical workgroups (neighbouring row
    // blocks, same x sub-tiles at about the same time) then share an L2.  Speed only, never correctness.
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    // The workgroup's first block is blocks[wg]; further ones are chained through Block::next (0 = none).  Everything a
    // consumer wavefront needs before its first stream load sits in the Block itself: ONE dependent load per block.
    // hs_run_partition: only the blocks that reach into one row partition -- the workgroup's chain is in row order, part_heads says
    // where this partition's stretch begins, and it goes on while the next block begins in this partition or before (Block::next_part).
    uint32_t bi = wg;
    if (row_part_filter >= 0) {
        bi = ((const __attribute__((address_space(4))) uint32_t*)part_heads)[static_cast<uint32_t>(row_part_filter) * gridDim.x + wg];
        if (bi == kNoBlock) return;
    }
    bool first_block = true;
    uint32_t block_no = 0;      // timeline build only
    for (uint32_t next = 0;; bi = next, ++block_no) {
        const BlockTable blk = (BlockTable)(blocks + bi);
        next = (row_part_filter >= 0 && blk->next_part > static_cast<uint32_t>(row_part_filter)) ? 0u : blk->next;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset;
        const UnitTable unit = (UnitTable)(units + blk->unit_begin);
        const uint32_t U = blk->unit_end - blk->unit_begin;

        if (!first_block) __syncthreads();   // the previous block's result store has read the accumulators (and has drained)
        first_block = false;
        timeline_stamp<kAblate>(block_no, wave, lane, 0);
        Consumer<kFloat> c;
        if (!loader && U > 0) {
            if constexpr (kRing == 3)
                consumer_begin_owner24<kFloat, kDepth>(c, image + blk->wave_offset[wave], unit, U, wave, lane, xs, ring, ys, nrows, blk->total_steps[wave], blk->first_end[wave]);
            else
                consumer_begin<kFloat, kRing, kDepth, kOwner>(c, image + blk->wave_offset[wave], unit, U, wave, lane, xs, ring, ys, nrows, blk->total_steps[wave],
                                                               blk->first_end[wave]);
        }
        // y of the PREVIOUS step (spmv_device.h: CarriedCombine), once per launch, HERE: the consumers' first stream loads (and the block
        // descriptor behind them) are already travelling, so the round trip of the partial rows overlaps theirs instead of preceding it
        if (carry.partial && block_no == 0) carried_combine<kFloat, kThreads>(carry, blockIdx.x, gridDim.x, tid);
        // Prologue.  (Measured in round 3, timeline build HISPARSE_ABLATE=512: 4.3 us pass between a workgroup's first wavefront
        // entering and this barrier, and most of that is the LAUNCH ramp -- the 16 wavefronts of a 1024-thread workgroup are started
        // 2 - 4 us apart, the loaders last -- not this code: letting the loaders DMA sub-tile 0 while only the consumers zero changed
        // nothing, same box: ogbl-ppa 57.5 / 57.4 us, mouse_gene 38.5 / 38.2 us.  Round 5: the copy of sub-tile 0 by the FIRST 256 threads
        // only, eight loads each -- the wavefronts that enter first -- is slower everywhere, +0.4 ... +1.1 us per step on eight
        // configurations: profiles/r05_early_fill_ab.txt.)
        if (kOwner) {
            if (!(kAblate & 16)) for (uint32_t i = tid; i < nrows + kConsumerWaves; i += kThreads) reinterpret_cast<float*>(ys)[i] = 0.0f;
        } else
        if (!(kAblate & 16)) for (uint32_t i = tid; i <= nrows; i += kThreads) ys[i] = 0;           // PE banks start at zero (pe.h:131-135)
        if (U > 0 && !(kAblate & 16)) fill_x<kSubTileCols / 4 / kThreads, kThreads>(xs, x, blk->first_col0, blk->first_ncols, tid);
        __syncthreads();
        timeline_stamp<kAblate>(block_no, wave, lane, 1);

        if (U > 0) {
            if (loader) {
                // ---- loader wavefronts: keep the ring of x buffers up to three sub-tiles ahead of the consumers ----
                const uint32_t lw = wave - kConsumerWaves;
                __builtin_amdgcn_s_setprio(3);                         // the refills are on every unit's critical path: issue them first
                uint32_t fill_slot = 1 == ring ? 0 : 1;                // ring slot of the next refill (sub-tile v lives in slot v % ring)
                // (Round 3: a ring of unit descriptors in LDS, filled by the loaders with vector loads riding on the refills and read by
                // the consumers with ds_read -- no scalar load anywhere in the unit loop -- was built, parity- and soak-green, and
                // changed nothing: mouse_gene slab 22.0 vs 21.7 us, pokec 89.9 vs 89.4, ogbl-ppa 58.5 vs 57.7, ogbn-products 206 vs 204,
                // same box.  The scalar descriptor loads below are not what a short unit waits for.)
                // Unit descriptors are 64 bytes apart and cold (they stream from HBM once per SpMV): the descriptor of the
                // refill after next is fetched while this one is in flight, so its miss latency is off the per-unit path
                // (hyper-sparse matrices have hundreds of short units per block: 1 us each used to add up to 300 us).
                uint32_t col0_next = U > 1 ? unit[1].col0 : 0, ncols_next = U > 1 ? unit[1].ncols : 8;
                uint32_t col0_after = U > 2 ? unit[2].col0 : col0_next, ncols_after = U > 2 ? unit[2].ncols : ncols_next;   // kOwner: two ahead
                auto refill = [&](uint32_t v) {
                    const uint32_t col0 = col0_next, ncols = ncols_next;
                    if (kOwner) {
                        col0_next = col0_after;
                        ncols_next = ncols_after;
                        const uint32_t ahead = min(v + 2, U - 1);
                        col0_after = unit[ahead].col0;
                        ncols_after = unit[ahead].ncols;
                    } else {
                        const uint32_t ahead = min(v + 1, U - 1);
                        col0_next = unit[ahead].col0;
                        ncols_next = unit[ahead].ncols;
                    }
                    if (!(kAblate & 4)) dma_fill_x(xs + fill_slot * kSubTileCols, x, col0, ncols, lw, lane);
                    fill_slot = fill_slot + 1 == ring ? 0 : fill_slot + 1;
                };
                uint32_t issued = 1;                                   // sub-tile 0 was copied synchronously above
                while (issued < U && issued < ring - 1) refill(issued++);
                uint64_t t_wait = 0, t_bar = 0, t_issue = 0;
                const uint64_t t_begin = (kOwner && (kAblate & 256)) ? __builtin_readcyclecounter() : 0;
                for (uint32_t u = 0; u < U; ++u) {
                    // slot (u + ring - 1) % ring last held sub-tile u-1, which every consumer left at the previous barrier
                    const uint64_t ti = (kOwner && (kAblate & 256)) ? __builtin_readcyclecounter() : 0;
                    if (issued < U) refill(issued++);
                    if (kOwner && (kAblate & 256)) t_issue += __builtin_readcyclecounter() - ti;
                    // sub-tile u+1 must be resident before the consumers enter it (they do so after this barrier)
                    const uint32_t younger = issued - min(issued, u + 2);   // refills issued after the one for u+1: 0..ring-2
                    const uint64_t t0 = (kOwner && (kAblate & 256)) ? __builtin_readcyclecounter() : 0;
                    if (younger >= 2) dma_wait<2>(); else if (younger == 1) dma_wait<1>(); else dma_wait<0>();
                    const uint64_t t1 = (kOwner && (kAblate & 256)) ? __builtin_readcyclecounter() : 0;
                    if (!(kAblate & 8)) __builtin_amdgcn_s_barrier();
                    if (kOwner && (kAblate & 256)) { const uint64_t t2 = __builtin_readcyclecounter(); t_wait += t1 - t0; t_bar += t2 - t1; }
                }
                if (kOwner && (kAblate & 256) && lane == 0 && g_owner_profile) {
                    uint64_t* p = g_owner_profile + (static_cast<size_t>(blockIdx.x) * kWavesPerWorkgroup + wave) * 8;
                    p[0] += __builtin_readcyclecounter() - t_begin;
                    p[1] += t_wait;                                    // waiting for a refill to land
                    p[2] += t_bar;                                     // waiting for the consumers at the unit barrier
                    p[3] += U;
                    p[4] += t_issue;                                   // descriptor of the refill after next + issuing this one
                }
            } else {
                // ---- consumer wavefronts: stream elements, gather x, accumulate rows -----------------------------
                if constexpr (kRing == 3) consumer_run_owner24<kFloat, kAblate, kDepth>(c);
                else if constexpr (kOwner) consumer_run<kFloat, kRing, kAblate, kDepth, false, true>(c);
                else if (blk->flags & kBlockDenseRows) consumer_run<kFloat, kRing, kAblate, kDepth, true>(c);
                else consumer_run<kFloat, kRing, kAblate, kDepth, false>(c);
            }
        }
        timeline_stamp<kAblate>(block_no, wave, lane, 2);
        // Every sub-tile barrier has passed -- but LDS atomics WITHOUT return value can still be queued behind it: with heavy
        // same-address conflicts (dense rows in the plain DELTA path, ds_add_f64) the s_waitcnt lgkmcnt(0) in front of the barrier
        // did not cover them and the accumulators were read too early (40 % of the launches of a 15 %-dense float matrix lost one
        // record's worth of products; found by tests/gpu_fuzz_soak.py).  LDS executes one wavefront's instructions in order, so a
        // RETURNING atomic on the spare accumulator, awaited, proves that everything this wavefront queued before it is done.
        if (kOwner) {
            // plain LDS writes: a read of the wavefront's own spare accumulator, awaited, comes back after every write it queued
            if (!loader) {
                const float flushed = reinterpret_cast<volatile float*>(ys)[nrows + wave];
                asm volatile("" ::"v"(flushed));
            }
            __syncthreads();
            timeline_stamp<kAblate>(block_no, wave, lane, 3);
            if (!(kAblate & 32)) for (uint32_t i = tid; i < nrows; i += kThreads) out[out0 + i] = reinterpret_cast<uint32_t*>(ys)[i];     // fp32 bits, or the saturated Q8.24 sum
            timeline_stamp<kAblate>(block_no, wave, lane, 4);
            if (!next) break;
            continue;
        }
        if (!loader) {
            const acc_t flushed = atomicAdd(ys + nrows, static_cast<acc_t>(0));
            asm volatile("" ::"v"(flushed));
        }
        __syncthreads();
        timeline_stamp<kAblate>(block_no, wave, lane, 3);
        // the accumulators are final (no barrier after the store: the last block's stores drain while the workgroup retires)
        if (!(kAblate & 32)) for (uint32_t i = tid; i < nrows; i += kThreads) out[out0 + i] = Rows<kFloat>::finish(ys[i]);
        timeline_stamp<kAblate>(block_no, wave, lane, 4);
        if (!next) break;
    }
}

// Column-sliced matrices: y[r] = sum over slices of the per-slice partial results.  Fixed point: each partial is
// already clamped to 2^32-1 and min(sum, MAX) == min(sum of min(part, MAX), MAX) for non-negative parts, so the
// result is still exactly the saturating sum of the PE (pe.h:72).
// Iterative callers: scale (*) y (+) shift in the numeric mode's own arithmetic (hisparse_hip.h, hs_feedback).
template <bool kFloat>
__device__ __forceinline__ uint32_t feedback_word(uint32_t y, uint32_t scale, uint32_t shift) {
    if (kFloat) {
        const float p = __uint_as_float(scale) * __uint_as_float(y);   // -ffp-contract=off: multiply, then add
        return __float_as_uint(p + __uint_as_float(shift));
    }
    const uint64_t s = static_cast<uint64_t>(q8_24_mul(scale, y)) + shift;
    return s > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(s);
}

// x_fb != nullptr: also feeds the combined rows below n_fb back into x (one launch less per iteration of hs_iterate).
// Four rows per thread (row counts, partition bounds and n_fb are multiples of 8): 16-byte loads and stores.
// kSlices: the number of partial vectors (2 .. kMaxSweepSlices), a template parameter so that all kSlices loads of a thread are in flight
// together -- with a run-time loop they went out one memory round trip after the other (4.9 us for ogbl-ppa's 4 x 2.3 MB).
template <bool kFloat, int kSlices>
__global__ __launch_bounds__(256) void combine_slices_kernel(const uint32_t* __restrict__ partial, uint32_t* __restrict__ y,
                                                             uint32_t num_rows, uint32_t row_lo, uint32_t row_hi,
                                                             uint32_t* __restrict__ x_fb, uint32_t n_fb, uint32_t scale, uint32_t shift) {
    const uint32_t r = row_lo + (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    if (r >= row_hi) return;
    uint4 p[kSlices];
#pragma unroll
    for (int k = 0; k < kSlices; ++k) p[k] = *reinterpret_cast<const uint4*>(partial + static_cast<size_t>(k) * num_rows + r);
    uint32_t word[4];
    if (kFloat) {
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < kSlices; ++k) {
            s[0] += __uint_as_float(p[k].x); s[1] += __uint_as_float(p[k].y); s[2] += __uint_as_float(p[k].z); s[3] += __uint_as_float(p[k].w);
        }
        for (int j = 0; j < 4; ++j) word[j] = __float_as_uint(s[j]);
    } else {
        uint64_t s[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < kSlices; ++k) {
            s[0] += p[k].x; s[1] += p[k].y; s[2] += p[k].z; s[3] += p[k].w;
        }
        for (int j = 0; j < 4; ++j) word[j] = s[j] > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(s[j]);
    }
    *reinterpret_cast<uint4*>(y + r) = make_uint4(word[0], word[1], word[2], word[3]);
    if (x_fb && r < n_fb)
        *reinterpret_cast<uint4*>(x_fb + r) = make_uint4(feedback_word<kFloat>(word[0], scale, shift), feedback_word<kFloat>(word[1], scale, shift),
                                                         feedback_word<kFloat>(word[2], scale, shift), feedback_word<kFloat>(word[3], scale, shift));
}

template <bool kFloat>
__global__ __launch_bounds__(256) void feedback_kernel(const uint32_t* __restrict__ y, uint32_t* __restrict__ x, uint32_t n, uint32_t scale,
                                                       uint32_t shift) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = feedback_word<kFloat>(y[i], scale, shift);
}

// hs_push_result: the y slab into up to kMaxPushTargets other buffers -- on a multi-GPU node those are the peers' gather buffers, written
// over xGMI with plain stores (hipDeviceEnablePeerAccess): a gather of the row slabs without a collective on the critical path.
struct PushTargets { uint32_t* dst[kMaxPushTargets]; };
__global__ __launch_bounds__(256) void push_result_kernel(const uint32_t* __restrict__ y, PushTargets t, uint32_t n_dst, uint32_t words) {
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;      // words is a multiple of 4 (padded row counts)
    if (i >= words) return;
    const uint4 v = *reinterpret_cast<const uint4*>(y + i);
    for (uint32_t k = 0; k < n_dst; ++k) *reinterpret_cast<uint4*>(t.dst[k] + i) = v;
}


// ---- the LIGHT kernel (round 4): small matrices, one launch ------------------------------------------------------------------------------
// A matrix of a few million non-zeros (a pruned-NN layer below BITMAP's density, one rank's slab of a graph split 8 ways, the 1k x 1k
// plumbing case) is launch-bound in the row-block kernel: its 1024-thread workgroups ramp up over 2-4 us, x is staged through LDS although
// the whole vector sits in L2, and a column-sliced plan pays a second launch for the combine pass -- 10-16 us for 1-4 us of bytes.  This
// kernel runs the SAME PAIRS image (stream_tiles.h: chunks of 64 x { value word, local_row << 16 | local_col }, stored block by block in
// dealing order, i.e. the chunks of a block are contiguous and unit by unit) with
//   * 256-thread workgroups, up to six per CU, every wavefront a consumer of a CONSECUTIVE quarter of the block's chunks, kLightBatch (four) in flight;
//     a lane therefore walks consecutive sorted elements and sums in a register while its row stays the same;
//   * no x ring, no loader wavefronts, no unit barriers: x[col0(unit) + local_col] is a plain gather (the vector is L2-resident at this
//     size); a unit is only the place where col0 changes, found per chunk from the block's (<= kLightMaxUnits) unit ends held in registers;
//   * one column slice always, so y is written by the kernel itself: ONE launch per SpMV;
//   * 8-byte LDS accumulators as in the row-block kernel (exact 64-bit sums / double sums): the same arithmetic, bit for bit in fixed point.
// (Two other accumulation schemes were measured on one rank's slab of mouse_gene split 8 ways and lost: linear dealing with one LDS atomic
// per element -- up to 64 lanes on one accumulator -- and linear dealing with a DPP segmented scan per chunk, ~80 vector instructions for
// 64 elements: 17.3 / 17.5 us per SpMV against 10.8 for the row-block plan, profiles/r04_light_*.txt.)
// The plan (stream_tiles.cpp: "light") cuts up to 4 x CUs row ranges of equal non-zero count.  Reference: the cluster / PE arithmetic as
// for spmv_rowblock_kernel (pe.h:62-81, pe-pob.h:63-65); drain order spmv_result_drain.cpp:104-113.
template <bool kFloat, int kAblate>
__global__ __launch_bounds__(kLightThreads, 6) void spmv_light_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks,
                                                                    const Unit* __restrict__ units, const uint32_t* __restrict__ x,
                                                                    uint32_t* __restrict__ out, int32_t row_part_filter,
                                                                    const uint32_t* __restrict__ part_heads) {
    using R = Rows<kFloat>;
    using acc_t = typename R::acc_t;
    using sum_t = typename R::sum_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    acc_t* ys = reinterpret_cast<acc_t*>(lds);                    // [nrows + 1]
    constexpr uint32_t kWaves = kLightThreads / kWaveLanes;
    const uint32_t tid = threadIdx.x, lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    uint32_t bi = wg;
    if (row_part_filter >= 0) {
        bi = ((const __attribute__((address_space(4))) uint32_t*)part_heads)[static_cast<uint32_t>(row_part_filter) * gridDim.x + wg];
        if (bi == kNoBlock) return;
    }
    bool first_block = true;
    for (uint32_t next = 0;; bi = next) {
        const BlockTable blk = (BlockTable)(blocks + bi);
        next = (row_part_filter >= 0 && blk->next_part > static_cast<uint32_t>(row_part_filter)) ? 0u : blk->next;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset, ub = blk->unit_begin;
        const uint32_t U = blk->unit_end - ub;
        const uint8_t* chunks = scalar_pointer(image + blk->wave_offset[0]);      // chunk g of the block at g * 512 (scalar base + 32-bit offset: a block's stream stays below 4 GiB)
        // chunks of the block = the 14 per-wavefront step counts the image was dealt for, added up: known from the Block itself, so the
        // first batch of element loads goes out BEFORE the unit table is fetched (one dependent round trip less in a kernel that is
        // nothing but a chain of them)
        uint32_t total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kConsumerWaves; ++w) total += blk->total_steps[w];
        // wavefront w takes the CONSECUTIVE chunks [g_begin, g_end): in the strided layout of a PAIRS unit (slot (chunk c, lane l) = sorted
        // element l x chunks + c) a lane then walks consecutive sorted elements -- a run that stays on one row for many chunks
        const uint32_t per_wave = (total + kWaves - 1) / kWaves;
        const uint32_t g_begin = min(wave * per_wave, total), g_end = min(g_begin + per_wave, total);
        const uint32_t lane_off = lane * 8u;
        uint2 e[kLightBatch];
        auto load_batch = [&](uint2 (&dst)[kLightBatch], uint32_t g0) {
#pragma unroll
            for (int j = 0; j < kLightBatch; ++j) {
                const uint32_t g = min(g0 + j, g_end - 1);                          // past the end: the last chunk again (its products are dropped)
                dst[j] = *reinterpret_cast<const uint2*>(chunks + (g * kChunkBytes + lane_off));
            }
        };
        if (g_begin < g_end) load_batch(e, g_begin);
        if (!first_block) __syncthreads();                        // the previous block's store has read the accumulators
        first_block = false;
        // lane u < U of every wavefront: where unit u ends among the block's chunks (its 14 per-wavefront stream positions add up to that:
        // the image was dealt for the row-block kernel's consumer wavefronts) and its first column
        uint32_t my_end = 0, my_col0 = 0;
        if (lane < U) {
            const uint4* u4 = reinterpret_cast<const uint4*>(units + ub + lane);     // 64 bytes: col0, ncols, end_step[14]
            const uint4 a = u4[0], b = u4[1], c = u4[2], d = u4[3];
            my_col0 = a.x;
            my_end = a.z + a.w + b.x + b.y + b.z + b.w + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
        }
        for (uint32_t i = tid; i <= nrows; i += kLightThreads) ys[i] = 0;
        __syncthreads();
        // the unit the wavefront's first chunk lies in; from there on units are entered in order (scalar bookkeeping only)
        uint32_t u = 0, unit_end = 0, col0 = 0;
        if (g_begin < g_end) {
            u = static_cast<uint32_t>(__builtin_ctzll(__ballot(lane < U && g_begin < my_end)));
            unit_end = __builtin_amdgcn_readlane(my_end, u);
            col0 = __builtin_amdgcn_readlane(my_col0, u);
        }
        uint32_t lane_row = nrows;                                // the row this lane is summing (starts on the spare accumulator: adds 0 there)
        sum_t lane_sum = 0;
        for (uint32_t g0 = g_begin; g0 < g_end; g0 += kLightBatch) {
            uint32_t xv[kLightBatch];
#pragma unroll
            for (int j = 0; j < kLightBatch; ++j) {
                const uint32_t g = min(g0 + j, g_end - 1);
                while (g >= unit_end) {                                              // wave-uniform: the next unit with chunks
                    ++u;
                    unit_end = __builtin_amdgcn_readlane(my_end, u);
                    col0 = __builtin_amdgcn_readlane(my_col0, u);
                }
                xv[j] = (kAblate & 2) ? e[j].y : x[col0 + (e[j].y & 0xffffu)];
            }
            // products first (one 32-bit register each: the Q8.24 product, or the fp32 one), so that the element registers are free for
            // the next batch, whose loads travel while this one is added up
            typename OwnerOps<kFloat>::val_t prod[kLightBatch];
            uint32_t rows[kLightBatch];
#pragma unroll
            for (int j = 0; j < kLightBatch; ++j) {
                prod[j] = OwnerOps<kFloat>::product(e[j].x, xv[j]);
                rows[j] = e[j].y >> 16;
            }
            if (g0 + kLightBatch < g_end) load_batch(e, g0 + kLightBatch);
#pragma unroll
            for (int j = 0; j < kLightBatch; ++j) {
                if (g0 + j >= g_end) break;                                          // wave-uniform
                if (kAblate & 1) { asm volatile("" ::"v"(prod[j]), "v"(rows[j])); continue; }
                // a lane sums in a register while its row stays the same and touches the LDS accumulator only when it changes
                if (rows[j] != lane_row) {                                           // per lane
                    R::add_sum(ys, lane_row, lane_sum);
                    lane_row = rows[j];
                    lane_sum = 0;
                }
                lane_sum += R::widen(static_cast<typename R::prod_t>(prod[j]));
            }
        }
        // hand the last sums over: a block of ONE long row (a pruned-NN layer) leaves all 64 lanes on the same row -- DPP total, one add
        if (!(kAblate & 1)) {
            const uint32_t row0 = __builtin_amdgcn_readfirstlane(lane_row);
            if (__ballot(lane_row != row0) == 0) {
                const sum_t sum = wave_total_in_lane63(lane_sum);
                if (lane == kWaveLanes - 1) R::add_sum(ys, row0, sum);
            } else {
                R::add_sum(ys, lane_row, lane_sum);
            }
        }
        // no-return LDS atomics can outlive lgkmcnt(0) (spmv_rowblock_kernel): a returning one on the spare accumulator, awaited, cannot
        const acc_t flushed = atomicAdd(ys + nrows, static_cast<acc_t>(0));
        asm volatile("" ::"v"(flushed));
        __syncthreads();
        if (!(kAblate & 32)) for (uint32_t i = tid; i < nrows; i += kLightThreads) out[out0 + i] = R::finish(ys[i]);
        if (!next) break;
    }
}

template <bool kFloat, int kRing, int kAblate, int kDepth, bool kOwner = false>
hipError_t configure_one(uint32_t lds_bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_rowblock_kernel<kFloat, kRing, kAblate, kDepth, kOwner>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
}

int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}

}  // namespace

bool profiling_depth_given() {
#ifdef HISPARSE_PROFILING
    return std::getenv("HISPARSE_DEPTH") != nullptr;
#else
    return false;
#endif
}
bool profiling_switches(int& ablate, int& depth) {
#ifdef HISPARSE_PROFILING
    ablate = env_int("HISPARSE_ABLATE", 0);
    depth = env_int("HISPARSE_DEPTH", 8);
    return true;
#else
    ablate = 0;
    depth = 8;
    return profiling_switch_error() == nullptr;
#endif
}
const char* profiling_switch_error() {
#ifdef HISPARSE_PROFILING
    return nullptr;
#else
    if (env_int("HISPARSE_ABLATE", 0) != 0)
        return "HISPARSE_ABLATE is set: the ablation builds (wrong results by design) live in libhisparse_hip_prof.so (make HISPARSE_PROFILING=1, "
               "HISPARSE_HIP_LIB=...); this library will not launch with it in the environment";
    if (env_int("HISPARSE_DEPTH", 8) != 8)      // (the default, spelled out, selects nothing)
        return "HISPARSE_DEPTH is set: prefetch-depth experiments live in libhisparse_hip_prof.so (make HISPARSE_PROFILING=1); this library will "
               "not launch with it in the environment";
    return nullptr;
#endif
}

// LDS plan: row accumulators first (64-bit integer sums / double sums + 1 spare; OWNER: floats + one spare per consumer
// wavefront), then the ring of x buffers.
uint32_t spmv_light_lds_bytes(uint32_t max_block_rows) { return ((max_block_rows + 1) * kAccumulatorBytes + 15u) & ~15u; }
uint32_t spmv_lds_bytes(uint32_t max_block_rows, uint32_t ring_buffers, uint32_t format) {
    const uint32_t acc = (format == kFormatOwner || format == kFormatOwner24) ? (max_block_rows + kConsumerWaves) * kOwnerAccumulatorBytes : (max_block_rows + 1) * kAccumulatorBytes;
    return ((acc + 15u) & ~15u) + ring_buffers * kBufBytes;
}

// The PRODUCT library (libhisparse_hip.so) carries one instantiation per (numeric mode, stream format) and nothing else.  The ablation
// and timeline instantiations -- most of which give WRONG results by design -- exist only in libhisparse_hip_prof.so
// (`make HISPARSE_PROFILING=1`, -DHISPARSE_PROFILING), which tools/ load through HISPARSE_HIP_LIB; in the product library a set
// HISPARSE_ABLATE / HISPARSE_DEPTH makes every launch fail (hs_run: HS_ERR_BAD_ARG) instead of being obeyed or silently ignored.
#ifdef HISPARSE_PROFILING
// OWNER24 profiling builds (three records in flight): 1 = no LDS work, 4 = no x refill, 8 = no unit barriers, and combinations
#define HS_FOR_EACH_OWNER24_ABLATION(X) X(0) X(1) X(4) X(5) X(8) X(12) X(13) X(512)
#define HS_FOR_EACH_LIGHT_VARIANT(X) X(0) X(1) X(2) X(3) X(32) X(35)
#define HS_FOR_EACH_OWNER24_DEPTH(X) X(2) X(3) X(4)
#define HS_FOR_EACH_OWNER24_FIXED(X) X(0) X(512)
// OWNER variants (float only): ablate values as for the other formats
#define HS_FOR_EACH_OWNER_VARIANT(X) X(0) X(1) X(2) X(3) X(4) X(7) X(8) X(11) X(12) X(15) X(127) X(256)
#else
#define HS_FOR_EACH_OWNER24_ABLATION(X) X(0)
#define HS_FOR_EACH_LIGHT_VARIANT(X) X(0)
#define HS_FOR_EACH_OWNER24_DEPTH(X) X(3)
#define HS_FOR_EACH_OWNER24_FIXED(X) X(0)
#define HS_FOR_EACH_OWNER_VARIANT(X) X(0)
#endif

// (float, ring format, ablate, depth): the product variants first, then the profiling builds (fixed point only)
// (ring 4 / 5: PAIRS / DELTA images that stay in the Infinity Cache -- the same kernels with `sc1` stream loads instead of `nt`)
#define HS_FOR_EACH_PRODUCT_VARIANT(X)                                                                           \
    X(true, 0, 0, 8) X(true, 1, 0, 8) X(false, 0, 0, 8) X(false, 1, 0, 8) X(true, 2, 0, 8) X(false, 2, 0, 8)      \
    X(true, 4, 0, 8) X(true, 5, 0, 8) X(false, 4, 0, 8) X(false, 5, 0, 8)
#ifndef HISPARSE_PROFILING
#define HS_FOR_EACH_VARIANT(X) HS_FOR_EACH_PRODUCT_VARIANT(X)
#else
#define HS_FOR_EACH_VARIANT(X)                                                                                   \
    HS_FOR_EACH_PRODUCT_VARIANT(X)                                                                               \
    X(false, false, 0, 16) X(false, false, 3, 8) X(false, true, 3, 8)                       \
    X(false, false, 4, 8) X(false, true, 4, 8) X(false, false, 8, 8) X(false, true, 8, 8) X(false, false, 15, 8) X(false, true, 15, 8) X(false, false, 31, 8) X(false, true, 31, 8)                     \
    X(false, false, 47, 8) X(false, true, 47, 8) X(false, false, 79, 8) X(false, true, 79, 8)                     \
    X(false, false, 127, 8) X(false, true, 127, 8)                                                                 \
    X(true, false, 3, 8) X(true, false, 4, 8) X(true, false, 8, 8) X(true, false, 15, 8) X(true, false, 127, 8) \
    X(false, false, 512, 8) X(false, true, 512, 8) X(true, false, 512, 8) X(true, true, 512, 8)
#endif

hipError_t configure_spmv_kernels(uint32_t lds_bytes) {
    hipError_t e;
#define X(F, T, A, D) if ((e = configure_one<F, T, A, D>(lds_bytes)) != hipSuccess) return e;
    HS_FOR_EACH_VARIANT(X)
#undef X
#define X(A) if ((e = configure_one<true, false, A, 8, true>(lds_bytes)) != hipSuccess) return e;
    HS_FOR_EACH_OWNER_VARIANT(X)
#undef X
#define X(D) if (D != 3 && (e = configure_one<true, 3, 0, D, true>(lds_bytes)) != hipSuccess) return e;
    HS_FOR_EACH_OWNER24_DEPTH(X)
#undef X
#define X(A) if ((e = configure_one<true, 3, A, 3, true>(lds_bytes)) != hipSuccess) return e;
    HS_FOR_EACH_OWNER24_ABLATION(X)
#undef X
#define X(A) if ((e = configure_one<false, 3, A, 3, true>(lds_bytes)) != hipSuccess) return e;
    HS_FOR_EACH_OWNER24_FIXED(X)
#undef X
    if ((e = configure_sweep_kernels(lds_bytes)) != hipSuccess) return e;
    return configure_bitmap_kernels(lds_bytes);
}

hipError_t launch_spmv(bool is_float, const SpmvLaunch& a, hipStream_t stream) {
    if (a.num_workgroups == 0) return hipSuccess;
    if (a.format == kFormatBitmap) return launch_spmv_bitmap(is_float, a, stream);
    if (a.format == kFormatSweep) return launch_spmv_sweep(is_float, a, stream);
    if (a.light) {
        if (a.format != kFormatPairs) return hipErrorInvalidValue;
        int ablate = 0, depth_unused = 8;
        if (!profiling_switches(ablate, depth_unused)) return hipErrorInvalidValue;
        const dim3 grid(a.num_workgroups), block(kLightThreads);
#define X(A)                                                                                                                                  \
    if (ablate == A) {                                                                                                                        \
        if (is_float) hipLaunchKernelGGL((spmv_light_kernel<true, A>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.x, a.out, a.row_part_filter, a.part_heads); \
        else hipLaunchKernelGGL((spmv_light_kernel<false, A>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.x, a.out, a.row_part_filter, a.part_heads);       \
        return hipGetLastError();                                                                                                             \
    }
        HS_FOR_EACH_LIGHT_VARIANT(X)
#undef X
        return hipErrorInvalidValue;
    }
    int ring = a.format == kFormatDelta ? 1 : a.format == kFormatPairs24 ? 2 : 0;
    const dim3 grid(a.num_workgroups), block(kThreads);
    const uint32_t x_base = a.lds_bytes - a.ring_buffers * kBufBytes;
    const CarriedCombine carry = carried(a);
    // profiling aids (libhisparse_hip_prof.so only): HISPARSE_ABLATE removes parts of the work (wrong results), HISPARSE_DEPTH picks the
    // prefetch depth; read per launch, a tool may change them between launches.  The product library refuses to run with either set.
    int ablate = 0, depth = 8;
    if (!profiling_switches(ablate, depth)) return hipErrorInvalidValue;
    // PAIRS / DELTA images the plan found resident in the Infinity Cache: the `sc1` instantiation (Ring<kRing | 4>); OWNER / OWNER24 / PAIRS24
    // images and the profiling builds keep `nt`
    if (a.stream_resident && ring < 2 && ablate == 0 && depth == 8 && (a.format == kFormatPairs || a.format == kFormatDelta)) ring |= 4;
    bool launched = false;
    // timeline build: HISPARSE_ABLATE=512 HISPARSE_TIMELINE_OUT=file -> the launch is synchronised and its timestamps (workgroups x
    // kTimelineBlocks x 2 wavefronts x kTimelineStamps u64, 100 MHz) overwrite the file (tools/rowblock_timeline.py)
    struct TimelineDump {
        hipStream_t stream; uint32_t workgroups; bool on;
        ~TimelineDump() {
            const char* path = std::getenv("HISPARSE_TIMELINE_OUT");
            if (!on || !path || !buffer()) return;
            (void)hipStreamSynchronize(stream);
            std::vector<uint64_t> host(size_t(workgroups) * kTimelineBlocks * 2 * kTimelineStamps);
            (void)hipMemcpy(host.data(), buffer(), host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
            if (FILE* f = std::fopen(path, "wb")) { std::fwrite(host.data(), sizeof(uint64_t), host.size(), f); std::fclose(f); }
        }
        static uint64_t*& buffer() {      // one per device (the device symbol is per device too)
            static uint64_t* b[16] = {};
            int dev = 0;
            (void)hipGetDevice(&dev);
            return b[dev >= 0 && dev < 16 ? dev : 0];
        }
    } timeline_dump{stream, a.num_workgroups, ablate == 512 && a.num_workgroups <= 4096};
    if (timeline_dump.on) {
        const size_t bytes = size_t(4096) * kTimelineBlocks * 2 * kTimelineStamps * sizeof(uint64_t);
        if (!TimelineDump::buffer() && hipMalloc(reinterpret_cast<void**>(&TimelineDump::buffer()), bytes) == hipSuccess)
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rowblock_timeline), &TimelineDump::buffer(), sizeof(uint64_t*));
        if (TimelineDump::buffer()) (void)hipMemsetAsync(TimelineDump::buffer(), 0, bytes, stream);
    }
    if (a.format == kFormatOwner24) {
        // records in flight per wavefront (1792 bytes each): HISPARSE_DEPTH=2|3|4 for experiments, kOwner24Depth otherwise
        const int records = profiling_depth_given() ? depth : kOwner24Depth;
        if (!is_float) {      // fixed point: saturating 32-bit accumulators (OwnerOps<false>); the product build and the timeline build
            if (records != 3) return hipErrorInvalidValue;
#define X(A)                                                                                                                       \
    if (ablate == A) {                                                                                                             \
        hipLaunchKernelGGL((spmv_rowblock_kernel<false, 3, A, 3, true>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.x, a.out, \
                           a.row_part_filter, a.ring_buffers, x_base, a.part_heads, carry);                                               \
        return hipGetLastError();                                                                                                  \
    }
            HS_FOR_EACH_OWNER24_FIXED(X)
#undef X
            return hipErrorInvalidValue;
        }
#define X(A)                                                                                                                       \
    if (ablate == A && records == 3) {                                                                                             \
        hipLaunchKernelGGL((spmv_rowblock_kernel<true, 3, A, 3, true>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.x, a.out, \
                           a.row_part_filter, a.ring_buffers, x_base, a.part_heads, carry);                                               \
        return hipGetLastError();                                                                                                  \
    }
        HS_FOR_EACH_OWNER24_ABLATION(X)
#undef X
        if (ablate != 0) return hipErrorInvalidValue;
#define X(D)                                                                                                                       \
    if (records == D) {                                                                                                            \
        hipLaunchKernelGGL((spmv_rowblock_kernel<true, 3, 0, D, true>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.x, a.out, \
                           a.row_part_filter, a.ring_buffers, x_base, a.part_heads, carry);                                               \
        return hipGetLastError();                                                                                                  \
    }
        HS_FOR_EACH_OWNER24_DEPTH(X)
#undef X
        return hipErrorInvalidValue;
    }
    if (a.format == kFormatOwner) {
        if (!is_float) return hipErrorInvalidValue;
        // profiling build: HISPARSE_ABLATE=256 HISPARSE_TIMELINE_OUT=file -> per-wavefront clock totals, accumulated over the
        // launches, rewritten after every launch (tools/owner_profile.py)
        static uint64_t* profile = nullptr;
        if (ablate == 256 && !profile) {
            (void)hipMalloc(reinterpret_cast<void**>(&profile), size_t(4096) * kWavesPerWorkgroup * 8 * sizeof(uint64_t));
            (void)hipMemset(profile, 0, size_t(4096) * kWavesPerWorkgroup * 8 * sizeof(uint64_t));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_owner_profile), &profile, sizeof(profile));
        }
#define X(A)                                                                                                                      \
    if (!launched && ablate == A) {                                                                                               \
        hipLaunchKernelGGL((spmv_rowblock_kernel<true, false, A, 8, true>), grid, block, a.lds_bytes, stream, a.image, a.blocks,  \
                           a.units, a.x, a.out, a.row_part_filter, a.ring_buffers, x_base, a.part_heads, carry);                        \
        launched = true;                                                                                                          \
    }
        if (depth == 8) {
        HS_FOR_EACH_OWNER_VARIANT(X)
        }
#undef X
        if (ablate == 256 && profile && a.num_workgroups <= 4096) {
            if (const char* path = std::getenv("HISPARSE_TIMELINE_OUT")) {
                (void)hipStreamSynchronize(stream);
                std::vector<uint64_t> host(size_t(a.num_workgroups) * kWavesPerWorkgroup * 8);
                (void)hipMemcpy(host.data(), profile, host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
                if (FILE* f = std::fopen(path, "wb")) {
                    std::fwrite(host.data(), sizeof(uint64_t), host.size(), f);
                    std::fclose(f);
                }
            }
        }
        return launched ? hipGetLastError() : hipErrorInvalidValue;
    }
#define X(F, T, A, D)                                                                                                           \
    if (!launched && is_float == F && ring == int(T) && ablate == A && depth == D) {                                     \
        hipLaunchKernelGGL((spmv_rowblock_kernel<F, T, A, D>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units,    \
                           a.x, a.out, a.row_part_filter, a.ring_buffers, x_base, a.part_heads, carry);                   \
        launched = true;                                                                                                        \
    }
    HS_FOR_EACH_VARIANT(X)
#undef X
    if (!launched) return hipErrorInvalidValue;   // unknown HISPARSE_ABLATE / HISPARSE_DEPTH combination
    return hipGetLastError();
}

hipError_t launch_combine_slices(bool is_float, const uint32_t* partial, uint32_t* y, uint32_t num_rows, uint32_t slices, uint32_t row_lo,
                                 uint32_t row_hi, hipStream_t stream, uint32_t* x_fb, uint32_t n_fb, uint32_t scale, uint32_t shift) {
    if (row_hi <= row_lo) return hipSuccess;
    const dim3 grid((row_hi - row_lo + 1023) / 1024), block(256);   // four rows per thread
    switch (slices) {
#define X(N)                                                                                                                                        \
    case N:                                                                                                                                         \
        if (is_float) hipLaunchKernelGGL((combine_slices_kernel<true, N>), grid, block, 0, stream, partial, y, num_rows, row_lo, row_hi, x_fb, n_fb, scale, shift); \
        else hipLaunchKernelGGL((combine_slices_kernel<false, N>), grid, block, 0, stream, partial, y, num_rows, row_lo, row_hi, x_fb, n_fb, scale, shift);         \
        break;
        X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
#undef X
        default: return hipErrorInvalidValue;
    }
    static_assert(kMaxColSlices <= 16 && kMaxSweepSlices == 16, "combine_slices_kernel is instantiated for 2 .. 16 slices");
    return hipGetLastError();
}

hipError_t launch_push_result(const uint32_t* y, void* const* dst, uint32_t n_dst, uint32_t words, hipStream_t stream) {
    if (n_dst == 0 || words == 0) return hipSuccess;
    if (n_dst > kMaxPushTargets || (words & 3u)) return hipErrorInvalidValue;
    PushTargets t{};
    for (uint32_t k = 0; k < n_dst; ++k) t.dst[k] = static_cast<uint32_t*>(dst[k]);
    hipLaunchKernelGGL(push_result_kernel, dim3((words / 4 + 255) / 256), dim3(256), 0, stream, y, t, n_dst, words);
    return hipGetLastError();
}

hipError_t launch_feedback(bool is_float, const uint32_t* y, uint32_t* x, uint32_t n, uint32_t scale, uint32_t shift, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const dim3 grid((n + 255) / 256), block(256);
    if (is_float) hipLaunchKernelGGL(feedback_kernel<true>, grid, block, 0, stream, y, x, n, scale, shift);
    else hipLaunchKernelGGL(feedback_kernel<false>, grid, block, 0, stream, y, x, n, scale, shift);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
