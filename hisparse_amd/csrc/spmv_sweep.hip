// spmv_sweep.hip — the SpMV kernel for SWEEP images (stream_tiles.h "SWEEP format"; builder: sweep_tiles.cpp).  gfx950.
//
// Same architecture as spmv_rowblock_kernel -- a workgroup owns a row range, the sums live in its LDS (the cluster's output buffer,
// pe.h:121-135), the matrix streams past (spmv_cluster.h:73-98) -- with ONE difference: the vector is not staged.  The FPGA keeps the
// current column partition of x in on-chip banks (vecbuf_access_unit.h:66-72,126-128); the row-block kernel does the same with 8192-column
// sub-tiles in LDS, and on a hyper-sparse matrix pays ~2 100 clocks per (row range x sub-tile) unit for a flush, a barrier and a refill
// that ~2 000 elements cannot hide.  Here x stays in L2 and every lane fetches its own word:
//   * the block's elements come in (column, row) order, chunk k to wavefront k % 8 as its step k / 8: the 8 wavefronts (512 threads; 16 were
//     measured 4 % slower on pokec, 4 much slower in fixed point: profiles/r04_sweep_waves.txt) move over the
//     block's column slice together, once, and the 64 lanes of one gather touch a handful of 128-byte lines;
//   * per wavefront FOUR 512-byte chunks and FOUR gathers are in flight (kSweepDepth; eight of each until round 5 -- see HS_SWEEP_DEPTH below), both in
//     accumulator registers behind ONE counted wait per step: the gather for the chunk taken at step s is issued just before the load of chunk
//     s + kSweepDepth (vmcnt retires in order), see sweep_step();
//   * products go to LDS accumulators with atomics (any lane may hit any row): float -- double sums of the fp32 products (ds_add_f64), rounded
//     once; fixed point -- a wrapping 32-bit sum of the rounded Q8.24 products (ds_add_rtn_u32) plus one carry bit per row, i.e. exactly the
//     saturating sum (SweepRows below) -- the arithmetic of the other formats, bit for bit in fixed point;
//   * nothing between the block's prologue and its epilogue: no units, no barriers, no refills.
// Measured at block level before any builder existed (tools/gather_bench.hip, profiles/r04_gather_bench.txt): the stream runs at 5.3 TB/s
// and a gathered line of x costs ~3.3 clocks per CU.
#include <hip/hip_runtime.h>

#include <utility>

#include "spmv_device.h"
#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr int kSweepThreads = kSweepWaves * kWaveLanes;
#ifndef HS_SWEEP_DEPTH
#define HS_SWEEP_DEPTH 4            // round 5, measured again on the kernel alone (pokec, fixed / float_pob, profiles/r05_sweep_ring_depth.txt): 2: 83.6 / 78.0 us,
                                    // 3: 65.4 / 69.2, 4: 61.4 / 70.5, 6: 64.2 / 71.7, 8: 64.6 / 74.1, 12: 68.2 / 74.6, 16: 70.5 / 77.9 -- and 16 wavefronts x 2 deep = 8 x 4
                                    // deep (61.2 / 70.4): what counts is 16 KB of stream in flight per CU (beyond it, presumably, the gathers' lines of x lose the 32 KB L1: not isolated).
                                    // (round 4 had read "flat from 4 to 8" on whole steps of the 8-byte-accumulator kernel: profiles/r04_sweep_ring_depth.txt)
#endif
constexpr int kSweepDepth = HS_SWEEP_DEPTH;      // chunks (and gathers) in flight per wavefront

__device__ __forceinline__ const uint8_t* sweep_scalar_pointer(const void* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));
    return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// Ring: chunk slot K in a[2K : 2K+1] (value word, position word), gather slot K in a[2 kSweepDepth + K].  hipcc never allocates accumulator registers
// in this kernel; every asm statement that issues into the ring names all of them as clobbered, so nothing else is scheduled across.
#define HS_SWEEP_RING8 "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
                       "a18", "a19", "a20", "a21", "a22", "a23"
#if HS_SWEEP_DEPTH == 2
#define HS_SWEEP_RING "a0", "a1", "a2", "a3", "a4", "a5"
#elif HS_SWEEP_DEPTH == 3
#define HS_SWEEP_RING "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8"
#elif HS_SWEEP_DEPTH == 4
#define HS_SWEEP_RING "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11"
#elif HS_SWEEP_DEPTH == 6
#define HS_SWEEP_RING "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17"
#elif HS_SWEEP_DEPTH == 8
#define HS_SWEEP_RING HS_SWEEP_RING8
#elif HS_SWEEP_DEPTH == 12
#define HS_SWEEP_RING HS_SWEEP_RING8, "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35"
#elif HS_SWEEP_DEPTH == 16
#define HS_SWEEP_RING HS_SWEEP_RING8, "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", \
                      "a42", "a43", "a44", "a45", "a46", "a47"
#else
#error "HS_SWEEP_DEPTH must be 2, 3, 4, 6, 8, 12 or 16"
#endif

// kNt: the non-temporal hint on the stream loads -- right for an image that cannot stay in the Infinity Cache anyway, wrong for one that can
// (SpmvLaunch::stream_resident: `sc1` instead, the policy that measured best of five, profiles/r05_sweep_stream_policy.txt)
template <int K, bool kNt>
__device__ __forceinline__ void sweep_issue_chunk(const uint8_t* base, uint32_t off) {
    if constexpr (kNt) asm volatile("s_nop 4\n\tglobal_load_dwordx2 a[%0:%1], %2, %3 nt" ::"n"(2 * K), "n"(2 * K + 1), "v"(off), "s"(base) : "memory", HS_SWEEP_RING);
    else asm volatile("s_nop 4\n\tglobal_load_dwordx2 a[%0:%1], %2, %3 sc1" ::"n"(2 * K), "n"(2 * K + 1), "v"(off), "s"(base) : "memory", HS_SWEEP_RING);
}
template <int K>
__device__ __forceinline__ void sweep_issue_gather(const uint8_t* x, uint32_t byte_off) {
    asm volatile("s_nop 4\n\tglobal_load_dword a[%0], %1, %2" ::"n"(2 * kSweepDepth + K), "v"(byte_off), "s"(x) : "memory", HS_SWEEP_RING);
}
// one counted wait: chunk slot K (issued kSweepDepth steps ago) AND the gather issued just before it have landed
template <int K, bool kNoGather = false>
__device__ __forceinline__ void sweep_take(uint32_t& value, uint32_t& where, uint32_t& xv) {
    asm volatile("s_waitcnt vmcnt(%6)\n\tv_accvgpr_read_b32 %0, a[%3]\n\tv_accvgpr_read_b32 %1, a[%4]\n\tv_accvgpr_read_b32 %2, a[%5]"
                 : "=v"(value), "=v"(where), "=v"(xv) : "n"(2 * K), "n"(2 * K + 1), "n"(2 * kSweepDepth + K), "n"((kNoGather ? 1 : 2) * (kSweepDepth - 1)) : "memory");
}

// Row accumulators.  Float: doubles, ds_add_f64 (ds_add_f32 runs at a ninth of its rate on this part, stream_tiles.h) -- 8 bytes per row.
// Fixed point: 4 bytes per row.  The saturating sum of unsigned products is min(exact sum, 2^32 - 1), so a wrapping 32-bit sum plus ONE BIT
// "a carry happened" is exact: ds_add_rtn_u32 returns the old value, old + p < old is the carry, and the (rare) carry sets the row's bit in a
// bitmap behind the accumulators with ds_or_b32.  Twice the rows per block of the 8-byte form = half the row ranges = half the lines of x
// gathered per SpMV (sweep_tiles.cpp), which is what the format's cost is made of.
struct SweepLane {
    uint32_t value[kSweepDepth], row[kSweepDepth];      // the elements whose x words are on their way
    uint32_t carry_old = 0, carry_p = 0, carry_row = 0; // fixed point: the last add, whose carry is looked at one step later
};

template <bool kFloat>
struct SweepRows;
template <>
struct SweepRows<true> {
    using acc_t = double;
    static __device__ __forceinline__ uint32_t lds_words(uint32_t nrows) { return (nrows + 1) * 2; }
    static __device__ __forceinline__ void add(uint8_t* lds, uint32_t, SweepLane&, uint32_t row, uint32_t value, uint32_t xv) {
        atomicAdd(reinterpret_cast<double*>(lds) + row, static_cast<double>(__uint_as_float(value) * __uint_as_float(xv)));
    }
    static __device__ __forceinline__ void settle(uint8_t*, uint32_t, SweepLane&) {}
    static __device__ __forceinline__ uint32_t finish(const uint8_t* lds, uint32_t, uint32_t row) {
        return __float_as_uint(static_cast<float>(reinterpret_cast<const double*>(lds)[row]));
    }
};
template <>
struct SweepRows<false> {
    static __device__ __forceinline__ uint32_t flag_word0(uint32_t nrows) { return nrows + 1; }      // the carry bitmap starts behind the nrows + 1 sums
    static __device__ __forceinline__ uint32_t lds_words(uint32_t nrows) { return nrows + 1 + (nrows + 32) / 32; }
    // The carry of an add is looked at ONE STEP LATER (settle), when the returning atomic has long come back: checked on the spot, the
    // wavefront would sit out an LDS round trip in every step (tools/lds_atomic_bench.hip: 4.0 against 4.7 lanes/clk; it matters once a
    // workgroup has only 8 wavefronts to hide it behind).
    static __device__ __forceinline__ void settle(uint8_t* lds, uint32_t nrows, SweepLane& st) {
        if (st.carry_old + st.carry_p < st.carry_old)                                             // AP_SAT (pe.h:72): once beyond 2^32 - 1, always
            atomicOr(reinterpret_cast<uint32_t*>(lds) + flag_word0(nrows) + (st.carry_row >> 5), 1u << (st.carry_row & 31u));
    }
    static __device__ __forceinline__ void add(uint8_t* lds, uint32_t nrows, SweepLane& st, uint32_t row, uint32_t value, uint32_t xv) {
        settle(lds, nrows, st);
        st.carry_p = q8_24_mul(value, xv);
        st.carry_row = row;
        st.carry_old = atomicAdd(reinterpret_cast<uint32_t*>(lds) + row, st.carry_p);             // ds_add_rtn_u32
    }
    static __device__ __forceinline__ uint32_t finish(const uint8_t* lds, uint32_t nrows, uint32_t row) {
        const uint32_t* acc = reinterpret_cast<const uint32_t*>(lds);
        return ((acc[flag_word0(nrows) + (row >> 5)] >> (row & 31u)) & 1u) ? 0xffffffffu : acc[row];
    }
};


// Step s of a wavefront (ring slot K = s % D, D = kSweepDepth).  In flight on entry, oldest first: gather(s - D), chunk(s), gather(s - D + 1), chunk(s + 1), ...
// Waiting until 2 (D - 1) loads are left means chunk(s) and the gather before it have landed: add the element taken at step s - D (its x word
// has just arrived), keep chunk(s)'s element, ask for ITS x word and for chunk(s + D).
// kAblate (libhisparse_hip_prof.so only, WRONG results): 1 = no LDS accumulation, 2 = the gather reads one line near the chunk's base
// instead of the elements' columns (keeps the wait count), 4 = no zeroing of the accumulators and no result store (the block's prologue and
// epilogue), 8 = no gather at all (one load per step, the wait count halved)
template <bool kFloat, int kAblate, bool kNt, int K>
__device__ __forceinline__ void sweep_step(SweepLane& st, const uint8_t* stream, const uint8_t* x, uint32_t s, uint32_t steps, uint32_t lane_off, uint32_t base,
                                           uint8_t* ys, uint32_t nrows) {
    uint32_t value, where, xv;
    sweep_take<K, (kAblate & 8) != 0>(value, where, xv);
    if (!(kAblate & 1)) SweepRows<kFloat>::add(ys, nrows, st, st.row[K], st.value[K], xv);      // (the first eight steps add 0 x x[..] to the spare accumulator)
    else asm volatile("" ::"v"(xv), "v"(st.value[K]), "v"(st.row[K]));
    st.value[K] = value;
    st.row[K] = where >> 16;
    if (kAblate & 8) {}
    else if (!(kAblate & 2)) sweep_issue_gather<K>(x, (base + (where & 0xffffu)) * 4u);
    else if (kAblate & 16) sweep_issue_gather<K>(x, lane_off >> 1 & 127u);       // (16, with 2: always the same line of x)
    else sweep_issue_gather<K>(x, (base * 4u & ~127u) + (lane_off >> 1 & 127u));
    sweep_issue_chunk<K, kNt>(stream, min(s + kSweepDepth, steps - 1) * (kSweepWaves * kChunkBytes) + lane_off);
}

// prime, in the steady-state order: gather K (a dummy: its word is multiplied by 0), then chunk K
template <bool kNoGather, bool kNt, int... Ks>
__device__ __forceinline__ void sweep_prime(std::integer_sequence<int, Ks...>, const uint8_t* stream, const uint8_t* x, uint32_t pad_col, uint32_t steps, uint32_t lane_off) {
    (((kNoGather ? (void)0 : sweep_issue_gather<Ks>(x, pad_col * 4u)), sweep_issue_chunk<Ks, kNt>(stream, min(uint32_t(Ks), steps - 1) * (kSweepWaves * kChunkBytes) + lane_off)), ...);
}
// one round of eight steps; steps at or beyond `end` are skipped (wave-uniform)
template <bool kFloat, int kAblate, bool kNt, int... Ks>
__device__ __forceinline__ void sweep_round(std::integer_sequence<int, Ks...>, SweepLane& st, const uint8_t* stream, const uint8_t* x, uint32_t s0, uint32_t steps,
                                            uint32_t end, uint32_t lane_off, const uint32_t (&b)[kSweepDepth], uint8_t* ys, uint32_t nrows) {
    ((s0 + Ks < end ? sweep_step<kFloat, kAblate, kNt, Ks>(st, stream, x, s0 + Ks, steps, lane_off, b[Ks], ys, nrows) : (void)0), ...);
}

// A block's prologue and epilogue.  Up to 39 716 rows on 512 threads is ~78 trips per thread: one ds_write_b32 per trip, and -- in the epilogue --
// a read of the row's carry word, a branch, a read of its sum and a 4-byte store per trip, each waiting for the one before, measured 9-10 us of
// pokec's 67 (profiles/r05_sweep_ablate.txt).  Now 16-byte LDS writes, and eight rows' reads in flight per thread before the first of their stores.
__device__ __forceinline__ void sweep_zero_rows(uint8_t* ys, uint32_t words, uint32_t tid) {
    uint4* q = reinterpret_cast<uint4*>(ys);
    const uint32_t quads = words / 4u;
    for (uint32_t i = tid; i < quads; i += kSweepThreads) q[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < (words & 3u)) reinterpret_cast<uint32_t*>(ys)[quads * 4u + tid] = 0u;
}
template <bool kFloat>
__device__ __forceinline__ void sweep_store_rows(const uint8_t* ys, uint32_t nrows, uint32_t* __restrict__ out, uint32_t tid) {
    constexpr uint32_t kRowsInFlight = 8;
    const uint32_t* acc = reinterpret_cast<const uint32_t*>(ys);
    for (uint32_t base = tid; base < nrows; base += kRowsInFlight * kSweepThreads) {
        if constexpr (kFloat) {
            double v[kRowsInFlight];
#pragma unroll
            for (uint32_t k = 0; k < kRowsInFlight; ++k) v[k] = reinterpret_cast<const double*>(ys)[min(base + k * kSweepThreads, nrows)];      // (row nrows: the spare accumulator)
#pragma unroll
            for (uint32_t k = 0; k < kRowsInFlight; ++k)
                if (base + k * kSweepThreads < nrows) out[base + k * kSweepThreads] = __float_as_uint(static_cast<float>(v[k]));
        } else {
            uint32_t sum[kRowsInFlight], flags[kRowsInFlight];
            const uint32_t flag0 = SweepRows<false>::flag_word0(nrows);
#pragma unroll
            for (uint32_t k = 0; k < kRowsInFlight; ++k) {
                const uint32_t row = min(base + k * kSweepThreads, nrows);
                sum[k] = acc[row];
                flags[k] = acc[flag0 + (row >> 5)];
            }
            asm volatile("" ::: "memory");      // all sixteen reads are issued before the first result is looked at
#pragma unroll
            for (uint32_t k = 0; k < kRowsInFlight; ++k) {
                const uint32_t row = base + k * kSweepThreads;
                if (row < nrows) out[row] = ((flags[k] >> (row & 31u)) & 1u) ? 0xffffffffu : sum[k];      // AP_SAT (pe.h:72), as SweepRows<false>::finish
            }
        }
    }
}

template <bool kFloat, int kAblate, bool kNt>
__global__ __launch_bounds__(kSweepThreads) void spmv_sweep_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks,
                                                                  const uint32_t* __restrict__ x, uint32_t* __restrict__ out,
                                                                  int32_t row_part_filter, const uint32_t* __restrict__ part_heads, CarriedCombine carry) {
    using R = SweepRows<kFloat>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* ys = lds;                                            // nrows + 1 sums (+ the carry bitmap in fixed point)
    const uint32_t tid = threadIdx.x, lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // logical workgroups [k n/8, (k+1) n/8) on XCD k
    if (carry.partial) carried_combine<kFloat, kSweepThreads>(carry, blockIdx.x, gridDim.x, tid);   // y of the PREVIOUS step (spmv_device.h)
    uint32_t bi = wg;
    if (row_part_filter >= 0) {
        bi = ((const __attribute__((address_space(4))) uint32_t*)part_heads)[static_cast<uint32_t>(row_part_filter) * gridDim.x + wg];
        if (bi == kNoBlock) return;
    }
    const uint8_t* xs = sweep_scalar_pointer(x);
    const uint32_t lane_off = lane * 8u;
    bool first_block = true;
    for (uint32_t next = 0;; bi = next) {
        const BlockTable blk = (BlockTable)(blocks + bi);
        next = (row_part_filter >= 0 && blk->next_part > static_cast<uint32_t>(row_part_filter)) ? 0u : blk->next;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset, steps = blk->total_steps[0], pad_col = blk->first_col0;
        const uint8_t* stream = sweep_scalar_pointer(image + blk->wave_offset[0] + uint64_t(wave) * kChunkBytes);
        const __attribute__((address_space(4))) uint32_t* bases =
            (const __attribute__((address_space(4))) uint32_t*)(image + blk->wave_offset[1]) + uint64_t(wave) * steps;
        SweepLane st;
#pragma unroll
        for (int k = 0; k < kSweepDepth; ++k) { st.value[k] = 0; st.row[k] = nrows; }
        if (steps) sweep_prime<(kAblate & 8) != 0, kNt>(std::make_integer_sequence<int, kSweepDepth>(), stream, xs, pad_col, steps, lane_off);
        if (!first_block) __syncthreads();                        // the previous block's store has read the accumulators
        first_block = false;
        if (!(kAblate & 4)) sweep_zero_rows(ys, R::lds_words(nrows), tid);
        __syncthreads();
        if (steps) {
            const uint32_t last = steps - 1;
            uint32_t b[kSweepDepth];
#pragma unroll
            for (int k = 0; k < kSweepDepth; ++k) b[k] = bases[min(uint32_t(k), last)];
            // steps 0 .. steps + 7: step s adds what step s - 8 took.  Steps s >= `steps` take the last chunk again (the clamped prefetch)
            // and nobody adds that: the step that would runs at s + 8 >= steps + 8, which sweep_round skips
            for (uint32_t s0 = 0; s0 < steps + kSweepDepth; s0 += kSweepDepth) {
                uint32_t nb[kSweepDepth];
#pragma unroll
                for (int k = 0; k < kSweepDepth; ++k) nb[k] = bases[min(s0 + kSweepDepth + k, last)];
                sweep_round<kFloat, kAblate, kNt>(std::make_integer_sequence<int, kSweepDepth>(), st, stream, xs, s0, steps, steps + kSweepDepth, lane_off, b, ys, nrows);
#pragma unroll
                for (int k = 0; k < kSweepDepth; ++k) b[k] = nb[k];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory", HS_SWEEP_RING);
        }
        // no-return LDS atomics can outlive lgkmcnt(0) (spmv_rowblock_kernel): a returning one on the spare accumulator, awaited, cannot
        R::settle(ys, nrows, st);                                    // the last add's carry
        st.carry_old = st.carry_p = 0;
        const uint32_t flushed = atomicOr(reinterpret_cast<uint32_t*>(ys) + R::lds_words(nrows) - 1, 0u);
        asm volatile("" ::"v"(flushed));
        __syncthreads();
        if (!(kAblate & 4)) sweep_store_rows<kFloat>(ys, nrows, out + out0, tid);
        if (!next) break;
    }
}

}  // namespace

uint32_t spmv_sweep_lds_bytes(uint32_t max_block_rows, bool is_float) {
    const uint32_t words = is_float ? (max_block_rows + 1) * 2 : max_block_rows + 1 + (max_block_rows + 32) / 32;      // == SweepRows<>::lds_words
    return (words * 4u + 15u) & ~15u;
}

#ifdef HISPARSE_PROFILING
#define HS_FOR_EACH_SWEEP_VARIANT(X) X(0) X(1) X(2) X(3) X(4) X(7) X(9) X(13) X(18) X(23)
#else
#define HS_FOR_EACH_SWEEP_VARIANT(X) X(0)
#endif

hipError_t configure_sweep_kernels(uint32_t lds_bytes) {
    hipError_t e;
#define Y(F, A, N)                                                                                                                                   \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_sweep_kernel<F, A, N>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes))) != hipSuccess) return e;
#define X(A) Y(false, A, true) Y(true, A, true) Y(false, A, false) Y(true, A, false)
    HS_FOR_EACH_SWEEP_VARIANT(X)
#undef X
#undef Y
    return hipSuccess;
}

hipError_t launch_spmv_sweep(bool is_float, const SpmvLaunch& a, hipStream_t stream) {
    int ablate = 0, depth = 8;
    if (!profiling_switches(ablate, depth)) return hipErrorInvalidValue;
    const dim3 grid(a.num_workgroups), block(kSweepThreads);
    const CarriedCombine carry = carried(a);
#define Y(F, A, N) hipLaunchKernelGGL((spmv_sweep_kernel<F, A, N>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.x, a.out, a.row_part_filter, a.part_heads, carry)
#define X(A)                                                                                                                                         \
    if (ablate == A) {                                                                                                                               \
        if (a.stream_resident) { if (is_float) Y(true, A, false); else Y(false, A, false); }                                                         \
        else { if (is_float) Y(true, A, true); else Y(false, A, true); }                                                                             \
        return hipGetLastError();                                                                                                                    \
    }
    HS_FOR_EACH_SWEEP_VARIANT(X)
#undef X
#undef Y
    return hipErrorInvalidValue;      // no such profiling build of this kernel
}

}  // namespace dev
}  // namespace hisparse
