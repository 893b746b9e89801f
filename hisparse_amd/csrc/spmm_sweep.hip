// spmm_sweep.hip — Y = A X for FOUR dense vectors at once over a SWEEP image (EXTENSION; the reference stubs the
// types -- spmv/libfpga/common.h:52-54 -- and has no SpMM).  Round 5: until now hs_spmm over an element-stream (graph) image was k SpMVs.
//
// The SWEEP image (stream_tiles.h, spmv_sweep.hip) is the element-stream format whose kernel has no x staging and no units: a block's
// elements in column order, { value word, row << 16 | column - chunk base }, x gathered from L2 per element.  That makes the k-wide
// variant a local change: X is interleaved [column][4] (16 bytes per column: ONE global_load_dwordx4 per element where the SpMV kernel
// issues a dword), a product per vector, four sets of row accumulators in LDS.  The matrix is streamed once per four columns of X.
// Four sets of accumulators need a quarter of the rows per block, so the image must have been PLANNED for it: hs_set_option
// "spmm_vectors" = 4 before the load (sweep_tiles.cpp: SWEEP forced, rows per block / 4); hs_run works on such an image as on any other.
//
// Per wavefront and step, in flight: eight chunks (a0..a15) and eight gathers (a16..a47), one counted wait (the ring of spmv_sweep.hip with
// wider gathers).  Arithmetic per column exactly as hs_run: fixed point = exact saturating sum of individually rounded products (32-bit
// wrapping sums + a carry bit per row and vector), float = fp32 products summed in double, rounded once per row and column slice.
#include <hip/hip_runtime.h>

#include <utility>

#include "spmv_device.h"
#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr int kVecs = 4;
constexpr int kThreads = kSweepWaves * kWaveLanes;
#ifndef HS_SPMM_DEPTH
#define HS_SPMM_DEPTH 3      // round 5 (profiles/r05_sweep_ring_depth.txt), k = 16, us per SpMM, ogbl-ppa / pokec: depth 8: 600 / 991, 6: 573 / 965, 4: 566 / 947,
#endif                       // 3: 544 / 933, 2: 588 / 902 -- as in spmv_sweep.hip, more in flight costs the gathers their L1 lines
static_assert(HS_SPMM_DEPTH >= 2 && HS_SPMM_DEPTH <= 8, "the ring's register numbers hold up to eight slots");
constexpr int kDepth = HS_SPMM_DEPTH;      // chunks (and 16-byte gathers) in flight per wavefront; the ring's register numbers below allow up to 8

#define HS_SPMM_RING "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", \
                     "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42",   \
                     "a43", "a44", "a45", "a46", "a47"

__device__ __forceinline__ const uint8_t* uniform_pointer(const void* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));
    return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// chunk slot K in a[2K : 2K+1]; gather slot K (four x words) in a[16 + 4K : 16 + 4K + 3]
template <int K>
__device__ __forceinline__ void issue_chunk(const uint8_t* base, uint32_t off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 a[%0:%1], %2, %3 nt" ::"n"(2 * K), "n"(2 * K + 1), "v"(off), "s"(base) : "memory", HS_SPMM_RING);
}
template <int K>
__device__ __forceinline__ void issue_gather(const uint8_t* x4, uint32_t byte_off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 a[%0:%1], %2, %3" ::"n"(16 + 4 * K), "n"(16 + 4 * K + 3), "v"(byte_off), "s"(x4) : "memory", HS_SPMM_RING);
}
template <int K>
__device__ __forceinline__ void take(uint32_t& value, uint32_t& where, uint32_t (&xv)[kVecs]) {
    asm volatile("s_waitcnt vmcnt(%12)\n\tv_accvgpr_read_b32 %0, a[%6]\n\tv_accvgpr_read_b32 %1, a[%7]\n\tv_accvgpr_read_b32 %2, a[%8]\n\t"
                 "v_accvgpr_read_b32 %3, a[%9]\n\tv_accvgpr_read_b32 %4, a[%10]\n\tv_accvgpr_read_b32 %5, a[%11]"
                 : "=v"(value), "=v"(where), "=v"(xv[0]), "=v"(xv[1]), "=v"(xv[2]), "=v"(xv[3])
                 : "n"(2 * K), "n"(2 * K + 1), "n"(16 + 4 * K), "n"(16 + 4 * K + 1), "n"(16 + 4 * K + 2), "n"(16 + 4 * K + 3), "n"(2 * (kDepth - 1))
                 : "memory");
}

// Row accumulators of ONE vector (the layouts of spmv_sweep.hip: SweepRows); the kernel keeps kVecs such sets back to back.
template <bool kFloat>
struct Sums;
template <>
struct Sums<true> {
    struct Carry {};
    static __device__ __forceinline__ uint32_t words(uint32_t nrows) { return (nrows + 1) * 2; }
    static __device__ __forceinline__ void add(uint32_t* set, uint32_t, Carry&, uint32_t row, uint32_t value, uint32_t xv) {
        atomicAdd(reinterpret_cast<double*>(set) + row, static_cast<double>(__uint_as_float(value) * __uint_as_float(xv)));
    }
    static __device__ __forceinline__ void settle(uint32_t*, uint32_t, Carry&) {}
    static __device__ __forceinline__ uint32_t finish(const uint32_t* set, uint32_t, uint32_t row) {
        return __float_as_uint(static_cast<float>(reinterpret_cast<const double*>(set)[row]));
    }
};
template <>
struct Sums<false> {
    struct Carry { uint32_t old = 0, p = 0, row = 0; };      // the last add, whose carry is looked at one step later (spmv_sweep.hip)
    static __device__ __forceinline__ uint32_t words(uint32_t nrows) { return nrows + 1 + (nrows + 32) / 32; }
    static __device__ __forceinline__ void settle(uint32_t* set, uint32_t nrows, Carry& c) {
        if (c.old + c.p < c.old) atomicOr(set + nrows + 1 + (c.row >> 5), 1u << (c.row & 31u));      // AP_SAT (pe.h:72): once beyond 2^32 - 1, always
    }
    static __device__ __forceinline__ void add(uint32_t* set, uint32_t nrows, Carry& c, uint32_t row, uint32_t value, uint32_t xv) {
        settle(set, nrows, c);
        c.p = q8_24_mul(value, xv);
        c.row = row;
        c.old = atomicAdd(set + row, c.p);                  // ds_add_rtn_u32
    }
    static __device__ __forceinline__ uint32_t finish(const uint32_t* set, uint32_t nrows, uint32_t row) {
        return ((set[nrows + 1 + (row >> 5)] >> (row & 31u)) & 1u) ? 0xffffffffu : set[row];
    }
};

template <bool kFloat>
struct Lane {
    uint32_t value[kDepth], row[kDepth];                     // the elements whose x words are on their way
    typename Sums<kFloat>::Carry carry[kVecs];
};

// step s of a wavefront (ring slot K = s % 8): the order of spmv_sweep.hip's sweep_step
template <bool kFloat, int K>
__device__ __forceinline__ void step(Lane<kFloat>& st, const uint8_t* stream, const uint8_t* x4, uint32_t s, uint32_t steps, uint32_t lane_off, uint32_t base,
                                     uint32_t* sets, uint32_t set_words, uint32_t nrows) {
    uint32_t value, where, xv[kVecs];
    take<K>(value, where, xv);
#pragma unroll
    for (int j = 0; j < kVecs; ++j) Sums<kFloat>::add(sets + j * set_words, nrows, st.carry[j], st.row[K], st.value[K], xv[j]);
    st.value[K] = value;
    st.row[K] = where >> 16;
    issue_gather<K>(x4, (base + (where & 0xffffu)) * (4u * kVecs));
    issue_chunk<K>(stream, min(s + kDepth, steps - 1) * (kSweepWaves * kChunkBytes) + lane_off);
}
template <int... Ks>
__device__ __forceinline__ void prime(std::integer_sequence<int, Ks...>, const uint8_t* stream, const uint8_t* x4, uint32_t pad_col, uint32_t steps, uint32_t lane_off) {
    ((issue_gather<Ks>(x4, pad_col * (4u * kVecs)), issue_chunk<Ks>(stream, min(uint32_t(Ks), steps - 1) * (kSweepWaves * kChunkBytes) + lane_off)), ...);
}
template <bool kFloat, int... Ks>
__device__ __forceinline__ void round_of_steps(std::integer_sequence<int, Ks...>, Lane<kFloat>& st, const uint8_t* stream, const uint8_t* x4, uint32_t s0, uint32_t steps,
                                               uint32_t end, uint32_t lane_off, const uint32_t (&b)[kDepth], uint32_t* sets, uint32_t set_words, uint32_t nrows) {
    ((s0 + Ks < end ? step<kFloat, Ks>(st, stream, x4, s0 + Ks, steps, lane_off, b[Ks], sets, set_words, nrows) : (void)0), ...);
}

// out: [column slice][vector][row] words (one slice: [vector][row]) -- the combine pass then adds the slices of all four vectors in one launch
template <bool kFloat>
__global__ __launch_bounds__(kThreads) void spmm_sweep_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks, const uint32_t* __restrict__ x4,
                                                            uint32_t* __restrict__ out, uint32_t num_rows) {
    using S = Sums<kFloat>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t* sets = reinterpret_cast<uint32_t*>(lds);
    const uint32_t tid = threadIdx.x, lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint8_t* xs = uniform_pointer(x4);
    const uint32_t lane_off = lane * 8u;
    bool first_block = true;
    uint32_t bi = wg;
    for (uint32_t next = 0;; bi = next) {
        const BlockTable blk = (BlockTable)(blocks + bi);
        next = blk->next;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset, steps = blk->total_steps[0], pad_col = blk->first_col0;
        const uint32_t set_words = (S::words(nrows) + 3u) & ~3u;
        const uint8_t* stream = uniform_pointer(image + blk->wave_offset[0] + uint64_t(wave) * kChunkBytes);
        const __attribute__((address_space(4))) uint32_t* bases =
            (const __attribute__((address_space(4))) uint32_t*)(image + blk->wave_offset[1]) + uint64_t(wave) * steps;
        Lane<kFloat> st;
#pragma unroll
        for (int k = 0; k < kDepth; ++k) { st.value[k] = 0; st.row[k] = nrows; }
        if (steps) prime(std::make_integer_sequence<int, kDepth>(), stream, xs, pad_col, steps, lane_off);
        if (!first_block) __syncthreads();
        first_block = false;
        for (uint32_t i = tid, n = set_words * kVecs; i < n; i += kThreads) sets[i] = 0;
        __syncthreads();
        if (steps) {
            const uint32_t last = steps - 1;
            uint32_t b[kDepth];
#pragma unroll
            for (int k = 0; k < kDepth; ++k) b[k] = bases[min(uint32_t(k), last)];
            for (uint32_t s0 = 0; s0 < steps + kDepth; s0 += kDepth) {
                uint32_t nb[kDepth];
#pragma unroll
                for (int k = 0; k < kDepth; ++k) nb[k] = bases[min(s0 + kDepth + k, last)];
                round_of_steps<kFloat>(std::make_integer_sequence<int, kDepth>(), st, stream, xs, s0, steps, steps + kDepth, lane_off, b, sets, set_words, nrows);
#pragma unroll
                for (int k = 0; k < kDepth; ++k) b[k] = nb[k];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory", HS_SPMM_RING);
        }
#pragma unroll
        for (int j = 0; j < kVecs; ++j) S::settle(sets + j * set_words, nrows, st.carry[j]);
        // no-return LDS atomics can outlive lgkmcnt(0) (spmv_kernels.hip): a returning one per wavefront, awaited, cannot
        const uint32_t flushed = atomicOr(sets + set_words * kVecs - 1, 0u);
        asm volatile("" ::"v"(flushed));
        __syncthreads();
        const uint32_t slice = out0 / num_rows, row0 = out0 - slice * num_rows;
        uint32_t* dst = out + static_cast<size_t>(slice) * kVecs * num_rows + row0;
#pragma unroll
        for (int j = 0; j < kVecs; ++j)
            for (uint32_t i = tid; i < nrows; i += kThreads) dst[static_cast<size_t>(j) * num_rows + i] = S::finish(sets + j * set_words, nrows, i);
        if (!next) break;
    }
}

// x4[c][j] = column j of X at word c (0 for the columns beyond `vectors`)
__global__ __launch_bounds__(256) void interleave4_kernel(const uint32_t* __restrict__ x, uint64_t ldx, uint32_t vectors, uint32_t num_cols, uint4* __restrict__ x4) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= num_cols) return;
    uint32_t w[kVecs];
#pragma unroll
    for (uint32_t j = 0; j < uint32_t(kVecs); ++j) w[j] = j < vectors ? x[static_cast<size_t>(j) * ldx + c] : 0u;
    x4[c] = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace

uint32_t spmm_sweep_lds_bytes(uint32_t max_block_rows, bool is_float) {
    const uint32_t words = is_float ? (max_block_rows + 1) * 2 : max_block_rows + 1 + (max_block_rows + 32) / 32;
    return (((words + 3u) & ~3u) * 4u * kVecs + 15u) & ~15u;
}

hipError_t configure_spmm_sweep_kernels(uint32_t lds_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_sweep_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_sweep_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
}

hipError_t launch_spmm_sweep(bool is_float, const SpmmSweepLaunch& a, hipStream_t stream) {
    if (a.vectors == 0 || a.vectors > uint32_t(kVecs) || a.num_workgroups == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(interleave4_kernel, dim3((a.num_cols + 255) / 256), dim3(256), 0, stream, a.x, a.ldx, a.vectors, a.num_cols, reinterpret_cast<uint4*>(a.x4));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint32_t lds = spmm_sweep_lds_bytes(a.max_block_rows, is_float);
    if (is_float) hipLaunchKernelGGL(spmm_sweep_kernel<true>, dim3(a.num_workgroups), dim3(kThreads), lds, stream, a.image, a.blocks, a.x4, a.out, a.num_rows);
    else hipLaunchKernelGGL(spmm_sweep_kernel<false>, dim3(a.num_workgroups), dim3(kThreads), lds, stream, a.image, a.blocks, a.x4, a.out, a.num_rows);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
