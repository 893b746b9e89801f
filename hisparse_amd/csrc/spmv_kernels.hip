// spmv_kernels.hip — the gfx950 (MI355X, CDNA4) kernel of the SpMV hot path.
//
// Replaces the FPGA dataflow  spmv_vector_loader -> spmv_sk0/1/2 -> spmv_result_drain
// (spmv/spmv_vector_loader.cpp:95-121, spmv/spmv_sk0.cpp:12-121, spmv/spmv_result_drain.cpp:10-126):
//
//   FPGA unit (reference)                                     here, per 1024-thread workgroup
//   --------------------------------------------------------  --------------------------------------------------
//   spmv_cluster: owns rows, PE output banks zeroed per        owns a row block; 64-bit (fixed) / fp32 (float) row
//     launch, dumped at the end (pe.h:121-178)                   accumulators in LDS, zeroed per block, written once
//   vector loader + vecbuf_access_unit: x partition in 8       4 LOADER wavefronts copy the next x sub-tile (64 KiB)
//     banks, writer/reader alternate per partition               into the idle one of two LDS buffers while ...
//     (vecbuf_access_unit.h:146-163)
//   CPSR_matrix_loader: one 64-byte packet per cycle           ... 12 CONSUMER wavefronts stream coalesced 8-byte
//     (spmv_cluster.h:73-98)                                      elements, 8 loads in flight per lane
//   shuffle 1 by col%8 + bank read (shuffle.h, vecbuf :126)    ds_read_b32 gather from the LDS sub-tile
//   shuffle 2 by row%8 + PE accumulate (pe.h:62-81)            ds_add_u64 / ds_add_f32 into the row accumulators
//   result packer + drain, natural row order                   coalesced y store, AP_SAT clamp applied once
//     (spmv_result_drain.cpp:104-113)
//
// One barrier per (row block, x sub-tile) hands the freshly filled buffer to the consumers.
// No global atomics, no second pass: y is complete when the kernel ends.  Bandwidth-bound
// integer / fp32 gather work — no MFMA anywhere.
// Numerics: fixed point = ap_ufixed<32,8,AP_RND,AP_SAT> products summed exactly in 64 bits and clamped
// once (bit-exact with the saturating PE because every term is non-negative);
// float = separate fp32 multiply and add (no FMA contraction: -ffp-contract=off); the order of the
// adds differs from the FPGA's arrival order, hence tolerance parity.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr int kThreads = kWaveLanes * kWavesPerWorkgroup;       // 1024
constexpr int kConsumerThreads = kWaveLanes * kConsumerWaves;   // 768
constexpr int kLoaderThreads = kThreads - kConsumerThreads;     // 256
constexpr uint32_t kBufBytes = kSubTileCols * 4u;               // one x buffer of the LDS ring (32 KiB)

// mat_val * vec_val narrowed to Q8.24: exact 64-bit product, + half LSB, >> 24, saturate (pe.h:64).
__device__ __forceinline__ uint32_t q8_24_mul(uint32_t a, uint32_t b) {
    const uint64_t wide = static_cast<uint64_t>(a) * b;
    const uint64_t r = (wide >> 24) + ((wide >> 23) & 1u);
    return r > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(r);
}

template <bool kFloat>
struct Rows;
template <>
struct Rows<false> {
    using acc_t = unsigned long long;   // LDS accumulator
    using prod_t = unsigned long long;
    static __device__ __forceinline__ prod_t product(uint32_t mat, uint32_t vec) { return q8_24_mul(mat, vec); }
    static __device__ __forceinline__ void add(acc_t* ys, uint32_t row, prod_t p) { atomicAdd(ys + row, p); }   // ds_add_u64
    static __device__ __forceinline__ uint32_t finish(acc_t s) { return s > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(s); }  // AP_SAT (pe.h:72)
};
template <>
struct Rows<true> {
    using acc_t = float;
    using prod_t = float;
    // multiply, then add: two roundings like the float PEs (pe-pob.h:63-65, pe-stall.h:52,138)
    static __device__ __forceinline__ prod_t product(uint32_t mat, uint32_t vec) { return __uint_as_float(mat) * __uint_as_float(vec); }
    static __device__ __forceinline__ void add(acc_t* ys, uint32_t row, prod_t p) { atomicAdd(ys + row, p); }   // ds_add_f32
    static __device__ __forceinline__ uint32_t finish(acc_t s) { return __float_as_uint(s); }
};

// Sum over the 64 lanes of a wavefront (result valid in every lane).
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWaveLanes);
    return v;
}

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
// "at most kFills refills are still in flight" (vmcnt retires in order; the loader wavefronts issue nothing else)
template <int kFills>
__device__ __forceinline__ void dma_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kFills * kDmaPerFill) : "memory");
}

// The element stream is loaded with hand-placed instructions: hipcc's s_waitcnt placement degrades to vmcnt(0)
// (a full drain per element) in the unit/barrier control flow below, which would serialise the HBM stream.
// Rules kept here: exactly ONE stream_load per step and no other vector-memory instruction inside the consumer
// loop, so "the load issued kDepth steps ago has landed" is exactly vmcnt(kDepth - 1) (vmcnt retires in order).
// `nt`: every element is read exactly once per SpMV, so it should not displace x in the L2 (measured: -8 % kernel time).
__device__ __forceinline__ void stream_load(uint64_t& dst, const void* addr) {
    asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(dst) : "v"(addr) : "memory");
}
template <int kOutstanding>
__device__ __forceinline__ void stream_wait(uint64_t& v) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(kOutstanding) : "memory");
}

// Workgroup barrier that orders LDS traffic only: the consumers' prefetched global loads stay in flight
// across it (a plain __syncthreads() carries a generic release fence and would drain them).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The consumer side of one row block: stream this wavefront's chunks through all sub-tiles of the block.
// kDense: the block has few, long rows and its chunks are row-sorted (Block::flags & kBlockDenseRows).
template <bool kFloat, int kAblate, int kDepth, bool kDense>
__device__ __forceinline__ void consume_block(const uint8_t* stream, const Unit* __restrict__ unit, uint32_t U, uint32_t wave, uint32_t lane,
                                              const uint32_t* xs, uint32_t ring, typename Rows<kFloat>::acc_t* ys) {
    const uint32_t total = unit[U - 1].end_step[wave];
    const uint32_t last = total ? total - 1 : 0;   // prefetches past the end re-read the last chunk (no branch)
    uint64_t buf[kDepth];
    // The loader branch of the kernel leaves "LDS-DMA may be pending" in hipcc's wait-count bookkeeping, and that state
    // reaches this loop around the block loop: every LDS store on a conditional path below (the dense-row hand-over)
    // would then get its own s_waitcnt vmcnt(0) and drain the prefetch ring.  A consumer wavefront never has LDS-DMA in
    // flight, so say so once, up front, where nothing is in flight yet.
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), lgkmcnt/expcnt untouched
#pragma unroll
    for (int k = 0; k < kDepth; ++k) stream_load(buf[k], stream + static_cast<size_t>(min(static_cast<uint32_t>(k), last)) * kWaveStrideBytes);
    constexpr uint32_t kNoRow = 0xffffffffu;
    uint32_t run_row = kNoRow;                      // kDense: row whose products are being summed in registers
    typename Rows<kFloat>::prod_t run_sum = 0;      // kDense: this lane's share of that sum
    uint32_t u = 0, slot = 0, end = unit[0].end_step[wave];   // slot = u % ring
    if (kAblate & 64) { u = U - 1; end = total; }                // profiling: one unit per block
    const uint32_t* xb = xs;
    for (uint32_t base = 0;; base += kDepth) {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            const uint32_t s = base + k;
            while (s == end) {                 // this wavefront finished sub-tile u (possibly with no work in it)
                if (kDense && u + 1 == U && run_row != kNoRow) {   // last sub-tile: hand the register sum over before the final barrier
                    const typename Rows<kFloat>::prod_t sum = wave_sum(run_sum);
                    if (lane == 0) Rows<kFloat>::add(ys, run_row, sum);
                    run_row = kNoRow;
                }
                if (!(kAblate & 8)) lds_barrier();
                if (++u == U) goto block_done;
                end = unit[u].end_step[wave];
                slot = slot + 1 == ring ? 0 : slot + 1;
                xb = xs + slot * kSubTileCols;
            }
            stream_wait<kDepth - 1>(buf[k]);
            const uint32_t mat = static_cast<uint32_t>(buf[k]), cr = static_cast<uint32_t>(buf[k] >> 32);
            const uint32_t xv = (kAblate & 2) ? cr : xb[cr & 0xffffu];
            if (kAblate & 1) {
                asm volatile("" ::"v"(xv), "v"(mat));
            } else {
                const typename Rows<kFloat>::prod_t prod = Rows<kFloat>::product(mat, xv);
                const uint32_t row = cr >> 16;
                if (kDense) {
                    // Chunks are row-sorted, so the whole wavefront is usually on ONE row, and stays on it for many chunks:
                    // keep a per-lane running sum in registers while the row does not change and touch the LDS accumulator
                    // only when it does (one wave_sum + one ds_add per row per wavefront instead of a 64-way conflicting
                    // atomic per chunk).
                    const uint32_t row0 = __builtin_amdgcn_readfirstlane(row);
                    const bool uniform = __ballot(row == row0) == ~0ull;
                    if (uniform && row0 == run_row) {
                        run_sum += prod;
                    } else {
                        if (run_row != kNoRow) {
                            const typename Rows<kFloat>::prod_t sum = wave_sum(run_sum);
                            if (lane == 0) Rows<kFloat>::add(ys, run_row, sum);
                        }
                        if (uniform) { run_row = row0; run_sum = prod; }
                        else { run_row = kNoRow; run_sum = 0; Rows<kFloat>::add(ys, row, prod); }   // chunk straddles rows
                    }
                } else {
                    Rows<kFloat>::add(ys, row, prod);
                }
            }
            stream_load(buf[k], stream + static_cast<size_t>(min(s + kDepth, last)) * kWaveStrideBytes);
        }
    }
block_done:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped tail prefetches must land before their registers are reused
}

// GATHER mode (x too large for LDS staging to pay): the same streams, but x is read per element straight from
// L2 / Infinity Cache.  Two software pipelines share vmcnt, both hand-counted: element loads run kDepth = 8 steps ahead,
// the x gather of an element is issued kGather = 4 steps before it is consumed.  Issue order per step s:
//   wait G(s) -> multiply-accumulate -> L(s+8) -> wait L(s+4) -> G(s+4)
// so G(s) has exactly 2*(4-1) = 6 younger operations when it is awaited and L(s+4) has 8 (the prologue is peeled with
// its own exact counts).  No x ring, no loaders, no barriers: a unit boundary only changes the column base.
__device__ __forceinline__ void gather_load(uint32_t& dst, uint32_t byte_offset, const uint32_t* base) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(byte_offset), "s"(base) : "memory");
}
template <int kOutstanding>
__device__ __forceinline__ void gather_wait(uint32_t& v) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(kOutstanding) : "memory");
}

template <bool kFloat>
__device__ __forceinline__ void consume_block_gather(const uint8_t* stream, const Unit* __restrict__ unit, uint32_t U, uint32_t wave,
                                                     uint32_t lane, const uint32_t* __restrict__ x, typename Rows<kFloat>::acc_t* ys) {
    constexpr int kDepth = 8, kGather = 4;
    const uint32_t total = unit[U - 1].end_step[wave];
    if (total == 0) return;
    const uint32_t last = total - 1;
    uint64_t buf[kDepth];
    uint32_t xg[kGather];
    // column base of the element whose gather is issued next (position p = s + kGather)
    uint32_t ug = 0, endg = unit[0].end_step[wave], col0g = unit[0].col0;
    auto issue_gather = [&](uint32_t p, uint64_t& element, uint32_t& dst) {
        while (p == endg && ug + 1 < U) { ++ug; endg = unit[ug].end_step[wave]; col0g = unit[ug].col0; }
        gather_load(dst, (col0g + (static_cast<uint32_t>(element >> 32) & 0xffffu)) * 4u, x);
    };
#pragma unroll
    for (int k = 0; k < kGather; ++k) stream_load(buf[k], stream + static_cast<size_t>(min(static_cast<uint32_t>(k), last)) * kWaveStrideBytes);
    // peeled prologue: virtual steps -4 .. -1 issue L(4..7) and G(0..3) with the exact number of younger operations
    stream_load(buf[4], stream + static_cast<size_t>(min(4u, last)) * kWaveStrideBytes); stream_wait<4>(buf[0]); issue_gather(0, buf[0], xg[0]);
    stream_load(buf[5], stream + static_cast<size_t>(min(5u, last)) * kWaveStrideBytes); stream_wait<5>(buf[1]); issue_gather(1, buf[1], xg[1]);
    stream_load(buf[6], stream + static_cast<size_t>(min(6u, last)) * kWaveStrideBytes); stream_wait<6>(buf[2]); issue_gather(2, buf[2], xg[2]);
    stream_load(buf[7], stream + static_cast<size_t>(min(7u, last)) * kWaveStrideBytes); stream_wait<7>(buf[3]); issue_gather(3, buf[3], xg[3]);
    for (uint32_t base = 0; base < total; base += kDepth) {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            const uint32_t s = base + k;
            if (s >= total) break;                               // wave-uniform
            gather_wait<2 * (kGather - 1)>(xg[k % kGather]);
            const uint32_t mat = static_cast<uint32_t>(buf[k]), cr = static_cast<uint32_t>(buf[k] >> 32);
            Rows<kFloat>::add(ys, cr >> 16, Rows<kFloat>::product(mat, xg[k % kGather]));
            stream_load(buf[k], stream + static_cast<size_t>(min(s + kDepth, last)) * kWaveStrideBytes);
            stream_wait<2 * kGather>(buf[(k + kGather) % kDepth]);
            issue_gather(min(s + kGather, last), buf[(k + kGather) % kDepth], xg[k % kGather]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail prefetches and gathers must land before their registers are reused
}

template <bool kFloat>
__global__ __launch_bounds__(kThreads) void spmv_gather_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks,
                                                                const Unit* __restrict__ units, const uint32_t* __restrict__ wg_first,
                                                                const uint32_t* __restrict__ block_order, const uint32_t* __restrict__ x,
                                                                uint32_t* __restrict__ out, int32_t row_part_filter) {
    using acc_t = typename Rows<kFloat>::acc_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    acc_t* ys = reinterpret_cast<acc_t*>(lds);                   // [nrows + 1]
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint32_t q_end = wg_first[wg + 1];
    for (uint32_t q = wg_first[wg]; q < q_end; ++q) {
        const Block* blk = blocks + block_order[q];
        if (row_part_filter >= 0 && blk->row_part != static_cast<uint32_t>(row_part_filter)) continue;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset;
        const uint32_t U = blk->unit_end - blk->unit_begin;
        for (uint32_t i = tid; i <= nrows; i += kThreads) ys[i] = 0;
        __syncthreads();
        if (U > 0 && wave < kConsumerWaves)
            consume_block_gather<kFloat>(image + blk->wave_offset[wave] + lane * 8u, units + blk->unit_begin, U, wave, lane, x, ys);
        __syncthreads();
        for (uint32_t i = tid; i < nrows; i += kThreads) out[out0 + i] = Rows<kFloat>::finish(ys[i]);
        __syncthreads();
    }
}

// kDepth: element loads in flight per lane (kDepth x 512 B per wavefront).
// kAblate (profiling builds only, HISPARSE_ABLATE): bit 0 = no LDS accumulate, bit 1 = no LDS gather,
// bit 2 = no x sub-tile refill, bit 3 = no per-sub-tile barrier, bit 4 = no block prologue (zero + first sub-tile),
// bit 5 = no result store, bit 6 = ignore unit boundaries.  Any non-zero value gives wrong results.
template <bool kFloat, int kAblate, int kDepth>
__global__ __launch_bounds__(kThreads) void spmv_rowblock_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks,
                                                                  const Unit* __restrict__ units, const uint32_t* __restrict__ wg_first,
                                                                  const uint32_t* __restrict__ block_order, const uint32_t* __restrict__ x,
                                                                  uint32_t* __restrict__ out, int32_t row_part_filter, uint32_t ring) {
    using acc_t = typename Rows<kFloat>::acc_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint32_t* xs = reinterpret_cast<uint32_t*>(lds);             // [ring][kSubTileCols]
    acc_t* ys = reinterpret_cast<acc_t*>(lds + ring * kBufBytes); // [nrows + 1]
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    const bool loader = wave >= kConsumerWaves;
    // Workgroup b runs on XCD b % 8 (observed dispatch order): neighbouring logical workgroups (neighbouring row
    // blocks, same x sub-tiles at about the same time) then share an L2.  Speed only, never correctness.
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    const uint32_t q_end = wg_first[wg + 1];
    bool first_block = true;
    for (uint32_t q = wg_first[wg]; q < q_end; ++q) {
        const Block* blk = blocks + block_order[q];
        if (row_part_filter >= 0 && blk->row_part != static_cast<uint32_t>(row_part_filter)) continue;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset;
        const Unit* unit = units + blk->unit_begin;
        const uint32_t U = blk->unit_end - blk->unit_begin;

        if (!first_block) __syncthreads();   // the previous block's result store has read the accumulators
        first_block = false;
        if (!(kAblate & 16)) for (uint32_t i = tid; i <= nrows; i += kThreads) ys[i] = 0;           // PE banks start at zero (pe.h:131-135)
        if (U > 0 && !(kAblate & 16)) fill_x<kSubTileCols / 4 / kThreads, kThreads>(xs, x, unit[0].col0, unit[0].ncols, tid);
        __syncthreads();

        if (U > 0) {
            if (loader) {
                // ---- loader wavefronts: keep the ring of x buffers up to three sub-tiles ahead of the consumers ----
                const uint32_t lw = wave - kConsumerWaves;
                uint32_t fill_slot = 1 == ring ? 0 : 1;                // ring slot of the next refill (sub-tile v lives in slot v % ring)
                auto refill = [&](uint32_t v) {
                    if (!(kAblate & 4)) dma_fill_x(xs + fill_slot * kSubTileCols, x, unit[v].col0, unit[v].ncols, lw, lane);
                    fill_slot = fill_slot + 1 == ring ? 0 : fill_slot + 1;
                };
                uint32_t issued = 1;                                   // sub-tile 0 was copied synchronously above
                while (issued < U && issued < ring - 1) refill(issued++);
                for (uint32_t u = 0; u < U; ++u) {
                    // slot (u + ring - 1) % ring last held sub-tile u-1, which every consumer left at the previous barrier
                    if (issued < U) refill(issued++);
                    // sub-tile u+1 must be resident before the consumers enter it (they do so after this barrier)
                    const uint32_t younger = issued - min(issued, u + 2);   // refills issued after the one for u+1: 0..ring-2
                    if (younger >= 2) dma_wait<2>(); else if (younger == 1) dma_wait<1>(); else dma_wait<0>();
                    if (!(kAblate & 8)) __builtin_amdgcn_s_barrier();
                }
            } else {
                // ---- consumer wavefronts: stream elements, gather x, accumulate rows -----------------------------
                const uint8_t* stream = image + blk->wave_offset[wave] + lane * 8u;
                if (blk->flags & kBlockDenseRows) consume_block<kFloat, kAblate, kDepth, true>(stream, unit, U, wave, lane, xs, ring, ys);
                else consume_block<kFloat, kAblate, kDepth, false>(stream, unit, U, wave, lane, xs, ring, ys);
            }
        }
        // every sub-tile barrier has passed: the accumulators are final
        // (no barrier after the store: the last block's stores drain while the workgroup retires)
        if (!(kAblate & 32)) for (uint32_t i = tid; i < nrows; i += kThreads) out[out0 + i] = Rows<kFloat>::finish(ys[i]);
    }
}

// Column-sliced matrices: y[r] = sum over slices of the per-slice partial results.  Fixed point: each partial is
// already clamped to 2^32-1 and min(sum, MAX) == min(sum of min(part, MAX), MAX) for non-negative parts, so the
// result is still exactly the saturating sum of the PE (pe.h:72).
template <bool kFloat>
__global__ __launch_bounds__(256) void combine_slices_kernel(const uint32_t* __restrict__ partial, uint32_t* __restrict__ y,
                                                             uint32_t num_rows, uint32_t slices, uint32_t row_lo, uint32_t row_hi) {
    const uint32_t r = row_lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= row_hi) return;
    if (kFloat) {
        float s = 0.0f;
        for (uint32_t k = 0; k < slices; ++k) s += __uint_as_float(partial[static_cast<size_t>(k) * num_rows + r]);
        y[r] = __float_as_uint(s);
    } else {
        uint64_t s = 0;
        for (uint32_t k = 0; k < slices; ++k) s += partial[static_cast<size_t>(k) * num_rows + r];
        y[r] = s > 0xffffffffull ? 0xffffffffu : static_cast<uint32_t>(s);
    }
}

template <bool kFloat, int kAblate, int kDepth>
hipError_t configure_one(uint32_t lds_bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_rowblock_kernel<kFloat, kAblate, kDepth>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
}

int env_int(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}

}  // namespace

uint32_t spmv_lds_bytes(uint32_t max_block_rows, uint32_t ring_buffers) {
    return ring_buffers * kBufBytes + (max_block_rows + 1) * 8u;
}

#define HS_FOR_EACH_VARIANT(X) \
    X(true, 0, 8) X(false, 0, 8) X(false, 0, 16) X(false, 3, 8) X(false, 4, 8) X(false, 8, 8) X(false, 15, 8) X(false, 79, 8) X(false, 127, 8) X(false, 31, 8) X(false, 47, 8)

hipError_t configure_spmv_kernels(uint32_t lds_bytes) {
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_gather_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 static_cast<int>(lds_bytes))) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_gather_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 static_cast<int>(lds_bytes))) != hipSuccess) return e;
#define X(F, A, D) if ((e = configure_one<F, A, D>(lds_bytes)) != hipSuccess) return e;
    HS_FOR_EACH_VARIANT(X)
#undef X
    return hipSuccess;
}

hipError_t launch_spmv(bool is_float, const SpmvLaunch& a, hipStream_t stream) {
    if (a.num_workgroups == 0) return hipSuccess;
    const dim3 grid(a.num_workgroups), block(kThreads);
    if (a.ring_buffers == 0) {   // gather mode
        if (is_float) hipLaunchKernelGGL(spmv_gather_kernel<true>, grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.wg_first,
                                         a.block_order, a.x, a.out, a.row_part_filter);
        else hipLaunchKernelGGL(spmv_gather_kernel<false>, grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.wg_first,
                                a.block_order, a.x, a.out, a.row_part_filter);
        return hipGetLastError();
    }
    // profiling aids: HISPARSE_ABLATE removes parts of the work (wrong results), HISPARSE_DEPTH picks the prefetch depth
    static const int ablate = env_int("HISPARSE_ABLATE", 0), depth = env_int("HISPARSE_DEPTH", 8);
    bool launched = false;
#define X(F, A, D)                                                                                                           \
    if (!launched && is_float == F && (F || (ablate == A && depth == D))) {                                                  \
        hipLaunchKernelGGL((spmv_rowblock_kernel<F, A, D>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units,    \
                           a.wg_first, a.block_order, a.x, a.out, a.row_part_filter, a.ring_buffers);                         \
        launched = true;                                                                                                     \
    }
    HS_FOR_EACH_VARIANT(X)
#undef X
    if (!launched) return hipErrorInvalidValue;   // unknown HISPARSE_ABLATE / HISPARSE_DEPTH combination
    return hipGetLastError();
}

hipError_t launch_combine_slices(bool is_float, const uint32_t* partial, uint32_t* y, uint32_t num_rows, uint32_t slices, uint32_t row_lo,
                                 uint32_t row_hi, hipStream_t stream) {
    if (row_hi <= row_lo) return hipSuccess;
    const dim3 grid((row_hi - row_lo + 255) / 256), block(256);
    if (is_float) hipLaunchKernelGGL(combine_slices_kernel<true>, grid, block, 0, stream, partial, y, num_rows, slices, row_lo, row_hi);
    else hipLaunchKernelGGL(combine_slices_kernel<false>, grid, block, 0, stream, partial, y, num_rows, slices, row_lo, row_hi);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
