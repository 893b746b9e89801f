// spmm_bitmap.hip — Y = A X for kVecs dense vectors at once over a BITMAP image (EXTENSION; the reference has
// no SpMM).  The pruned-NN layers the BITMAP format exists for (transformer-50: 512 x 33 288, half the positions set) are multiplied
// with a BATCH of activations in practice: masks and values -- all of the HBM traffic of the SpMV -- are streamed ONCE for the kVecs
// vectors; only the x reads (L2) and the multiply-adds grow.
//
// Same walk as spmv_bitmap_kernel (spmv_bitmap.hip: one wavefront per run of (row, 64-column group) steps, vector mask loads four
// ahead + v_readlane, every lane loads values[running offset + set bits below it], consume under exec = mask, two batches in flight),
// without its profiling builds.  What differs:
//   * x is INTERLEAVED: word [column][vector], so a lane's kVecs x words of a step are one 8- or 16-byte buffer load
//     (interleave_vectors_kernel makes that layout from the caller's column-major X);
//   * kVecs private sums per lane, kVecs wavefront reductions per row, accumulators ys[vector][row] in LDS;
//   * column j of Y goes to y + j * ldy.
// Arithmetic per column is exactly the SpMV kernel's (fixed: individually rounded / saturated products summed in 64 bits and clamped
// once; float: fp32 products, eight of them added in fp32 per batch, double across batches and lanes, rounded to fp32 once per row).
#include <hip/hip_runtime.h>

#include "spmv_device.h"
#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr int kBmThreads = kWaveLanes * kBitmapWaves;   // 1024
constexpr int kBatch = 8;
constexpr uint32_t kRsrcFlags = 0x00020000u;            // raw 32-bit buffer, gfx9 family

template <int kVecs>
struct XWords;
template <>
struct XWords<2> {
    uint32_t w[2];
    static __device__ __forceinline__ XWords load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        typedef uint32_t v2 __attribute__((ext_vector_type(2)));
        const v2 t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        return XWords{{t.x, t.y}};
    }
};
template <>
struct XWords<4> {
    uint32_t w[4];
    static __device__ __forceinline__ XWords load(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        typedef uint32_t v4 __attribute__((ext_vector_type(4)));
        const v4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        return XWords{{t.x, t.y, t.z, t.w}};
    }
};

template <bool kFloat, int kVecs>
struct Batch {
    uint32_t v[kBatch];
    XWords<kVecs> xv[kBatch];
    uint64_t m[kBatch];        // wave-uniform
};

// One wavefront, one row: groups [0, steps) of the run that starts at mask `mp`, value `vp`, column `col0`; acc[j] += the lane's share of
// row . X[:, j]
template <bool kFloat, int kVecs>
__device__ __forceinline__ void bitmap_row_run(const uint64_t* mp, const uint32_t*& vp, const uint32_t* xi, uint32_t num_cols, uint32_t col0,
                                               uint32_t steps, uint32_t lane, bool have_first, uint32_t first_masks,
                                               typename Rows<kFloat>::sum_t (&acc)[kVecs]) {
    using R = Rows<kFloat>;
    const auto vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(vp), 0, 0x7fffffffu, kRsrcFlags);
    uint32_t xk[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) xk[k] = (lane + k * kBitmapGroupCols) * (4u * kVecs);
    uint32_t voff = 0;
    const auto mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t*>(mp), 0, steps * 8u, kRsrcFlags);
    uint32_t moff = lane * 4u;
    uint32_t mcur = have_first ? first_masks : __builtin_amdgcn_raw_buffer_load_b32(mr, moff, 0, 0);
    uint32_t m1 = __builtin_amdgcn_raw_buffer_load_b32(mr, moff + 256u, 0, 0);
    uint32_t m2 = __builtin_amdgcn_raw_buffer_load_b32(mr, moff + 512u, 0, 0);
    uint32_t m3 = __builtin_amdgcn_raw_buffer_load_b32(mr, moff + 768u, 0, 0);
    auto issue = [&](Batch<kFloat, kVecs>& b, uint32_t bb) {
        const uint32_t sel = (bb & 3u) * (2 * kBatch);
        const uint32_t xc = min(col0 + bb * (kBatch * kBitmapGroupCols), num_cols);      // first column of the batch (scalar)
        // the last group of a row may hang over the end of x: range-checked by the descriptor (reads 0)
        const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(xi + size_t(xc) * kVecs), 0, (num_cols - xc) * (4u * kVecs), kRsrcFlags);
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const uint32_t lo = __builtin_amdgcn_readlane(mcur, sel + 2 * k), hi = __builtin_amdgcn_readlane(mcur, sel + 2 * k + 1);
            b.m[k] = (static_cast<uint64_t>(hi) << 32) | lo;
            const uint32_t off = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0)) * 4u;
            b.v[k] = __builtin_amdgcn_raw_buffer_load_b32(vr, off, voff, 0);
            b.xv[k] = XWords<kVecs>::load(xr, xk[k]);
            voff += static_cast<uint32_t>(__builtin_popcountll(b.m[k])) * 4u;
        }
    };
    auto consume = [&](const Batch<kFloat, kVecs>& b) {
        typename R::prod_t part[kVecs];
#pragma unroll
        for (int j = 0; j < kVecs; ++j) part[j] = 0;
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (__builtin_amdgcn_inverse_ballot_w64(b.m[k])) {
#pragma unroll
                for (int j = 0; j < kVecs; ++j) {
                    if (kFloat) part[j] += R::product(b.v[k], b.xv[k].w[j]);
                    else acc[j] += R::widen(R::product(b.v[k], b.xv[k].w[j]));
                }
            }
        }
        if (kFloat) {
#pragma unroll
            for (int j = 0; j < kVecs; ++j) acc[j] += R::widen(part[j]);
        }
    };
    Batch<kFloat, kVecs> A, B;
    auto rotate_masks = [&](uint32_t bb) {
        if ((bb & 3u) == 0) {
            mcur = m1; m1 = m2; m2 = m3;
            moff += 256u;
            m3 = __builtin_amdgcn_raw_buffer_load_b32(mr, moff + 768u, 0, 0);
        }
    };
    issue(A, 0);
    issue(B, 1);
    for (uint32_t bb = 0; bb * kBatch < steps; bb += 2) {
        consume(A);
        rotate_masks(bb + 2);
        issue(A, bb + 2);
        consume(B);
        rotate_masks(bb + 3);
        issue(B, bb + 3);
    }
    vp += voff / 4u;
}

template <bool kFloat, int kVecs>
__global__ __launch_bounds__(kBmThreads) void spmm_bitmap_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks,
                                                                const Unit* __restrict__ units, const uint32_t* __restrict__ xi, uint32_t num_cols,
                                                                uint32_t* __restrict__ y, uint64_t ldy) {
    using R = Rows<kFloat>;
    using acc_t = typename R::acc_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    acc_t* ys = reinterpret_cast<acc_t*>(lds);                    // [kVecs][nrows + 1]
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    uint32_t wg = blockIdx.x;                                     // same XCD-aware remap as the SpMV kernels
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    bool first_block = true;
    for (uint32_t bi = wg, next = 0;; bi = next) {
        const BlockTable blk = (BlockTable)(blocks + bi);
        next = blk->next;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset, stride = nrows + 1;
        const uint32_t* hdr = reinterpret_cast<const uint32_t*>(units) + (static_cast<size_t>(bi) * kBitmapWaves + wave) * (kBitmapRunSlots * 16);
        const uint32_t seg_word = hdr[min(lane, 15u)];
        const uint32_t first_masks = hdr[16 + lane];
        const uint32_t row_begin = __builtin_amdgcn_readlane(seg_word, 0), row_end = __builtin_amdgcn_readlane(seg_word, 1);
        const uint32_t g_begin = __builtin_amdgcn_readlane(seg_word, 2), steps = __builtin_amdgcn_readlane(seg_word, 3) - g_begin;
        const uint64_t* mp = reinterpret_cast<const uint64_t*>(image) +
                             ((static_cast<uint64_t>(__builtin_amdgcn_readlane(seg_word, 7)) << 32) | __builtin_amdgcn_readlane(seg_word, 6));
        const uint32_t* vp = reinterpret_cast<const uint32_t*>(image) +
                             ((static_cast<uint64_t>(__builtin_amdgcn_readlane(seg_word, 5)) << 32) | __builtin_amdgcn_readlane(seg_word, 4));
        const uint32_t col0 = blk->first_col0 + g_begin * kBitmapGroupCols;

        if (!first_block) __syncthreads();   // the previous block's result store has read the accumulators
        first_block = false;
        for (uint32_t i = tid; i < stride * kVecs; i += kBmThreads) ys[i] = 0;
        __syncthreads();
        for (uint32_t r = row_begin; r < row_end; ++r) {
            typename R::sum_t mine[kVecs];
#pragma unroll
            for (int j = 0; j < kVecs; ++j) mine[j] = 0;
            bitmap_row_run<kFloat, kVecs>(mp, vp, xi, num_cols, col0, steps, lane, r == row_begin, first_masks, mine);
            mp += (steps + 7u) / 8u * 8u + 16u;     // the run's masks + its zero padding (bitmap_tiles.cpp)
#pragma unroll
            for (int j = 0; j < kVecs; ++j) {
                const typename R::sum_t total = wave_sum(mine[j]);
                if (lane == 0) R::add_sum(ys, j * stride + r, total);
            }
        }
        // no-return LDS atomics can outlive s_waitcnt lgkmcnt(0) (spmv_kernels.hip): a RETURNING atomic per wavefront, awaited
        const acc_t flushed = atomicAdd(ys + nrows, static_cast<acc_t>(0));
        asm volatile("" ::"v"(flushed));
        __syncthreads();
        for (uint32_t i = tid; i < nrows * kVecs; i += kBmThreads) {
            const uint32_t j = i / nrows, r = i - j * nrows;
            y[size_t(j) * ldy + out0 + r] = R::finish(ys[j * stride + r]);
        }
        if (!next) break;
    }
}

// X columns (vector j at x + j * ldx) -> interleaved words [column][vector]
template <int kVecs>
__global__ __launch_bounds__(256) void interleave_vectors_kernel(const uint32_t* __restrict__ x, uint64_t ldx, uint32_t num_cols, uint32_t* __restrict__ xi) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= num_cols) return;
    uint32_t w[kVecs];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) w[j] = x[size_t(j) * ldx + c];
#pragma unroll
    for (int j = 0; j < kVecs; ++j) xi[size_t(c) * kVecs + j] = w[j];
}

template <bool kFloat, int kVecs>
hipError_t launch(const SpmmLaunch& a, hipStream_t stream) {
    const uint32_t lds = (a.max_block_rows + 1) * kVecs * uint32_t(sizeof(typename Rows<kFloat>::acc_t));
    // (per launch, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE, and a process may drive several)
    const hipError_t configured = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_bitmap_kernel<kFloat, kVecs>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, int(kMaxLdsBytes));
    if (configured != hipSuccess) return configured;
    hipLaunchKernelGGL((interleave_vectors_kernel<kVecs>), dim3((a.num_cols + 255) / 256), dim3(256), 0, stream, a.x, a.ldx, a.num_cols, a.x_interleaved);
    hipLaunchKernelGGL((spmm_bitmap_kernel<kFloat, kVecs>), dim3(a.num_workgroups), dim3(kBmThreads), lds, stream, a.image, a.blocks, a.units,
                       a.x_interleaved, a.num_cols, a.y, a.ldy);
    return hipGetLastError();
}

}  // namespace

uint32_t spmm_bitmap_max_block_rows(bool is_float, uint32_t vectors) {
    const uint32_t acc = is_float ? sizeof(Rows<true>::acc_t) : sizeof(Rows<false>::acc_t);
    return kMaxLdsBytes / (vectors * acc) - 1;
}

hipError_t launch_spmm_bitmap(bool is_float, const SpmmLaunch& a, hipStream_t stream) {
    if (a.num_workgroups == 0) return hipSuccess;
    if (a.vectors == 4) return is_float ? launch<true, 4>(a, stream) : launch<false, 4>(a, stream);
    if (a.vectors == 2) return is_float ? launch<true, 2>(a, stream) : launch<false, 2>(a, stream);
    return hipErrorInvalidValue;
}

}  // namespace dev
}  // namespace hisparse
