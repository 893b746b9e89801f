// spmv_bitmap.hip — the gfx950 SpMV kernel for dense-row matrices in the BITMAP format (stream_tiles.h, bitmap_tiles.cpp).
//
// Same architecture as spmv_kernels.hip -- a workgroup owns rows, their sums live in its LDS, y is written once -- but for
// rows that are dense enough (>= 1/8 of the columns set; the pruned-NN layers of sw/bm.sh:21-27) the FPGA's two gather
// stages have nothing left to do:
//   CPSR_matrix_loader (spmv_cluster.h:73-98)       a wavefront step = one 64-column GROUP of one row: a 64-bit occupancy mask
//     streams {value, column} pairs                  + the values of the set columns, compacted (4 B + 1 bit per position, not 8 B)
//   vecbuf_access_unit + shuffle 1 (x[col] gather)   x[64 g + lane]: no gather and no per-sub-tile barrier -- a conflict-free ds_read_b32 from the
//                                                    block's whole stretch of x, copied into LDS once per workgroup (kXLds; up to
//                                                    36 864 columns), or a coalesced 256-byte read straight from L2 where that does not fit
//   shuffle 2 + PE accumulate (pe.h:62-81)           per-lane register sums along the wavefront's run of groups, ONE wavefront-wide
//                                                    sum and ONE LDS add per (wavefront, row)
//   result packer + drain                            coalesced store, AP_SAT clamp / fp32 rounding once (column-sliced blocks write
//                                                    partials, combine_slices_kernel adds them)
// Lane l of a step takes column 64 g + l: its value sits at (values before this group) + (set bits below l) = a scalar running
// offset + v_mbcnt(mask).  All global loads are BUFFER loads: the x read of a row's last (partial) group is range-checked by the
// hardware, and so are mask reads past the end of a run -- the whole inner loop is branch-free straight-line code, which is what lets
// hipcc count its own s_waitcnt (the loads of the next batch stay in flight while a batch is consumed).
// The 16 wavefronts' runs are NOT equal: a SIMD serves its oldest wavefront first and this loop is hungry for issue slots, so the builder
// weights the shares by the wavefront's place on its SIMD (bitmap_tiles.cpp, kBitmapSkew) and all sixteen finish together.
// Numerics: Q8.24 products rounded/saturated one by one and summed exactly in 64 bits -- bit-identical to every other format and to
// the oracle.  Float: one fp32 multiply per product (no FMA); the (up to 8) products of a batch are added in fp32, the batch sums
// join the lane's double sum, one double LDS add per (wavefront, row) -- the order of those adds is not fixed when a row is split over
// several wavefronts, so float results are TOLERANCE parity (1e-4, like every float path here), not bit-identical to the PAIRS / DELTA
// paths or from launch to launch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "spmv_device.h"
#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr int kBmThreads = kWaveLanes * kBitmapWaves;   // 1024
constexpr int kBatch = 8;                               // steps per batch: 16 loads issued back to back
constexpr uint32_t kInvalidOffset = 0x80000000u;        // beyond any num_records below: the load returns 0 (no memory access)
constexpr uint32_t kRsrcFlags = 0x00020000u;            // raw 32-bit buffer, gfx9 family
constexpr int kXCopyPasses = (kBitmapMaxXLdsGroups * (kBitmapGroupCols / 4) + kBmThreads - 1) / kBmThreads;   // 16-byte loads per thread that cover the largest x stretch kept in LDS

typedef const __attribute__((address_space(4))) WaveSeg* WaveSegTable;

template <bool kFloat>
struct Batch {
    uint32_t v[kBatch], xv[kBatch];
    uint64_t m[kBatch];        // wave-uniform (SGPR pairs)
};

// One wavefront, one row: groups [0, steps) of the run that starts at mask `mp`, value `vp`, column `col0`.  Returns the lane's sum.
// kAblate (profiling builds, HISPARSE_ABLATE): bit 0 = no value loads, bit 1 = no x loads, bit 2 = no arithmetic (wrong results);
// 64 = timeline, 256 = nt cache policy on the value loads (correct results)
typedef const __attribute__((address_space(3))) uint32_t* LdsWords;

// A wave-uniform pointer, said so: a buffer descriptor built from a pointer hipcc cannot PROVE uniform is loaded under a "waterfall" loop
// (v_readfirstlane + compare + branch around every load) -- seen in one instantiation of this kernel, where nothing is divergent.
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32)));     // (the builtin returns int: no sign extension)
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v)));
    return reinterpret_cast<T*>((static_cast<uint64_t>(hi) << 32) | lo);
}

// kXLds: the block's stretch of x has been copied into LDS (xs_run = the word of the run's first column); the x operand of a step is then
// a conflict-free ds_read_b32 (lane l reads word 64 g + l) instead of a second buffer load per step -- the loads are what the run waits for
// (ablation builds, round 3: no x loads 11.6 us against 14.1), and a block's x stretch is read once per row AND piece otherwise.
template <bool kFloat, int kAblate, bool kXLds>
__device__ __forceinline__ typename Rows<kFloat>::sum_t bitmap_row_run(const uint64_t* mp, const uint32_t*& vp, const uint32_t* x, uint32_t num_cols,
                                                                       uint32_t col0, uint32_t steps, uint32_t lane, bool have_first,
                                                                       const uint32_t (&first_masks)[4], LdsWords xs_run, uint64_t* stamps = nullptr) {
    using R = Rows<kFloat>;
    typename R::sum_t acc = 0;
    // values: this run's compacted values; offset = running scalar byte offset + 4 * (set bits below the lane)
    const auto vr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<uint32_t*>(vp)), 0, 0x7fffffffu, kRsrcFlags);
    // x: a fresh descriptor per batch whose base is the batch's first column and whose length is what is left of the vector
    // (scalar arithmetic only): the per-step lane offsets are then eight loop-invariant registers, and the last group of a row,
    // which may hang over the end of x, is still range-checked
    uint32_t xk[kBatch];
#pragma unroll
    for (int k = 0; k < kBatch; ++k) xk[k] = (lane + k * kBitmapGroupCols) * 4u;
    uint32_t voff = 0;                                   // bytes into the run's values (scalar)
    // masks: the 8 masks of a batch = 16 dwords, fetched by lanes 0-15 of ONE vector load (the other lanes aim beyond the descriptor), four
    // batches ahead.  (Until round 3 a vector held 32 masks -- four batches -- and v_readlane picked dword (bb % 4) * 16 + j with a
    // computed lane select: two scalar adds per step in a loop that turned out to be bound by instruction issue.  Now the selects are the
    // constants 0..15 and a batch costs one more load instruction.)  Reads past the run's own masks find the zero masks the image pads
    // every run with (bitmap_tiles.cpp), or 0 from the range check (the advancing part of the offset is in the VECTOR offset: the scalar
    // offset is not range-checked): no tail code anywhere.
    // (Fetching the masks through the scalar cache instead -- s_load_dwordx16 = one batch, no v_readlane -- was tried: the
    // scalar loads must be requested a batch ahead with hand-placed waits, and hipcc copies the destination registers of an asm
    // load before the wait; not worth reserving registers for.)
    const auto mr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<uint64_t*>(mp)), 0, steps * 8u, kRsrcFlags);
    uint32_t moff = lane < 2 * kBatch ? lane * 4u : kInvalidOffset;
    constexpr uint32_t kBatchMaskBytes = kBatch * 8u;
    // the first four batches' masks of a wavefront's run arrive with its descriptor (one round trip less in front of the first value load)
    uint32_t mcur = have_first ? first_masks[0] : __builtin_amdgcn_raw_buffer_load_b32(mr, moff, 0, 0);
    uint32_t m1 = have_first ? first_masks[1] : __builtin_amdgcn_raw_buffer_load_b32(mr, moff + kBatchMaskBytes, 0, 0);
    uint32_t m2 = have_first ? first_masks[2] : __builtin_amdgcn_raw_buffer_load_b32(mr, moff + 2 * kBatchMaskBytes, 0, 0);
    uint32_t m3 = have_first ? first_masks[3] : __builtin_amdgcn_raw_buffer_load_b32(mr, moff + 3 * kBatchMaskBytes, 0, 0);
    // batch bb (8 steps, 16 loads); its masks are dwords 0 .. 15 of the current mask vector.  ONE copy of this code serves every batch
    // and the kernel stays a few KiB (every launch starts with a cold instruction cache).
    auto issue = [&](Batch<kFloat>& b, uint32_t bb) {
        constexpr uint32_t sel = 0;
        const uint32_t xc = min(col0 + bb * (kBatch * kBitmapGroupCols), num_cols);      // first column of the batch (scalar)
        const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(x + xc), 0, (num_cols - xc) * 4u, kRsrcFlags);
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const uint32_t lo = __builtin_amdgcn_readlane(mcur, sel + 2 * k), hi = __builtin_amdgcn_readlane(mcur, sel + 2 * k + 1);
            b.m[k] = (static_cast<uint64_t>(hi) << 32) | lo;
            // EVERY lane loads values[running offset + set bits below it]: for a lane whose own bit is clear that is the value of
            // the next set column (or the first value of the next step) -- same cache lines, no extra traffic, and consume() never
            // looks at it.  Cheaper than steering those lanes to an out-of-range offset (two more vector instructions per step).
            const uint32_t off = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0)) * 4u;
            b.v[k] = (kAblate & 1) ? off : __builtin_amdgcn_raw_buffer_load_b32(vr, off, voff, (kAblate & 256) ? 2 : 0);
            if constexpr (kXLds) b.xv[k] = (kAblate & 2) ? lo : xs_run[(bb * kBatch + k) * kBitmapGroupCols + lane];
            else b.xv[k] = (kAblate & 2) ? lo : __builtin_amdgcn_raw_buffer_load_b32(xr, xk[k], 0, 0);
            // voff += 4 * popcount: one s_bcnt1 + one s_lshl2_add (hipcc keeps a running count, shifts it and adds the base: four)
            const uint32_t set = static_cast<uint32_t>(__builtin_popcountll(b.m[k]));
            asm("s_lshl2_add_u32 %0, %1, %2" : "=s"(voff) : "s"(set), "s"(__builtin_amdgcn_readfirstlane(voff)) : "scc");      // (readfirstlane: a no-op that tells hipcc the value is uniform)
        }
    };
    // Only the lanes whose bit is set take part; in float mode the eight products of a batch are added up in fp32 first and join
    // the double sum once per batch (one conversion + one double add instead of eight).
    auto consume = [&](const Batch<kFloat>& b) {
        typename R::prod_t part = 0;
        if constexpr (kFloat && !(kAblate & 4) && (kBatch % 2) == 0) {
            // two steps per multiply and per add (v_pk_mul_f32 / v_pk_add_f32 on register pairs): the loop is bound by vector-ALU issue.
            // The products are the same fp32 products; their fp32 sum inside a batch now runs in two interleaved chains.
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 part2 = {0.f, 0.f};
#pragma unroll
            for (int k = 0; k < kBatch; k += 2) {
                const f2 v = {__uint_as_float(b.v[k]), __uint_as_float(b.v[k + 1])}, xx = {__uint_as_float(b.xv[k]), __uint_as_float(b.xv[k + 1])};
                f2 prod = v * xx;
                if (!__builtin_amdgcn_inverse_ballot_w64(b.m[k])) prod.x = 0.f;
                if (!__builtin_amdgcn_inverse_ballot_w64(b.m[k + 1])) prod.y = 0.f;
                part2 += prod;
            }
            acc += R::widen(part2.x + part2.y);
            return;
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            if (kAblate & 4) { acc += static_cast<typename R::sum_t>(b.v[k] ^ b.xv[k]); continue; }
            if (__builtin_amdgcn_inverse_ballot_w64(b.m[k])) {
                if (kFloat) part += R::product(b.v[k], b.xv[k]);
                else acc += R::widen(R::product(b.v[k], b.xv[k]));
            }
        }
        if (kFloat) acc += R::widen(part);
    };
    // (Round 3: 128-column steps -- lane l takes columns 2 l and 2 l + 1 of a pair of groups with one 8-byte value load and one 8-byte x
    // load, half the round trips -- were built and measured at 23.9 us against 13.8: picking a lane's two bits and its prefix count out of
    // two standard masks costs ~20 vector instructions and 4 scalar registers per step, the unrolled batch spilled 59 SGPRs, and the
    // larger code pays twice at the cold instruction cache of every launch.  The timeline says the run is bound by round trips
    // (transformer-80, with 2.5 x fewer non-zeros, takes the same 7.0 us for its runs), so wide steps remain the lever -- but they need
    // masks stored as (even columns, odd columns) pairs so that v_mbcnt and SGPR-pair selects do the work; an image-format change.
    // A second experiment says that would not pay either: batches of 16 steps (twice the loads in flight, per-lane set bits instead of
    // masks in SGPRs) ran 20.3 us against 16.2, same box, with the run phase at 7.6 us instead of 7.0.  So the run is bound neither by
    // bytes nor by round trips but by vector-memory INSTRUCTIONS: 266 K steps x 2 loads / 256 CUs = 2080 wave-loads per CU in 7 us =
    // one per 7 clocks, and a 64-lane dword load occupies the CU's address path for 4 of them whether the lanes carry 32 useful
    // values (50 % density) or 13 (20 %).  The levers left are structural: several ROWS of the same group range per wavefront (one x
    // load for all of them) and compacted value loads handed to the lanes through ds_bpermute.)
    // A third experiment: ablation builds say that with every load and all arithmetic removed the run still takes 5.8 of its 9.6 us (8
    // vector-ALU instructions per step), so the eight multiply-adds of a batch were put under exec = mask in one asm block (hipcc emits
    // multiply + v_cndmask + add: one vector instruction more per step).  Slower, 16.5 against 15.8 us on the same box: the asm block has to
    // wait for the LAST of the batch's 16 loads before its first multiply, while hipcc's code consumes the loads as they land
    // (vmcnt(14), (12), ...) -- the loop lives on that overlap, not on instruction count.
    // Two batches: 16 loads of the next one are in flight while a batch is consumed.  Three in flight measured SLOWER (16.6 ->
    // 17.7 us on transformer-50), and so did the nt policy on the value loads (-> 17.6 us: the 36 MB image lives in the 256 MiB
    // Infinity Cache between launches); the timeline (tools/bitmap_timeline.py) shows the run itself streaming at ~5.5 TB/s and
    // ~7 us of the kernel's life spent in front of it (dispatch ramp, descriptor -> masks -> first values: three dependent round
    // trips) and behind it (wavefronts finish 6 us apart, the last one holds the barrier).
    Batch<kFloat> A, B;
    auto rotate_masks = [&](uint32_t) {    // the next batch's masks become current; the vector four batches ahead is requested
        mcur = m1; m1 = m2; m2 = m3;
        moff += kBatchMaskBytes;
        m3 = __builtin_amdgcn_raw_buffer_load_b32(mr, moff + 3 * kBatchMaskBytes, 0, 0);
    };
    issue(A, 0);
    rotate_masks(1);
    if (kAblate & 64) {   // timeline build: [2] = the first masks have arrived (issue() has used them)
        uint64_t t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(voff));
        if (lane == 0) stamps[2] = t;
    }
    issue(B, 1);
    for (uint32_t bb = 0; bb * kBatch < steps; bb += 2) {
        consume(A);
        if ((kAblate & 64) && bb == 0) {   // [3] = the first batch of values and x has arrived and been consumed
            uint64_t t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(acc));
            if (lane == 0) stamps[3] = t;
        }
        rotate_masks(bb + 2);
        issue(A, bb + 2);                // batches past the end find all-zero masks: no memory traffic
        consume(B);
        rotate_masks(bb + 3);
        issue(B, bb + 3);
    }
    vp += voff / 4u;
    return acc;
}

template <bool kFloat, int kAblate, bool kXLds = false>
__global__ __launch_bounds__(kBmThreads) void spmv_bitmap_kernel(const uint8_t* __restrict__ image, const Block* __restrict__ blocks,
                                                                const Unit* __restrict__ units, const uint32_t* __restrict__ x, uint32_t num_cols,
                                                                uint32_t* __restrict__ out, int32_t row_part_filter,
                                                                const uint32_t* __restrict__ part_heads, uint64_t* __restrict__ timeline,
                                                                uint32_t x_lds_offset, CarriedCombine carry) {
    using R = Rows<kFloat>;
    using acc_t = typename R::acc_t;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    acc_t* ys = reinterpret_cast<acc_t*>(lds);                    // [nrows + 1]
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & (kWaveLanes - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid / kWaveLanes);
    // timeline build (kAblate & 64): eight 100 MHz timestamps per wavefront: 0 entry, 1 descriptors read, 2 first masks in,
    // 3 first batch consumed, 4 run finished, 5 row sums in LDS, 6 after the barrier, 7 results stored
    uint64_t* stamps = (kAblate & 64) ? timeline + (static_cast<size_t>(blockIdx.x) * kBitmapWaves + wave) * 8 : nullptr;
    auto stamp = [&](int i, uint32_t dep) {
        if (!(kAblate & 64)) return;
        uint64_t t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(dep));
        if (lane == 0) stamps[i] = t;
    };
    stamp(0, wave);
    uint32_t wg = blockIdx.x;                                     // same XCD-aware remap as spmv_rowblock_kernel
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (carry.partial) carried_combine<kFloat, kBmThreads>(carry, blockIdx.x, gridDim.x, tid);      // y of the PREVIOUS step (spmv_device.h)
    uint32_t bi = wg;
    if (row_part_filter >= 0) {
        bi = ((const __attribute__((address_space(4))) uint32_t*)part_heads)[static_cast<uint32_t>(row_part_filter) * gridDim.x + wg];
        if (bi == kNoBlock) return;
    }
    bool first_block = true;
    uint32_t staged_col0 = 0, staged_groups = 0;      // kXLds: the stretch of x the LDS holds
    for (uint32_t next = 0;; bi = next) {
        const BlockTable blk = (BlockTable)(blocks + bi);
        next = (row_part_filter >= 0 && blk->next_part > static_cast<uint32_t>(row_part_filter)) ? 0u : blk->next;
        const uint32_t nrows = blk->nrows, out0 = blk->out_offset;
        // the block's 16 wavefront runs sit at units[16 bi ..] (the builder stores them in final block order): descriptor and run
        // are fetched side by side, one dependent round trip before the first mask load instead of two
        // Run header of this wavefront: 64 bytes of WaveSeg + the run's first 32 masks, fetched with two VECTOR loads whose
        // addresses depend on nothing but the block index -- descriptor and first masks in one round trip (timeline before:
        // descriptors 0.8 us, then masks 2.0 us, then the first values).
        const uint32_t* hdr = reinterpret_cast<const uint32_t*>(units) + (static_cast<size_t>(bi) * kBitmapWaves + wave) * (kBitmapRunSlots * 16);
        const uint32_t seg_word = hdr[min(lane, 15u)];
        // (the copy of the run's first 32 masks: batch b's 16 dwords in lanes 0-15 of vector b)
        uint32_t first_masks[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) first_masks[b] = hdr[16 + 16 * b + (lane & 15u)];
        const uint32_t row_begin = __builtin_amdgcn_readlane(seg_word, 0), row_end = __builtin_amdgcn_readlane(seg_word, 1);
        const uint32_t g_begin = __builtin_amdgcn_readlane(seg_word, 2), steps = __builtin_amdgcn_readlane(seg_word, 3) - g_begin;
        const uint64_t* mp = reinterpret_cast<const uint64_t*>(image) +
                             ((static_cast<uint64_t>(__builtin_amdgcn_readlane(seg_word, 7)) << 32) | __builtin_amdgcn_readlane(seg_word, 6));
        const uint32_t* vp = reinterpret_cast<const uint32_t*>(image) +
                             ((static_cast<uint64_t>(__builtin_amdgcn_readlane(seg_word, 5)) << 32) | __builtin_amdgcn_readlane(seg_word, 4));
        const uint32_t col0 = blk->first_col0 + g_begin * kBitmapGroupCols;
        stamp(1, col0 + steps + nrows);

        if (!first_block) __syncthreads();   // the previous block's result store has read the accumulators
        for (uint32_t i = tid; i <= nrows; i += kBmThreads) ys[i] = 0;           // PE banks start at zero (pe.h:131-135)
        LdsWords xs_run = nullptr;
        if constexpr (kXLds) {
            // the block's stretch of x -> LDS, 16 bytes per thread and pass, every load in flight before the first LDS write (the copy
            // rides on the run headers' round trip); words past the end of x are range-checked to 0.  A workgroup's next block of the
            // same column slice finds it there.
            const uint32_t c0 = blk->first_col0, groups = blk->first_ncols;
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            auto* xs = (__attribute__((address_space(3))) u4*)(lds + x_lds_offset);
            if (first_block || c0 != staged_col0 || groups != staged_groups) {
                const auto xr = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(const_cast<uint32_t*>(x + min(c0, num_cols))), 0, (num_cols - min(c0, num_cols)) * 4u, kRsrcFlags);
                const uint32_t quads = groups * (kBitmapGroupCols / 4);
                u4 r[kXCopyPasses];
#pragma unroll
                for (int j = 0; j < kXCopyPasses; ++j) r[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, (tid + j * kBmThreads) * 16u, 0, 0);
#pragma unroll
                for (int j = 0; j < kXCopyPasses; ++j)
                    if (tid + j * kBmThreads < quads) xs[tid + j * kBmThreads] = r[j];
                staged_col0 = c0;
                staged_groups = groups;
            }
            xs_run = (LdsWords)xs + g_begin * kBitmapGroupCols;
        }
        first_block = false;
        __syncthreads();
        for (uint32_t r = row_begin; r < row_end; ++r) {
            const typename R::sum_t mine = (kAblate & 8) ? typename R::sum_t(steps) : bitmap_row_run<kFloat, kAblate, kXLds>(mp, vp, x, num_cols, col0, steps, lane, r == row_begin, first_masks, xs_run, stamps);
            if (kAblate & 64) { uint64_t t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(mine)); if (lane == 0) stamps[4] = t; }
            mp += (steps + 7u) / 8u * 8u + 16u;     // the run's masks + its zero padding (bitmap_tiles.cpp)
            if (kAblate & 16) { asm volatile("" ::"v"(mine)); continue; }
            const typename R::sum_t total = wave_sum(mine);
            if (lane == 0) R::add_sum(ys, r, total);
        }
        // no-return LDS atomics can outlive s_waitcnt lgkmcnt(0) (spmv_kernels.hip): a RETURNING atomic per wavefront, awaited
        const acc_t flushed = atomicAdd(ys + nrows, static_cast<acc_t>(0));
        asm volatile("" ::"v"(flushed));
        stamp(5, nrows);
        __syncthreads();
        stamp(6, nrows);
        for (uint32_t i = tid; i < nrows; i += kBmThreads) out[out0 + i] = R::finish(ys[i]);
        if (kAblate & 64) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(7, nrows); }
        if (!next) break;
    }
}

// product library: the two numeric modes; the ablation / timeline instantiations only in libhisparse_hip_prof.so (spmv_kernels.hip)
#ifdef HISPARSE_PROFILING
#define HS_FOR_EACH_BITMAP_XLDS_VARIANT(X) X(false, 0) X(true, 0) X(true, 1) X(true, 2) X(true, 3) X(true, 7) X(true, 15) X(true, 64)
#define HS_FOR_EACH_BITMAP_VARIANT(X) X(false, 0) X(true, 0) X(true, 1) X(true, 2) X(true, 3) X(true, 4) X(true, 7) X(true, 15) X(true, 31) X(true, 23) X(true, 64) X(true, 256)
#else
#define HS_FOR_EACH_BITMAP_XLDS_VARIANT(X) X(false, 0) X(true, 0)
#define HS_FOR_EACH_BITMAP_VARIANT(X) X(false, 0) X(true, 0)
#endif

}  // namespace

hipError_t configure_bitmap_kernels(uint32_t lds_bytes) {
    hipError_t e;
#define X(F, A) \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_bitmap_kernel<F, A>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes))) != hipSuccess) return e;
    HS_FOR_EACH_BITMAP_VARIANT(X)
#undef X
#define X(F, A) \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmv_bitmap_kernel<F, A, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes))) != hipSuccess) return e;
    HS_FOR_EACH_BITMAP_XLDS_VARIANT(X)
#undef X
    return hipSuccess;
}

hipError_t launch_spmv_bitmap(bool is_float, const SpmvLaunch& a, hipStream_t stream) {
    if (a.num_workgroups == 0) return hipSuccess;
    const dim3 grid(a.num_workgroups), block(kBmThreads);
    const CarriedCombine carry = carried(a);
    int ablate = 0, depth_unused = 8;       // read per launch (profiling library only): a process may switch profiling builds between runs
    if (!profiling_switches(ablate, depth_unused)) return hipErrorInvalidValue;
    // timeline build: HISPARSE_ABLATE=64 HISPARSE_TIMELINE_OUT=file -> every launch is synchronised and its per-wavefront
    // timestamps (workgroups x 16 x 8 u64, 100 MHz) overwrite the file (tools/bitmap_timeline.py reads it)
    // (one buffer per device: a process may drive several, benchmark.cpp --gpus N)
    static uint64_t* timelines[16] = {};
    uint64_t* timeline = nullptr;
    if (ablate & 64) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
        if (!timelines[dev]) (void)hipMalloc(reinterpret_cast<void**>(&timelines[dev]), size_t(4096) * kBitmapWaves * 8 * sizeof(uint64_t));
        timeline = timelines[dev];
    }
    bool launched = false;
    // a.bitmap_x_groups != 0: the launch's LDS has room for that many groups of x behind the accumulators (hs_api.cpp sizes it)
    const uint32_t x_lds_offset = a.bitmap_x_groups ? a.lds_bytes - a.bitmap_x_groups * kBitmapGroupCols * 4u : 0u;
#define X(F, A)                                                                                                                                  \
    if (!launched && a.bitmap_x_groups && is_float == F && ablate == A) {                                                                        \
        hipLaunchKernelGGL((spmv_bitmap_kernel<F, A, true>), grid, block, a.lds_bytes, stream, a.image, a.blocks, a.units, a.x, a.num_cols, a.out, \
                           a.row_part_filter, a.part_heads, timeline, x_lds_offset, carry);                                                             \
        launched = true;                                                                                                                         \
    }
    HS_FOR_EACH_BITMAP_XLDS_VARIANT(X)
#undef X
#define X(F, A)                                                                                                                                  \
    if (!launched && is_float == F && ablate == A) {                                                                                             \
        hipLaunchKernelGGL((spmv_bitmap_kernel<F, A>), grid, block, a.bitmap_x_groups ? x_lds_offset : a.lds_bytes, stream, a.image, a.blocks, a.units, a.x, a.num_cols, a.out, \
                           a.row_part_filter, a.part_heads, timeline, 0u, carry);                                                                       \
        launched = true;                                                                                                                         \
    }
    HS_FOR_EACH_BITMAP_VARIANT(X)
#undef X
    if (!launched) return hipErrorInvalidValue;
    if ((ablate & 64) && timeline && a.num_workgroups <= 4096) {
        if (const char* path = std::getenv("HISPARSE_TIMELINE_OUT")) {
            (void)hipStreamSynchronize(stream);
            std::vector<uint64_t> host(size_t(a.num_workgroups) * kBitmapWaves * 8);
            (void)hipMemcpy(host.data(), timeline, host.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
            if (FILE* f = std::fopen(path, "wb")) {
                std::fwrite(host.data(), sizeof(uint64_t), host.size(), f);
                std::fclose(f);
            }
        }
    }
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
