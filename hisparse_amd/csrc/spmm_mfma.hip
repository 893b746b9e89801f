// spmm_mfma.hip — Y = A X for 16 dense vectors at once on the MATRIX ENGINE, over the second image of a float BITMAP matrix
// (EXTENSION, SURVEY.md section 8(f)-4: the reference has no SpMM -- spmv/libfpga/common.h:52-54 only stubs the neighbouring types --
// and the paper's section 7 names SpMM as the step that would make a dense engine relevant).
//
// Why the matrix engine here and nowhere else in this library: SpMV is 0.25 op/byte and bound by the stream.  A pruned-NN layer
// (transformer-50: 512 x 33 288, half the positions set) times a BATCH of 16 activation vectors is a 50 %-dense contraction whose cost
// on the vector ALUs is not the flops but the x traffic: the fused 4-column kernel (spmm_bitmap.hip) re-reads 16 bytes of x per lane
// and step for every ROW, 1.1 GB through the L2 for k = 16.  A 16 x 16 x 4 MFMA tile shares every x word among 16 rows and every
// matrix word among 16 vectors in registers: x traffic / 16, matrix streamed once.
//
// One wavefront = one UNIT: a tile of 16 rows x a chunk of 64-column groups (stream_tiles.h: MfmaImage).  Per group, 16 x
// v_mfma_f32_16x16x4_f32:  D[16 rows x 16 vectors] += A[16 rows x 4 columns] . B[4 columns x 16 vectors]
//   lane l = (i = l % 16, k = l / 16):   A operand = A[row i][column 4 t + k] -- bit 4 t + k of the lane's row mask decides between a
//                                        stored value and 0.0.  The image stores the values in exactly this (step, lane) order, so the
//                                        lanes whose bit is set read CONSECUTIVE words at (running base + set bits below the lane in
//                                        the step's ballot): one coalesced load per step.  (A first version gathered from row-major
//                                        values: 16 rows = 16 cache lines per load, 34 us for the kernel against 62 us per SpMM.)
//                                        B operand = X[column 4 t + k][vector l % 16] = word 64 t + l of the group's interleaved x:
//                                        one fully coalesced 256-byte load.
// The next group's 32 operands are loaded while this group's 16 MFMAs run (two accumulators alternate: a 16x16x4 f32 MFMA has 40
// cycles of dependent latency for 32 of issue).  Partial tiles of the units of one row tile are added (in double) by a second, small
// kernel that also rounds to fp32 and writes Y.
// Numerics: fp32 products accumulated by the MFMA's fused multiply-adds in fp32 (the SpMV kernels multiply, then add in double):
// tolerance parity (1e-4) per column against the oracle's SpMV of that column, like every float path.
// A clear bit multiplies 0.0 with x: for a non-finite x word that would poison rows that do not touch the column, so
// interleave_x16_kernel marks the call when it sees one; the MFMA kernel then does nothing and spmm_finish_kernel computes every row from
// its stored elements only (slow, rare, exact).
#include <hip/hip_runtime.h>

#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kVec = 16;                 // vectors per pass = the N of the MFMA tile
constexpr uint32_t kUnitThreads = 256;        // 4 wavefronts per workgroup, one unit each

struct Operands {
    float a[16], b[16];
};

// X columns (vector j at x + j * ldx) -> interleaved words [column][16 vectors], zero-filled up to `padded_cols` (whole groups); *flag
// becomes `call` when any word is not finite (a call number instead of a flag that would need zeroing first)
__global__ __launch_bounds__(256) void interleave_x16_kernel(const uint32_t* __restrict__ x, uint64_t ldx, uint32_t num_cols, uint32_t padded_cols,
                                                            uint32_t vectors, uint32_t* __restrict__ xi, uint32_t* __restrict__ flag, uint32_t call) {
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= padded_cols) return;
    uint32_t w[kVec];
    bool bad = false;
#pragma unroll
    for (uint32_t j = 0; j < kVec; ++j) {
        w[j] = (c < num_cols && j < vectors) ? x[size_t(j) * ldx + c] : 0u;       // (a pass of fewer than 16 vectors: zero vectors fill the tile)
        bad |= (w[j] & 0x7f800000u) == 0x7f800000u;
    }
    uint4* dst = reinterpret_cast<uint4*>(xi + size_t(c) * kVec);
#pragma unroll
    for (uint32_t q = 0; q < kVec / 4; ++q) dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    if (bad) *flag = call;
}

__global__ __launch_bounds__(kUnitThreads) void spmm_mfma_kernel(const uint32_t* __restrict__ words, uint64_t offsets_word, uint64_t values_word,
                                                                uint32_t groups, uint32_t chunk, uint32_t chunks, const float* __restrict__ xi,
                                                                const uint32_t* __restrict__ flag, uint32_t call, float* __restrict__ partial) {
    if (*flag == call) return;                  // a non-finite x word somewhere: the exact path computes this product (spmm_finish_kernel)
    __shared__ float tile_sum[3][kMfmaTileRows * kVec];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64u);
    const uint32_t unit = blockIdx.x * (kUnitThreads / 64u) + wave;       // chunks is a multiple of 4: the four units of a workgroup share a tile
    const uint32_t tile = unit / chunks, c = unit - tile * chunks;
    const uint32_t g0 = min(groups, c * chunk), g1 = min(groups, g0 + chunk);
    const uint32_t i = lane & 15u, k = lane >> 4;
    const uint64_t* masks = reinterpret_cast<const uint64_t*>(words) + (uint64_t(tile) * groups + g0) * kMfmaTileRows + i;
    const float* values = reinterpret_cast<const float*>(words + values_word);
    uint32_t base = __builtin_amdgcn_readfirstlane(words[offsets_word + unit]);   // value index of the unit's first value (scalar, running)
    const float* xg = xi + size_t(g0) * 64u * kVec + lane;                        // B operand of MFMA s of group g: xg[(g - g0) * 1024 + 64 s]

    // the 32 operands of one group: step s takes columns 4 s .. 4 s + 3; the lanes whose bit 4 s + k is set read consecutive values
    auto load = [&](Operands& o, uint32_t g, uint64_t m) {
#pragma unroll
        for (uint32_t s4 = 0; s4 < 16; ++s4) {
            const bool set = (m >> (4u * s4 + k)) & 1ull;
            const uint64_t vote = __ballot(set);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(vote >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(vote), 0u));
            const float v = values[base + rank];                                 // (a clear bit reads a neighbour's value: valid memory, not used)
            o.a[s4] = set ? v : 0.0f;
            o.b[s4] = xg[size_t(g - g0) * (64u * kVec) + 64u * s4];
            base += static_cast<uint32_t>(__builtin_popcountll(vote));
        }
    };
    f32x4 d0 = {0.0f, 0.0f, 0.0f, 0.0f}, d1 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (g0 < g1) {
        Operands cur, nxt;
        uint64_t m_next = g0 + 1 < g1 ? masks[kMfmaTileRows] : 0ull;
        load(cur, g0, masks[0]);
        for (uint32_t g = g0; g < g1; ++g) {
            const uint64_t m = m_next;
            m_next = g + 2 < g1 ? masks[uint64_t(g + 2 - g0) * kMfmaTileRows] : 0ull;
            if (g + 1 < g1) load(nxt, g + 1, m);
#pragma unroll
            for (uint32_t s4 = 0; s4 < 16; s4 += 2) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[s4], cur.b[s4], d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.a[s4 + 1], cur.b[s4 + 1], d1, 0, 0, 0);
            }
            cur = nxt;
        }
    }
    // D[row 4 (l / 16) + r][vector l % 16] sits in register r of lane l.  The four units of the workgroup belong to one row tile: add
    // them up here (wavefronts 1-3 through LDS, wavefront 0 writes) -- a quarter of the partial tiles for the finish pass to read.
    if (wave) {
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) tile_sum[wave - 1][(4u * k + r) * kVec + i] = d0[r] + d1[r];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partial + size_t(blockIdx.x) * (kMfmaTileRows * kVec);
#pragma unroll
        for (uint32_t r = 0; r < 4; ++r) {
            const uint32_t at = (4u * k + r) * kVec + i;
            out[at] = (d0[r] + d1[r]) + (tile_sum[0][at] + tile_sum[1][at]) + tile_sum[2][at];
        }
    }
}

// Y[vector j][row] = fp32( sum over the row tile's partial tiles, in double ) -- one thread per (row, vector).  When X held a non-finite
// word (*flag == call) the MFMA kernel has done nothing and this kernel computes the row the way the PEs would: it walks the row's set
// bits -- stored elements only, fp32 product, double sum -- so that inf / NaN reach exactly the rows that touch their column.  Slow, rare.
__global__ __launch_bounds__(256) void spmm_finish_kernel(const uint32_t* __restrict__ words, uint64_t offsets_word, uint64_t values_word, uint32_t groups,
                                                         uint32_t chunks, uint32_t num_rows, uint32_t num_cols, const float* __restrict__ partial,
                                                         const uint32_t* __restrict__ x, uint64_t ldx, const uint32_t* __restrict__ flag, uint32_t call,
                                                         uint32_t* __restrict__ y, uint64_t ldy, uint32_t vectors) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t row = t / kVec, j = t % kVec;
    if (row >= num_rows || j >= vectors) return;
    const uint32_t tile = row / kMfmaTileRows, i = row % kMfmaTileRows;
    double sum = 0.0;
    if (*flag != call) {
        const uint32_t per_tile = chunks / 4u;              // one partial tile per workgroup of four units
        const float* p = partial + (size_t(tile) * per_tile * kMfmaTileRows + i) * kVec + j;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // independent loads in flight
        uint32_t c = 0;
        for (; c + 4 <= per_tile; c += 4) {
            s0 += static_cast<double>(p[size_t(c) * 256u]);
            s1 += static_cast<double>(p[size_t(c + 1) * 256u]);
            s2 += static_cast<double>(p[size_t(c + 2) * 256u]);
            s3 += static_cast<double>(p[size_t(c + 3) * 256u]);
        }
        for (; c < per_tile; ++c) s0 += static_cast<double>(p[size_t(c) * 256u]);
        sum = (s0 + s1) + (s2 + s3);
    } else {
        // the tile's values are stored group by group in (step, lane) order: walk the 16 rows' masks, count everybody's set bits, use this row's
        const uint64_t* masks = reinterpret_cast<const uint64_t*>(words) + uint64_t(tile) * groups * kMfmaTileRows;
        const float* values = reinterpret_cast<const float*>(words + values_word);
        const uint32_t* xj = x + size_t(j) * ldx;
        uint32_t idx = words[offsets_word + uint64_t(tile) * chunks];      // the tile's first value (unit 0 of the tile)
        for (uint32_t g = 0; g < groups; ++g) {
            const uint64_t* m = masks + uint64_t(g) * kMfmaTileRows;
            for (uint32_t p = 0; p < 64; ++p) {                            // p = 4 s + k: step-major, then k, then row -- the storage order
                for (uint32_t r = 0; r < kMfmaTileRows; ++r) {
                    if (!((m[r] >> p) & 1ull)) continue;
                    if (r == i) {
                        const uint32_t col = g * 64u + p;
                        if (col < num_cols) sum += static_cast<double>(values[idx] * __uint_as_float(xj[col]));
                    }
                    ++idx;
                }
            }
        }
    }
    y[size_t(j) * ldy + row] = __float_as_uint(static_cast<float>(sum));
}

}  // namespace

size_t spmm_mfma_x_words(uint32_t groups) { return size_t(groups) * 64u * kVec; }
size_t spmm_mfma_partial_words(uint32_t tiles, uint32_t chunks) { return size_t(tiles) * (chunks / 4u) * kMfmaTileRows * kVec; }

hipError_t launch_spmm_mfma(const SpmmMfmaLaunch& a, hipStream_t stream) {
    const uint32_t padded_cols = a.groups * 64u, workgroups = a.tiles * a.chunks / 4u;
    hipLaunchKernelGGL(interleave_x16_kernel, dim3((padded_cols + 255) / 256), dim3(256), 0, stream, a.x, a.ldx, a.num_cols, padded_cols, a.vectors, a.x_interleaved,
                       a.flag, a.call);
    hipLaunchKernelGGL(spmm_mfma_kernel, dim3(workgroups), dim3(kUnitThreads), 0, stream, a.words, a.offsets_word, a.values_word, a.groups, a.chunk, a.chunks,
                       reinterpret_cast<const float*>(a.x_interleaved), a.flag, a.call, a.partial);
    hipLaunchKernelGGL(spmm_finish_kernel, dim3((a.num_rows * kVec + 255) / 256), dim3(256), 0, stream, a.words, a.offsets_word, a.values_word, a.groups,
                       a.chunks, a.num_rows, a.num_cols, a.partial, a.x, a.ldx, a.flag, a.call, a.y, a.ldy, a.vectors);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
