// spmspv.hip — SpMSpV on gfx950: y = A x for a SPARSE x over a CSC matrix.  EXTENSION.
//
// The reference only stubs this operator -- the packet / pair types SPMSPV_MAT_PKT_T and IDX_VAL_T
// (spmv/libfpga/common.h:52-54) and the CSC conversion csr2csc (sw/data_loader.h:109-144); the paper (section 7) names it as the
// natural next kernel on the same datapath.  The work is proportional to the non-zeros of the SELECTED columns only, which is the
// point of the operator.
//
// Round 4: two launches, nothing else -- no scan, no sort, no host synchronisation, no H2D inside the call (round 3's path took a
// hipCUB scan, a stream sync to learn the product count, a radix sort and an H2D before its first product: 53-205 us where the dense
// SpMV takes 56):
//   1. EXPAND: a workgroup takes up to 64 entries of x, lines their columns' products up (DPP prefix sum of the column lengths) and walks
//      them flat, twice: once to count its products per ROW BLOCK (LDS histogram), then -- after ONE global atomic per (workgroup, non-empty
//      row block) has claimed a stretch of that block's BIN -- to compute the products with the PE arithmetic of the numeric mode and place
//      (row, product word) in the bin;
//   2. ACCUMULATE: one workgroup per row block reads ITS bin and nothing else, adds the products into 64-bit LDS accumulators (ds_add_u64
//      / ds_add_f64: exact integer sums, double sums of the fp32 products), writes ITS rows of y -- every row exactly once, so y needs no
//      zeroing pass and no finish pass -- and re-arms its bin's cursor for the next call.
// A bin's capacity is the number of non-zeros the matrix holds in that row block (a histogram taken when the CSC image is loaded): entries
// that name every column at most once can never overflow it, so the bins together are one list of nnz entries.  The order inside a bin is
// whatever the atomics make it; the sums do not care (fixed point: exact; float: tolerance, as everywhere).  No memory-side atomic per
// product (they run at ~24 G/s on this chip): one per (workgroup of 64 columns, row block touched).
// Cost: ~12-18 us (two launches and two dependent chains: entry -> column pointer -> elements; cursor -> bin) + products / ~40 G/s (a column
// entry is read, its product written and read again: 24 bytes per product at ~1 TB/s, mostly L2); above the crossover with the caller's
// dense SpMV hs_spmspv dispatches to that when it can (hs_api.cpp; hisparse_hip.h says so).
//   fixed: products rounded / saturated one by one (q8_24_mul), summed exactly in 64 bits, clamped once -- bit-identical to the
//          saturating PE sum, in any order;
//   float: one fp32 multiply per product, double sums per row block, rounded once: tolerance parity like every float path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "spmv_device.h"
#include "spmv_kernels.h"

namespace hisparse {
namespace dev {

namespace {

constexpr uint32_t kMaxBlockBits = 13;                    // a row block = 8192 rows at most (64 KiB of 8-byte LDS accumulators) ...
constexpr uint32_t kHugeBlockBits = 14;                   // ... 16 384 (128 KiB) once the matrix has more than 16.7 M rows
constexpr uint32_t kSmallBins = 2048;                     // bins an expand workgroup counts in 16 KiB of LDS; beyond: up to kMaxBins in 128 KiB
constexpr uint32_t kMinBlockBits = 9;
constexpr uint32_t kExpandColumns = 64;                   // x entries per workgroup of the expand kernel
constexpr uint32_t kExpandThreads = 256;

template <bool kFloat>
__device__ __forceinline__ uint32_t product_word(uint32_t value_word, uint32_t x_word) {
    if (kFloat) return __float_as_uint(__uint_as_float(value_word) * __uint_as_float(x_word));
    return q8_24_mul(value_word, x_word);
}

// EXPAND.  A workgroup takes up to 64 entries of x: lane c of its first wavefront reads entry c's column pointer, a DPP prefix sum lines the
// columns' products up, and all 256 threads walk them FLAT -- product p belongs to the column whose prefix range holds p (binary search over
// the prefix sums in LDS) -- so that a hub column of 60 K entries is spread over the workgroup like everything else.  TWICE: the first walk
// counts the workgroup's products per ROW BLOCK in an LDS histogram; then one global atomic per (workgroup, non-empty block) claims a stretch
// of that block's BIN; the second walk computes the products and writes each (row, product word) into its bin, at the claimed base + its
// rank (a returning LDS atomic).  A bin's capacity is the number of non-zeros the matrix has in that row block -- x entries that name every
// column at most once can never ask for more -- so the bins together are one list of nnz entries, and the accumulate kernel reads ITS bin and
// nothing else.  (Round 4's first version wrote one unsorted list and had every row block's workgroup sweep the block ids of ALL products:
// ~5 G products/s whatever the matrix, 86 us for 1 % of ogbl-ppa's columns; profiles/r04_spmspv.txt.)
constexpr uint32_t kMaxBins = 16384;                      // LDS histogram + claims: 8 bytes per bin, 128 KiB at most -> 268 M rows
template <bool kFloat>
__global__ __launch_bounds__(kExpandThreads) void spmspv_expand_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ row_indices,
                                                                      const uint32_t* __restrict__ value_words, const uint2* __restrict__ x_entries,
                                                                      uint32_t x_count, uint32_t num_cols, const uint32_t* __restrict__ bin_base,
                                                                      uint32_t* __restrict__ cursors, uint32_t* __restrict__ overflow, uint32_t block_bits,
                                                                      uint32_t bins, uint32_t columns, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    // columns (<= 64): entries of x per workgroup -- 64 when x has thousands of entries, fewer when it has few, so that a handful of LONG
    // columns (mouse_gene: 22 columns of 640 non-zeros) is still spread over many workgroups
    __shared__ uint32_t s_lo[kExpandColumns], s_xw[kExpandColumns], s_pre[kExpandColumns + 1];
    extern __shared__ uint32_t s_bins[];                   // [bins] histogram, then running rank; [bins] claimed base (absolute index into keys / vals)
    uint32_t* hist = s_bins;
    uint32_t* claim = s_bins + bins;
    const uint32_t tid = threadIdx.x;
    for (uint32_t b = tid; b < bins; b += kExpandThreads) hist[b] = 0;
    if (tid < kExpandColumns) {
        const uint32_t k = blockIdx.x * columns + tid;
        uint32_t lo = 0, len = 0, xw = 0;
        if (tid < columns && k < x_count) {
            const uint2 entry = x_entries[k];              // IDX_VAL_T { index, val }
            if (entry.x < num_cols) {                      // (checked on the host where the host holds the entries)
                lo = indptr[entry.x];
                len = indptr[entry.x + 1] - lo;
                xw = entry.y;
            }
        }
        const uint32_t incl = wave_inclusive_scan(len);    // (64 columns stay below 2^32 products: the matrix has fewer non-zeros than that)
        s_lo[tid] = lo;
        s_xw[tid] = xw;
        s_pre[tid + 1] = incl;
        if (tid == 0) s_pre[0] = 0;
    }
    __syncthreads();
    const uint32_t total = s_pre[kExpandColumns];
    auto column_of = [&](uint32_t p) {                     // the last column whose prefix is <= p
        uint32_t c = 0;
#pragma unroll
        for (uint32_t step = kExpandColumns / 2; step; step >>= 1)
            if (s_pre[c + step] <= p) c += step;
        return c;
    };
    for (uint32_t p = tid; p < total; p += kExpandThreads) {
        const uint32_t c = column_of(p);
        atomicAdd(&hist[row_indices[s_lo[c] + (p - s_pre[c])] >> block_bits], 1u);
    }
    __syncthreads();
    for (uint32_t b = tid; b < bins; b += kExpandThreads) {
        const uint32_t n = hist[b];
        uint32_t at = 0xffffffffu;                         // "do not write": the bin is full
        if (n) {
            const uint32_t first = atomicAdd(&cursors[b], n), room = bin_base[b + 1] - bin_base[b];
            if (first <= room && n <= room - first) at = bin_base[b] + first;
            else atomicOr(overflow, 1u);                   // (an x that names a column more than once can ask for more than the bin holds)
        }
        claim[b] = at;
        hist[b] = 0;                                       // from here on: the rank of the workgroup's next product in this bin
    }
    __syncthreads();
    for (uint32_t p = tid; p < total; p += kExpandThreads) {
        const uint32_t c = column_of(p), e = s_lo[c] + (p - s_pre[c]);
        const uint32_t row = row_indices[e], b = row >> block_bits;
        const uint32_t base = claim[b], rank = atomicAdd(&hist[b], 1u);
        if (base != 0xffffffffu) {
            keys[base + rank] = row;
            vals[base + rank] = product_word<kFloat>(value_words[e], s_xw[c]);
        }
    }
}

// ACCUMULATE.  One workgroup per row block of 2^block_bits rows: its bin holds its products and nothing else -- cursors[b] of them, behind
// bin_base[b] -- read four per lane at a time, added into 64-bit LDS accumulators (ds_add_u64 / ds_add_f64); every row of y is written
// exactly once, so y needs no zeroing pass; the workgroup re-arms its own cursor for the next call.  kAdd: y += (a later pass of a call
// whose x names columns more than once; saturating / fp32 add).
template <bool kFloat, bool kAdd>
__global__ __launch_bounds__(1024) void spmspv_accumulate_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                const uint32_t* __restrict__ bin_base, uint32_t* __restrict__ cursors, uint32_t block_bits,
                                                                uint32_t* __restrict__ y, uint32_t num_rows) {
    using R = Rows<kFloat>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typename R::acc_t* acc = reinterpret_cast<typename R::acc_t*>(lds);                              // [block rows]
    const uint32_t block_rows = 1u << block_bits;
    const uint32_t tid = threadIdx.x, b = blockIdx.x, row0 = b << block_bits;
    const uint32_t first = ((const __attribute__((address_space(4))) uint32_t*)bin_base)[b];
    const uint32_t room = ((const __attribute__((address_space(4))) uint32_t*)bin_base)[b + 1] - first;
    // The cursor is read with an ORDINARY load by one thread and handed round through LDS: it is rewritten below, so it must not be read
    // through the constant address space (an invariant load the compiler may move across the barrier and the store).  A bin that was asked
    // for more than it holds (hs_spmspv_device with a column named twice: the expand kernel dropped whole claims and raised the overflow
    // word) has slots below `room` that nobody wrote: such a bin contributes NOTHING -- zeros, never an earlier call's products.
    __shared__ uint32_t s_count;
    if (tid == 0) {
        const uint32_t claimed = *reinterpret_cast<volatile uint32_t*>(cursors + b);
        s_count = claimed <= room ? claimed : 0u;
        cursors[b] = 0;                                    // re-armed for the next call
    }
    for (uint32_t i = tid; i < block_rows; i += 1024) acc[i] = 0;
    __syncthreads();
    const uint32_t n = s_count;
    for (uint32_t q = tid; q < n; q += 4 * 1024) {
        uint32_t key[4], val[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = first + min(q + k * 1024u, n - 1);
            key[k] = keys[i];
            val[k] = vals[i];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (q + k * 1024u >= n) break;
            const uint32_t local = key[k] & (block_rows - 1u);
            if (kFloat) R::add(acc, local, __uint_as_float(val[k]));                                // ds_add_f64 of the fp32 product
            else atomicAdd(acc + local, static_cast<unsigned long long>(val[k]));                   // ds_add_u64
        }
    }
    // no-return LDS atomics can outlive s_waitcnt lgkmcnt(0) (spmv_kernels.hip): a RETURNING atomic per wavefront, awaited
    const typename R::acc_t flushed = atomicAdd(acc + (tid / kWaveLanes), static_cast<typename R::acc_t>(0));
    asm volatile("" ::"v"(flushed));
    __syncthreads();
    for (uint32_t i = tid; i < block_rows && row0 + i < num_rows; i += 1024) {
        uint32_t word = R::finish(acc[i]);
        if (kAdd) {
            const uint32_t old = y[row0 + i];
            if (kFloat) word = __float_as_uint(__uint_as_float(old) + __uint_as_float(word));
            else word = __builtin_elementwise_add_sat(old, word);      // min(a + b, 2^32-1): saturating sums compose (spmv_kernels.hip: combine)
        }
        y[row0 + i] = word;
    }
}

// x scattered into a dense zero vector (the dense dispatch of hs_spmspv: unique indices only, checked by the caller)
__global__ __launch_bounds__(256) void spmspv_scatter_x_kernel(const uint2* __restrict__ x_entries, uint32_t x_count, uint32_t num_cols, uint32_t* __restrict__ x_dense) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= x_count) return;
    const uint2 e = x_entries[k];
    if (e.x < num_cols) x_dense[e.x] = e.y;
}

}  // namespace


// rows per block: 8192 unless the matrix has too few rows to give every other CU a block that way (mouse_gene: 45 K rows -> 512-row blocks)
uint32_t spmspv_block_bits(uint32_t num_rows) {
    uint32_t bits = kMaxBlockBits;
    while (bits > kMinBlockBits && ((uint64_t(num_rows) + (1u << bits) - 1) >> bits) < 128) --bits;
    if (((uint64_t(num_rows) + (1u << bits) - 1) >> bits) > kSmallBins) bits = kHugeBlockBits;      // > 16.7 M rows
    return bits;
}
uint32_t spmspv_bins(uint32_t num_rows) { return uint32_t((uint64_t(num_rows) + (1u << spmspv_block_bits(num_rows)) - 1) >> spmspv_block_bits(num_rows)); }
uint32_t spmspv_max_bins() { return kMaxBins; }

hipError_t launch_spmspv(bool is_float, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, const hs_idx_val_dev* x_entries,
                         uint32_t x_count, uint32_t num_rows, uint32_t num_cols, const SpmspvScratch& s, bool add_to_y, uint32_t* y, hipStream_t stream) {
    const uint32_t block_bits = spmspv_block_bits(num_rows), bins = spmspv_bins(num_rows);
    if (bins > kMaxBins) return hipErrorInvalidValue;
    if (x_count) {
        // Entries of x per workgroup: ~1500 products each (a workgroup claims room with one global atomic per row block it touches -- with a
        // column or two per workgroup that is nearly an atomic per PRODUCT on 71 addresses: ogbl-ppa 0.1 % 18.3 us -- and walks its products
        // twice, 256 threads wide -- with 64 long columns per workgroup that walk is the critical path: mouse_gene 1 % 34-87 us), but at
        // least ~128 workgroups while x has the entries for it.  Measured over 16 ... 512 workgroups-before-growth on three matrices
        // (profiles/r04_spmspv_expand_columns.txt): ogbl-ppa 0.1 % 18.3 -> 9.9 us, 1 % 22.7 -> 19.3, pokec 0.1 % 17.6 -> 11.0, 1 % 21.9 -> 17.8.
        const double avg_len = num_cols ? double(s.capacity) / double(num_cols) : 1.0;
        uint32_t columns = uint32_t(std::max(1.0, std::min<double>(kExpandColumns, 1500.0 / std::max(1.0, avg_len))));
        columns = std::min<uint32_t>(kExpandColumns, std::max(columns, x_count / 512));      // (beyond ~512 workgroups the claims add up again: ogbl-ppa 5 %, 57 columns 57 us, 20 columns 70)
        columns = std::min(columns, std::max<uint32_t>(1, x_count / 128));
        const dim3 grid((x_count + columns - 1) / columns), block(kExpandThreads);
        const uint2* xe = reinterpret_cast<const uint2*>(x_entries);
        const uint32_t lds = bins * 8u;
        if (lds > 48u * 1024u) {                 // > 6144 bins (100 M rows): the histogram needs the function's dynamic-LDS cap raised
            static bool raised_on[64] = {};
            int dev_ = 0;
            (void)hipGetDevice(&dev_);
            bool& raised = raised_on[dev_ >= 0 && dev_ < 64 ? dev_ : 0];
            if (!raised) {
                hipError_t ce = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmspv_expand_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(kMaxBins * 8u));
                if (ce == hipSuccess)
                    ce = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmspv_expand_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(kMaxBins * 8u));
                if (ce != hipSuccess) return ce;
                raised = true;
            }
        }
        if (is_float)
            hipLaunchKernelGGL(spmspv_expand_kernel<true>, grid, block, lds, stream, indptr, row_indices, value_words, xe, x_count, num_cols, s.bin_base, s.cursors,
                               s.overflow, block_bits, bins, columns, s.keys, s.vals);
        else
            hipLaunchKernelGGL(spmspv_expand_kernel<false>, grid, block, lds, stream, indptr, row_indices, value_words, xe, x_count, num_cols, s.bin_base, s.cursors,
                               s.overflow, block_bits, bins, columns, s.keys, s.vals);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const dim3 grid(bins);
    const uint32_t lds = 8u << block_bits;      // accumulators: 64 KiB (128 KiB above 16.7 M rows)
#define X(F, A)                                                                                                                                                  \
    do {                                                                                                                                                         \
        static bool configured_on[64] = {};      /* the dynamic-LDS cap is a property of the function, per device */                                            \
        int dev_ = 0;                                                                                                                                            \
        (void)hipGetDevice(&dev_);                                                                                                                               \
        bool& configured = configured_on[dev_ >= 0 && dev_ < 64 ? dev_ : 0];                                                                                     \
        if (!configured) {                                                                                                                                       \
            const hipError_t ce = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmspv_accumulate_kernel<F, A>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      int(8u << kHugeBlockBits));                                                                                 \
            if (ce != hipSuccess) return ce;                                                                                                                     \
            configured = true;                                                                                                                                   \
        }                                                                                                                                                        \
        hipLaunchKernelGGL((spmspv_accumulate_kernel<F, A>), grid, dim3(1024), lds, stream, s.keys, s.vals, s.bin_base, s.cursors, block_bits, y, num_rows);      \
    } while (0)
    if (is_float) { if (add_to_y) X(true, true); else X(true, false); }
    else { if (add_to_y) X(false, true); else X(false, false); }
#undef X
    return hipGetLastError();
}

hipError_t launch_spmspv_scatter_x(const hs_idx_val_dev* x_entries, uint32_t x_count, uint32_t num_cols, uint32_t* x_dense, hipStream_t stream) {
    const hipError_t e = hipMemsetAsync(x_dense, 0, size_t(num_cols) * 4, stream);
    if (e != hipSuccess || x_count == 0) return e;
    hipLaunchKernelGGL(spmspv_scatter_x_kernel, dim3((x_count + 255) / 256), dim3(256), 0, stream, reinterpret_cast<const uint2*>(x_entries), x_count, num_cols, x_dense);
    return hipGetLastError();
}

}  // namespace dev
}  // namespace hisparse
