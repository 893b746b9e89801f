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
//   1. EXPAND: one wavefront per stored x entry claims room in the product list with ONE atomic on a device counter (its column's
//      length), streams the column (row index + value word: two coalesced loads per 64 non-zeros), multiplies with the PE arithmetic of
//      the numeric mode and writes, per product: the row (u32), the product word (u32) and the row BLOCK it falls in (u16, 8192 rows);
//   2. ACCUMULATE: one workgroup per row block sweeps the 2-byte block ids of the whole list (16 bytes = 8 products per lane and load:
//      the list is L2-resident and the ids are all a workgroup reads of the products that are not its own), fetches row and product
//      of the matches, adds them into 64-bit LDS accumulators (ds_add_u64 / ds_add_f64: exact integer sums, double sums of the fp32
//      products) and writes ITS rows of y -- every row exactly once, so y needs no zeroing pass and no finish pass.
// The order of the list is whatever the atomics make it; the sums do not care (fixed point: exact; float: tolerance, as everywhere).
// No memory-side atomic per product (they run at ~24 G/s on this chip): one per selected COLUMN.
// Every workgroup of the second launch reads 2 bytes per product, so the cost grows with (row blocks x products): the operator pays
// below a few per cent of the columns; above the measured crossover the caller's dense SpMV is faster and hs_spmspv dispatches to it
// when it can (hs_api.cpp; hisparse_hip.h says so).
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

constexpr uint32_t kBlockBits = 13;                       // a row block = 8192 rows: 64 KiB of 8-byte LDS accumulators
constexpr uint32_t kBlockRows = 1u << kBlockBits;
constexpr uint32_t kIdsPerLoad = 8;                       // block ids a lane reads at once (16 bytes)

template <bool kFloat>
__device__ __forceinline__ uint32_t product_word(uint32_t value_word, uint32_t x_word) {
    if (kFloat) return __float_as_uint(__uint_as_float(value_word) * __uint_as_float(x_word));
    return q8_24_mul(value_word, x_word);
}

// counters: [0], [1] = product counts of the calls with even / odd call number (the other one is reset by this call's accumulate
// kernel, so no memset sits between two calls), [2] = overflow flag (sticky until hs_load_matrix_csc)
template <bool kFloat>
__global__ __launch_bounds__(256) void spmspv_expand_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ row_indices,
                                                           const uint32_t* __restrict__ value_words, const uint2* __restrict__ x_entries, uint32_t x_count,
                                                           uint32_t num_cols, unsigned long long* __restrict__ counter, unsigned long long* __restrict__ overflow, uint32_t capacity,
                                                           uint32_t* __restrict__ keys, uint32_t* __restrict__ vals, uint16_t* __restrict__ blks) {
    const uint32_t lane = threadIdx.x & (kWaveLanes - 1);
    const uint32_t wave = blockIdx.x * (blockDim.x / kWaveLanes) + threadIdx.x / kWaveLanes;
    const uint32_t waves = gridDim.x * (blockDim.x / kWaveLanes);
    for (uint32_t k = wave; k < x_count; k += waves) {
        const uint2 entry = x_entries[k];                  // IDX_VAL_T { index, val }
        const uint32_t col = entry.x, xw = entry.y;
        if (col >= num_cols) continue;                     // (checked on the host where the host holds the entries)
        const uint32_t lo = indptr[col], hi = indptr[col + 1];
        if (hi == lo) continue;
        unsigned long long claimed = 0;
        if (lane == 0) claimed = atomicAdd(counter, static_cast<unsigned long long>(hi - lo));
        const uint32_t at_hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(claimed >> 32));
        const uint32_t at = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(claimed));
        if (at_hi != 0 || at > capacity || hi - lo > capacity - at) {    // the list is full (an x with repeated entries can ask for more than nnz products)
            if (lane == 0) atomicOr(overflow, 1ull);
            continue;
        }
        for (uint32_t e = lo + lane; e < hi; e += kWaveLanes) {
            const uint32_t row = row_indices[e];
            keys[at + (e - lo)] = row;
            vals[at + (e - lo)] = product_word<kFloat>(value_words[e], xw);
            blks[at + (e - lo)] = static_cast<uint16_t>(row >> kBlockBits);
        }
    }
}

// One workgroup per row block.  kAdd: y += (a second pass of a call whose products did not fit the list at once; saturating / fp32 add).
// The sweep over the block ids and the fetch of the matching products are DECOUPLED through a queue in LDS: a trip of the loop looks at
// 8192 products (one 16-byte load of ids per lane, the next trip's already in flight) and pushes the indices of its own into the queue;
// the queue is drained -- row and product word of every entry loaded four at a time, one LDS add each -- only when the next trip might
// overflow it, and at the end.  (The first version fetched a match where it found it: with 1 product in 71 matching, nearly every one of
// a lane's 8 ids had SOME lane of the wavefront taking the branch, i.e. 8 dependent load round trips per trip: 179 us for 420 K
// products, profiles/r04_spmspv.txt.)
constexpr uint32_t kTripProducts = 1024 * kIdsPerLoad;     // 8192
constexpr uint32_t kQueueEntries = 2 * kTripProducts;      // 64 KiB: drained when fewer than one trip's worth of room is left
template <bool kFloat, bool kAdd>
__global__ __launch_bounds__(1024) void spmspv_accumulate_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                const uint16_t* __restrict__ blks, const unsigned long long* __restrict__ counter,
                                                                unsigned long long* __restrict__ counter_next, uint32_t capacity, uint32_t* __restrict__ y,
                                                                uint32_t num_rows) {
    using R = Rows<kFloat>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typename R::acc_t* acc = reinterpret_cast<typename R::acc_t*>(lds);                      // [kBlockRows]
    uint32_t* queue = reinterpret_cast<uint32_t*>(lds + kBlockRows * sizeof(typename R::acc_t));   // [kQueueEntries]
    __shared__ uint32_t queued;
    const uint32_t tid = threadIdx.x, b = blockIdx.x, row0 = b << kBlockBits;
    const unsigned long long claimed = ((const __attribute__((address_space(4))) unsigned long long*)counter)[0];
    const uint32_t total = claimed > capacity ? capacity : static_cast<uint32_t>(claimed);      // (beyond the capacity: the overflow flag is up and the call reports it)
    const uint4* ids4 = reinterpret_cast<const uint4*>(blks);
    const uint4 none = make_uint4(0, 0, 0, 0);
    uint4 w = tid * kIdsPerLoad < total ? ids4[tid] : none;      // (the allocation is padded to whole 16-byte words)
    for (uint32_t i = tid; i < kBlockRows; i += 1024) acc[i] = 0;
    if (tid == 0) queued = 0;
    if (b == 0 && tid == 0) *counter_next = 0;             // the next call's counter (nobody reads it before that call's expand kernel)
    __syncthreads();
    auto drain = [&](uint32_t n) {
        for (uint32_t q = tid; q < n; q += 4 * 1024) {
            uint32_t idx[4], key[4], val[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) idx[k] = queue[min(q + k * 1024u, n - 1)];
#pragma unroll
            for (int k = 0; k < 4; ++k) { key[k] = keys[idx[k]]; val[k] = vals[idx[k]]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (q + k * 1024u >= n) break;
                const uint32_t local = key[k] & (kBlockRows - 1u);
                if (kFloat) R::add(acc, local, __uint_as_float(val[k]));                                // ds_add_f64 of the fp32 product
                else atomicAdd(acc + local, static_cast<unsigned long long>(val[k]));                   // ds_add_u64
            }
        }
    };
    for (uint32_t base = 0; base < total; base += kTripProducts) {
        const uint32_t i = base + tid * kIdsPerLoad;
        const uint4 ahead = i + kTripProducts < total ? ids4[(i + kTripProducts) / kIdsPerLoad] : none;
        const uint32_t word[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (uint32_t j = 0; j < kIdsPerLoad; ++j) {
            const uint32_t id = (word[j / 2] >> (16 * (j % 2))) & 0xffffu;
            if (id == b && i + j < total) queue[atomicAdd(&queued, 1u)] = i + j;      // (ids past `total` are leftovers of earlier calls)
        }
        __syncthreads();
        const uint32_t n = queued;
        __syncthreads();                                   // everybody has read the count before the next trip's pushes move it
        if (n > kQueueEntries - kTripProducts || base + kTripProducts >= total) {
            drain(n);
            __syncthreads();
            if (tid == 0) queued = 0;
            __syncthreads();
        }
        w = ahead;
    }
    // no-return LDS atomics can outlive s_waitcnt lgkmcnt(0) (spmv_kernels.hip): a RETURNING atomic per wavefront, awaited
    const typename R::acc_t flushed = atomicAdd(acc + (tid / kWaveLanes), static_cast<typename R::acc_t>(0));
    asm volatile("" ::"v"(flushed));
    __syncthreads();
    for (uint32_t i = tid; i < kBlockRows && row0 + i < num_rows; i += 1024) {
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

size_t spmspv_list_bytes(uint64_t capacity) { return ((size_t(capacity) * 2 + 15) & ~size_t(15)) + 16; }      // the block-id array, padded to whole loads

hipError_t launch_spmspv(bool is_float, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, const hs_idx_val_dev* x_entries,
                         uint32_t x_count, uint32_t num_rows, uint32_t num_cols, const SpmspvScratch& s, uint32_t call, bool add_to_y, uint32_t* y,
                         hipStream_t stream) {
    unsigned long long* counter = s.counters + (call & 1u);
    unsigned long long* counter_next = s.counters + ((call + 1u) & 1u);
    const uint32_t capacity = static_cast<uint32_t>(std::min<uint64_t>(s.capacity, 0xffffffffull));
    const dim3 block(256);
    if (x_count) {
        const dim3 grid(std::min<uint32_t>((x_count + 3) / 4, 8192));       // 4 wavefronts per workgroup, one x entry each
        const uint2* xe = reinterpret_cast<const uint2*>(x_entries);
        if (is_float)
            hipLaunchKernelGGL(spmspv_expand_kernel<true>, grid, block, 0, stream, indptr, row_indices, value_words, xe, x_count, num_cols, counter, s.counters + 2,
                               capacity, s.keys, s.vals, s.blks);
        else
            hipLaunchKernelGGL(spmspv_expand_kernel<false>, grid, block, 0, stream, indptr, row_indices, value_words, xe, x_count, num_cols, counter, s.counters + 2,
                               capacity, s.keys, s.vals, s.blks);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const dim3 grid((num_rows + kBlockRows - 1) / kBlockRows);
    const uint32_t lds = kBlockRows * 8u + kQueueEntries * 4u;      // accumulators + queue: 128 KiB
#define X(F, A)                                                                                                                                                  \
    do {                                                                                                                                                         \
        static bool configured_on[64] = {};      /* the dynamic-LDS cap is a property of the function, per device */                                            \
        int dev_ = 0;                                                                                                                                            \
        (void)hipGetDevice(&dev_);                                                                                                                               \
        bool& configured = configured_on[dev_ >= 0 && dev_ < 64 ? dev_ : 0];                                                                                     \
        if (!configured) {                                                                                                                                       \
            const hipError_t ce = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmspv_accumulate_kernel<F, A>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)); \
            if (ce != hipSuccess) return ce;                                                                                                                     \
            configured = true;                                                                                                                                   \
        }                                                                                                                                                        \
        hipLaunchKernelGGL((spmspv_accumulate_kernel<F, A>), grid, dim3(1024), lds, stream, s.keys, s.vals, s.blks, counter, counter_next, capacity, y, num_rows); \
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
