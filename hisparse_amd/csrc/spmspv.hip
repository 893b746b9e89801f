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

constexpr uint32_t kMaxBlockBits = 13;                    // a row block = at most 8192 rows: 64 KiB of 8-byte LDS accumulators
constexpr uint32_t kMinBlockBits = 9;
constexpr uint32_t kIdsPerLoad = 8;                       // block ids a lane reads at once (16 bytes)
constexpr uint32_t kExpandColumns = 64;                   // x entries per workgroup of the expand kernel
constexpr uint32_t kExpandThreads = 256;

template <bool kFloat>
__device__ __forceinline__ uint32_t product_word(uint32_t value_word, uint32_t x_word) {
    if (kFloat) return __float_as_uint(__uint_as_float(value_word) * __uint_as_float(x_word));
    return q8_24_mul(value_word, x_word);
}

// counters: [0], [1] = product counts of the calls with even / odd call number (the other one is reset by this call's accumulate
// kernel, so no memset sits between two calls), [2] = overflow flag (sticky until hs_read_spmspv_result reports it)
//
// EXPAND.  A workgroup takes 64 entries of x: lane c of its first wavefront reads entry c's column length, a DPP prefix sum places the 64
// columns' products behind each other, ONE atomic on the device counter claims the room for all of them (an atomic per column was the
// first version: 5762 atomics on one address took 115 of the 142 us of a 1 % selection of ogbl-ppa), and then all 256 threads walk the
// workgroup's products FLAT -- product p belongs to the column whose prefix range holds p (binary search over the 65 prefix sums in LDS) --
// so that a hub column of 60 K entries is spread over the workgroup like everything else and every store is coalesced.
template <bool kFloat>
__global__ __launch_bounds__(kExpandThreads) void spmspv_expand_kernel(const uint32_t* __restrict__ indptr, const uint32_t* __restrict__ row_indices,
                                                                      const uint32_t* __restrict__ value_words, const uint2* __restrict__ x_entries,
                                                                      uint32_t x_count, uint32_t num_cols, unsigned long long* __restrict__ counter,
                                                                      unsigned long long* __restrict__ overflow, uint32_t capacity, uint32_t block_bits,
                                                                      uint32_t columns, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                                      uint16_t* __restrict__ blks) {
    // columns (<= 64): entries of x per workgroup -- 64 when x has thousands of entries, fewer when it has few, so that a handful of LONG
    // columns (mouse_gene: 22 columns of 640 non-zeros) is still spread over many workgroups
    __shared__ uint32_t s_lo[kExpandColumns], s_xw[kExpandColumns], s_pre[kExpandColumns + 1], s_base, s_ok;
    const uint32_t tid = threadIdx.x, lane = tid & (kWaveLanes - 1);
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
        const uint32_t incl = wave_inclusive_scan(len);    // (a workgroup's 64 columns stay below 2^32 products: capacity < 2^32)
        s_lo[tid] = lo;
        s_xw[tid] = xw;
        s_pre[tid + 1] = incl;
        if (tid == 0) s_pre[0] = 0;
        if (lane == kWaveLanes - 1) {
            uint32_t ok = 1, base = 0;
            if (incl) {
                const unsigned long long claimed = atomicAdd(counter, static_cast<unsigned long long>(incl));
                if (claimed > capacity || incl > capacity - claimed) {      // the list is full (an x with repeated entries can ask for more than nnz products)
                    atomicOr(overflow, 1ull);
                    ok = 0;
                }
                base = static_cast<uint32_t>(claimed);
            }
            s_base = base;
            s_ok = ok;
        }
    }
    __syncthreads();
    const uint32_t total = s_pre[kExpandColumns], base = s_base;
    if (!s_ok) return;
    for (uint32_t p = tid; p < total; p += kExpandThreads) {
        uint32_t c = 0;                                    // the last column whose prefix is <= p
#pragma unroll
        for (uint32_t step = kExpandColumns / 2; step; step >>= 1)
            if (s_pre[c + step] <= p) c += step;
        const uint32_t e = s_lo[c] + (p - s_pre[c]);
        const uint32_t row = row_indices[e];
        keys[base + p] = row;
        vals[base + p] = product_word<kFloat>(value_words[e], s_xw[c]);
        blks[base + p] = static_cast<uint16_t>(row >> block_bits);
    }
}

// ACCUMULATE.  One workgroup per row block of 2^block_bits rows (8192 at most; fewer for matrices of few rows, so that there are
// workgroups enough).  kAdd: y += (a later pass of a call whose products did not fit the list at once; saturating / fp32 add).
// The sweep over the block ids and the fetch of the matching products are DECOUPLED through a queue in LDS: a trip looks at 32 K products
// (four 16-byte loads of ids per lane, the next trip's already in flight), the lanes that found one of their own push its index -- one LDS
// atomic per wavefront and id position, the lanes' places from the ballot -- and the queue is drained after the trip: row and product word
// of every entry loaded four at a time, one LDS add each.  (Fetching a match where it was found cost 8 dependent load round trips per
// 8 ids; a trip per 8 K products with a barrier pair each left the id loads' latency exposed: 179 / 142 us for 420 K products,
// profiles/r04_spmspv.txt.)  An entry that does not fit the queue any more is added on the spot.
constexpr uint32_t kTripLoads = 4;
constexpr uint32_t kTripProducts = 1024 * kIdsPerLoad * kTripLoads;      // 32768
constexpr uint32_t kQueueEntries = 8192;                                 // 32 KiB
template <bool kFloat, bool kAdd>
__global__ __launch_bounds__(1024) void spmspv_accumulate_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                                const uint16_t* __restrict__ blks, const unsigned long long* __restrict__ counter,
                                                                unsigned long long* __restrict__ counter_next, uint32_t capacity, uint32_t block_bits,
                                                                uint32_t* __restrict__ y, uint32_t num_rows) {
    using R = Rows<kFloat>;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typename R::acc_t* acc = reinterpret_cast<typename R::acc_t*>(lds);                              // [block rows]
    const uint32_t block_rows = 1u << block_bits;
    uint32_t* queue = reinterpret_cast<uint32_t*>(lds + block_rows * sizeof(typename R::acc_t));     // [kQueueEntries]
    __shared__ uint32_t queued;
    const uint32_t tid = threadIdx.x, lane = tid & (kWaveLanes - 1), b = blockIdx.x, row0 = b << block_bits;
    const unsigned long long claimed = ((const __attribute__((address_space(4))) unsigned long long*)counter)[0];
    const uint32_t total = claimed > capacity ? capacity : static_cast<uint32_t>(claimed);      // (beyond the capacity: the overflow flag is up and the call reports it)
    const uint4* ids4 = reinterpret_cast<const uint4*>(blks);
    const uint4 none = make_uint4(0, 0, 0, 0);
    auto load_trip = [&](uint4 (&w)[kTripLoads], uint32_t base) {
#pragma unroll
        for (uint32_t q = 0; q < kTripLoads; ++q) {
            const uint32_t i = base + (q * 1024 + tid) * kIdsPerLoad;
            w[q] = i < total ? ids4[i / kIdsPerLoad] : none;      // (the allocation is padded to whole 16-byte words)
        }
    };
    uint4 w[kTripLoads];
    load_trip(w, 0);
    for (uint32_t i = tid; i < block_rows; i += 1024) acc[i] = 0;
    if (tid == 0) queued = 0;
    if (b == 0 && tid == 0) *counter_next = 0;             // the next call's counter (nobody reads it before that call's expand kernel)
    __syncthreads();
    auto add_one = [&](uint32_t idx) {
        const uint32_t local = keys[idx] & (block_rows - 1u), v = vals[idx];
        if (kFloat) R::add(acc, local, __uint_as_float(v));
        else atomicAdd(acc + local, static_cast<unsigned long long>(v));
    };
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
                const uint32_t local = key[k] & (block_rows - 1u);
                if (kFloat) R::add(acc, local, __uint_as_float(val[k]));                                // ds_add_f64 of the fp32 product
                else atomicAdd(acc + local, static_cast<unsigned long long>(val[k]));                   // ds_add_u64
            }
        }
    };
    for (uint32_t base = 0; base < total; base += kTripProducts) {
        uint4 ahead[kTripLoads];
        load_trip(ahead, base + kTripProducts);
        // which of this lane's 32 ids are the block's own (bit q * 8 + j), how many, and where they go: ONE LDS atomic per wavefront and
        // trip claims the wavefront's stretch of the queue (a DPP prefix sum of the lanes' counts gives every lane its place in it).
        // (One returning LDS atomic per id position that had a match anywhere in the wavefront: 8 us per trip, 106 us for 420 K products.)
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t q = 0; q < kTripLoads; ++q) {
            const uint32_t i = base + (q * 1024 + tid) * kIdsPerLoad;
            const uint32_t word[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
#pragma unroll
            for (uint32_t j = 0; j < kIdsPerLoad; ++j) {
                const uint32_t id = (word[j / 2] >> (16 * (j % 2))) & 0xffffu;
                if (id == b && i + j < total) mine |= 1u << (q * kIdsPerLoad + j);      // (ids past `total` are leftovers of earlier calls)
            }
        }
        const uint32_t count = __builtin_popcount(mine);
        const uint32_t upto = wave_inclusive_scan(count);
        uint32_t at = 0;
        if (lane == kWaveLanes - 1 && upto) at = atomicAdd(&queued, upto);
        at = __builtin_amdgcn_readlane(at, kWaveLanes - 1) + upto - count;
        while (mine) {                                           // per lane: a handful of iterations at most
            const uint32_t bit = __builtin_ctz(mine);
            mine &= mine - 1;
            const uint32_t idx = base + ((bit / kIdsPerLoad) * 1024 + tid) * kIdsPerLoad + bit % kIdsPerLoad;
            if (at < kQueueEntries) queue[at] = idx;
            else add_one(idx);                                   // the queue is full (a row block that takes most of the products): add it here
            ++at;
        }
        __syncthreads();
        const uint32_t n = min(queued, kQueueEntries);
        __syncthreads();                                         // everybody has read the count
        if (n > kQueueEntries / 2 || base + kTripProducts >= total) {      // drain when the next trip might not fit, and at the end
            drain(n);
            __syncthreads();
            if (tid == 0) queued = 0;
            __syncthreads();
        }
#pragma unroll
        for (uint32_t q = 0; q < kTripLoads; ++q) w[q] = ahead[q];
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

size_t spmspv_list_bytes(uint64_t capacity) { return ((size_t(capacity) * 2 + 15) & ~size_t(15)) + 16; }      // the block-id array, padded to whole loads

// rows per block: 8192 unless the matrix has too few rows to give every other CU a block that way (mouse_gene: 45 K rows -> 512-row blocks)
uint32_t spmspv_block_bits(uint32_t num_rows) {
    uint32_t bits = kMaxBlockBits;
    while (bits > kMinBlockBits && ((uint64_t(num_rows) + (1u << bits) - 1) >> bits) < 128) --bits;
    return bits;
}

hipError_t launch_spmspv(bool is_float, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, const hs_idx_val_dev* x_entries,
                         uint32_t x_count, uint32_t num_rows, uint32_t num_cols, const SpmspvScratch& s, uint32_t call, bool add_to_y, uint32_t* y,
                         hipStream_t stream) {
    unsigned long long* counter = s.counters + (call & 1u);
    unsigned long long* counter_next = s.counters + ((call + 1u) & 1u);
    const uint32_t capacity = static_cast<uint32_t>(std::min<uint64_t>(s.capacity, 0xffffffffull));
    const uint32_t block_bits = spmspv_block_bits(num_rows);
    if (x_count) {
        const uint32_t columns = std::max<uint32_t>(1, std::min<uint32_t>(kExpandColumns, x_count / 512));      // >= 512 workgroups before they grow
        const dim3 grid((x_count + columns - 1) / columns), block(kExpandThreads);
        const uint2* xe = reinterpret_cast<const uint2*>(x_entries);
        if (is_float)
            hipLaunchKernelGGL(spmspv_expand_kernel<true>, grid, block, 0, stream, indptr, row_indices, value_words, xe, x_count, num_cols, counter, s.counters + 2,
                               capacity, block_bits, columns, s.keys, s.vals, s.blks);
        else
            hipLaunchKernelGGL(spmspv_expand_kernel<false>, grid, block, 0, stream, indptr, row_indices, value_words, xe, x_count, num_cols, counter, s.counters + 2,
                               capacity, block_bits, columns, s.keys, s.vals, s.blks);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const dim3 grid((num_rows + (1u << block_bits) - 1) >> block_bits);
    const uint32_t lds = (8u << block_bits) + kQueueEntries * 4u;      // accumulators + queue: 96 KiB at most
#define X(F, A)                                                                                                                                                  \
    do {                                                                                                                                                         \
        static bool configured_on[64] = {};      /* the dynamic-LDS cap is a property of the function, per device */                                            \
        int dev_ = 0;                                                                                                                                            \
        (void)hipGetDevice(&dev_);                                                                                                                               \
        bool& configured = configured_on[dev_ >= 0 && dev_ < 64 ? dev_ : 0];                                                                                     \
        if (!configured) {                                                                                                                                       \
            const hipError_t ce = hipFuncSetAttribute(reinterpret_cast<const void*>(&spmspv_accumulate_kernel<F, A>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                      int((8u << kMaxBlockBits) + kQueueEntries * 4u));                                                          \
            if (ce != hipSuccess) return ce;                                                                                                                     \
            configured = true;                                                                                                                                   \
        }                                                                                                                                                        \
        hipLaunchKernelGGL((spmspv_accumulate_kernel<F, A>), grid, dim3(1024), lds, stream, s.keys, s.vals, s.blks, counter, counter_next, capacity, block_bits, y, num_rows); \
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
