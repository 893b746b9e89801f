// tools/gather_bench.hip — round 4, VERDICT item 2: what does a row-owner block cost when x is NOT staged in LDS?
//
// The hyper-sparse path (OWNER24) pays 1.3-1.5 us per (row range x 8192-column sub-tile) unit whatever the unit holds: flush, barrier,
// refill issue.  This bench prices the alternative before any builder is written for it: a block = (row range x
// column slice), its elements sorted by COLUMN, streamed as 512-byte chunks of 64 x { value word, row << 16 | column - chunk base };
// x[column] is fetched with a per-lane global_load_dword (x is 6-10 MB: L2 / Infinity-Cache resident), eight gathers in flight behind
// eight chunks in flight, products go to 8-byte LDS accumulators with ds_add_u64 / ds_add_f64.  No units, no barriers, no refills.
// Column-sorted chunks make a wavefront's gather touch nnz_block / lines_of_x_slice elements per 128-byte line.
//   modes: 0 = everything, 1 = no LDS accumulate, 2 = no gather (x word := column), 3 = stream only
// Synthetic blocks are generated on the device: columns evenly spread with jitter (sorted by construction), rows hashed.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

constexpr int kWaves = 16, kLanes = 64, kThreads = kWaves * kLanes;
constexpr uint32_t kChunkBytes = 512;

__device__ __forceinline__ uint32_t hash32(uint32_t a) {
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

// chunk (step s, wave w) of block b lives at ((b * steps + s) * 16 + w) * 512: the workgroup sweeps one contiguous region and its 16
// wavefronts are in the same column neighbourhood at the same time.
__global__ void generate(uint8_t* image, uint32_t* chunk_base, uint32_t steps, uint32_t rows, uint32_t col_lo, uint32_t col_span, uint32_t slices, uint32_t seed) {
    const uint32_t b = blockIdx.x, slice = b % slices;
    const uint64_t n = uint64_t(steps) * kWaves * kLanes;
    const uint32_t lo = col_lo + slice * col_span;
    for (uint64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t chunk = uint32_t(i / kLanes), lane = uint32_t(i % kLanes);
        const uint64_t first = uint64_t(chunk) * kLanes;
        const uint32_t base = lo + uint32_t(first * col_span / n);
        uint32_t col = lo + uint32_t((i * col_span + hash32(uint32_t(i) * 2654435761u + b + seed) % (col_span / 2 + 1)) / n);
        if (col >= lo + col_span) col = lo + col_span - 1;
        uint32_t d = col - base;
        if (d > 0xffffu) d = 0xffffu;
        const uint32_t row = hash32(uint32_t(i) ^ (b * 0x9e3779b9u) ^ seed) % rows;
        const uint32_t s = chunk / kWaves, w = chunk % kWaves;
        uint32_t* e = reinterpret_cast<uint32_t*>(image + ((uint64_t(b) * steps + s) * kWaves + w) * kChunkBytes) + lane * 2;
        e[0] = 0x3f000000u + (hash32(uint32_t(i) + 7u * b) & 0x7fffffu);      // a float in [0.5, 1) / a Q8.24 word
        e[1] = row << 16 | d;
        if (lane == 0) chunk_base[(uint64_t(b) * kWaves + w) * steps + s] = base;   // (steps is a multiple of 8: main)
    }
}

#define RING_AGPRS "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
                   "a18", "a19", "a20", "a21", "a22", "a23"

template <int K>
__device__ __forceinline__ void issue_stream(const uint8_t* base, uint32_t off) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 a[%0:%1], %2, %3 nt" ::"n"(2 * K), "n"(2 * K + 1), "v"(off), "s"(base) : "memory", RING_AGPRS);
}
#ifndef GATHER_POLICY
#define GATHER_POLICY ""
#endif
template <int K>
__device__ __forceinline__ void issue_gather(const uint32_t* x, uint32_t byte_off) {
    asm volatile("s_nop 4\n\tglobal_load_dword a[%0], %1, %2 " GATHER_POLICY ::"n"(16 + K), "v"(byte_off), "s"(x) : "memory", RING_AGPRS);
}
// one counted wait: chunk s AND the gather issued just before it (for chunk s - D) have landed
template <int K, int D>
__device__ __forceinline__ void take(uint32_t& value, uint32_t& where, uint32_t& xv) {
    asm volatile("s_waitcnt vmcnt(%6)\n\tv_accvgpr_read_b32 %0, a[%3]\n\tv_accvgpr_read_b32 %1, a[%4]\n\tv_accvgpr_read_b32 %2, a[%5]"
                 : "=v"(value), "=v"(where), "=v"(xv) : "n"(2 * K), "n"(2 * K + 1), "n"(16 + K), "n"(2 * (D - 1)) : "memory");
}
__device__ __forceinline__ const uint8_t* scalar_pointer(const void* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a >> 32));
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(a));
    return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ uint32_t q8_24_mul(uint32_t a, uint32_t b) {
    const uint64_t wide = static_cast<uint64_t>(a) * b + 0x800000ull;
    uint32_t hi = static_cast<uint32_t>(wide >> 32);
    const uint32_t lo = static_cast<uint32_t>(wide);
    asm("" : "+v"(hi));
    const uint32_t r = __builtin_amdgcn_alignbit(hi, lo, 24);
    return (hi >> 24) ? 0xffffffffu : r;
}

struct Lane {
    uint32_t value[8], row[8];
};

// kBases = 1: the eight chunk bases of a round come with ONE vector load (lanes 0-7 read base[8 r + lane]) issued just before the first
// chunk of that round -- vmcnt retires in order, so when that chunk has landed the bases have too -- and are handed to the scalar unit
// with v_readlane: no scalar load, no lgkmcnt wait anywhere in the loop.
__device__ __forceinline__ void issue_bases(const uint32_t* table, uint32_t byte_off) {
    asm volatile("s_nop 4\n\tglobal_load_dword a24, %0, %1" ::"v"(byte_off), "s"(table) : "memory", RING_AGPRS, "a24");
}
template <int K, int D, int kExtra>
__device__ __forceinline__ void take_n(uint32_t& value, uint32_t& where, uint32_t& xv) {
    asm volatile("s_waitcnt vmcnt(%6)\n\tv_accvgpr_read_b32 %0, a[%3]\n\tv_accvgpr_read_b32 %1, a[%4]\n\tv_accvgpr_read_b32 %2, a[%5]"
                 : "=v"(value), "=v"(where), "=v"(xv) : "n"(2 * K), "n"(2 * K + 1), "n"(16 + K), "n"(2 * (D - 1) + kExtra) : "memory");
}

template <bool kFloat, int kMode, int K, int kBases = 0>
__device__ __forceinline__ void step(Lane& st, const uint8_t* stream, const uint8_t* x, uint32_t s, uint32_t last, uint32_t lane_off, uint32_t base,
                                     void* acc, const uint32_t* table = nullptr, uint32_t table_off = 0, uint32_t* sb = nullptr) {
    constexpr int D = 8;
    uint32_t value, where, xv;
    if (kBases == 0) take<K, D>(value, where, xv);
    else take_n<K, D, K == 0 ? 0 : 1>(value, where, xv);
    if (kBases == 1 && K == 0) {          // the round's bases have landed with its first chunk: hand them to the scalar unit, fetch the next round's
        uint32_t v;
        asm volatile("v_accvgpr_read_b32 %0, a24" : "=v"(v) :: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) sb[k] = __builtin_amdgcn_readlane(v, k);
        issue_bases(table, table_off);
        base = sb[0];
    }
    // the element taken D steps ago: its x word has just landed
    if (kMode & 2) xv = st.row[K];
    if (!(kMode & 1)) {
        if (kFloat) atomicAdd(reinterpret_cast<double*>(acc) + st.row[K], static_cast<double>(__uint_as_float(st.value[K]) * __uint_as_float(xv)));
        else atomicAdd(reinterpret_cast<unsigned long long*>(acc) + st.row[K], static_cast<unsigned long long>(q8_24_mul(st.value[K], xv)));
    } else {
        asm volatile("" ::"v"(xv), "v"(st.value[K]), "v"(st.row[K]));
    }
    st.value[K] = value;
    st.row[K] = where >> 16;
    const uint32_t col = base + (where & 0xffffu);
    if (!(kMode & 2)) issue_gather<K>(reinterpret_cast<const uint32_t*>(x), col * 4u);
    else issue_gather<K>(reinterpret_cast<const uint32_t*>(x), (base * 4u & 0x3ffe00u) + lane_off / 2);      // keeps the wait count: one coalesced load near the chunk's columns
    issue_stream<K>(stream, min(s + D, last) * (kWaves * kChunkBytes) + lane_off);
}

template <bool kFloat, int kMode, int kBases = 0>
__global__ __launch_bounds__(kThreads) void gather_kernel(const uint8_t* __restrict__ image, const uint32_t* __restrict__ chunk_base,
                                                          const uint32_t* __restrict__ x, uint32_t* __restrict__ out, uint32_t steps, uint32_t rows) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid / 64u);
    uint32_t wg = blockIdx.x;
    if ((gridDim.x & 7u) == 0) wg = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const uint8_t* stream = scalar_pointer(image + (uint64_t(wg) * steps * kWaves + wave) * kChunkBytes);
    const uint8_t* xs = scalar_pointer(x);
    const __attribute__((address_space(4))) uint32_t* bases =
        (const __attribute__((address_space(4))) uint32_t*)(chunk_base + (uint64_t(wg) * kWaves + wave) * steps);
    const uint32_t lane_off = lane * 8u, last = steps - 1;
    // prime: 8 chunks in flight, 8 dummy gathers (slot order = the steady-state order: gather K, then chunk K)
    Lane st;
#pragma unroll
    for (int k = 0; k < 8; ++k) { st.value[k] = 0; st.row[k] = rows; }
    const uint32_t* table = reinterpret_cast<const uint32_t*>(scalar_pointer(chunk_base + (uint64_t(wg) * kWaves + wave) * steps));
    const uint32_t lane8 = min(lane, 7u) * 4u, table_last = (steps - 1) / 8 * 32;      // (the table is padded to whole rounds)
    if (kBases == 1) issue_bases(table, lane8);
    [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
        ((issue_gather<Ks>(reinterpret_cast<const uint32_t*>(xs), lane_off), issue_stream<Ks>(stream, min(uint32_t(Ks), last) * (kWaves * kChunkBytes) + lane_off)), ...);
    }(std::make_integer_sequence<int, 8>());
    for (uint32_t i = tid; i <= rows; i += kThreads) acc[i] = 0;
    __syncthreads();
    uint32_t b[8];
    if (kBases == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = bases[min(uint32_t(k), last)];
    }
    for (uint32_t s0 = 0; s0 < steps + 8; s0 += 8) {       // 8 extra steps drain the gathers of the last 8 chunks (they re-read the last chunk)
        if (kBases == 1) {
            const uint32_t next_off = min((s0 + 8) * 4u, table_last) + lane8;
            [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
                (step<kFloat, kMode, Ks, 1>(st, stream, xs, s0 + Ks, last, lane_off, b[Ks], acc, table, next_off, b), ...);
            }(std::make_integer_sequence<int, 8>());
            continue;
        }
        uint32_t nb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) nb[k] = bases[min(s0 + 8 + k, last)];
        [&]<int... Ks>(std::integer_sequence<int, Ks...>) {
            (step<kFloat, kMode, Ks>(st, stream, xs, s0 + Ks, last, lane_off, b[Ks], acc), ...);
        }(std::make_integer_sequence<int, 8>());
#pragma unroll
        for (int k = 0; k < 8; ++k) b[k] = nb[k];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", RING_AGPRS);
    const unsigned long long flushed = atomicAdd(acc + rows, 0ull);
    asm volatile("" ::"v"(flushed));
    __syncthreads();
    for (uint32_t i = tid; i < rows; i += kThreads) {
        const unsigned long long v = acc[i];
        out[uint64_t(wg) * rows + i] = kFloat ? __float_as_uint(static_cast<float>(__longlong_as_double(v))) : (v > 0xffffffffull ? 0xffffffffu : uint32_t(v));
    }
}

struct Case { const char* name; uint64_t nnz; uint32_t ncols, rows, ranges, slices; };

template <bool kFloat, int kMode, int kBases = 0>
float run_mode(const Case& c, const uint8_t* image, const uint32_t* bases, const uint32_t* x, uint32_t* out, uint32_t steps, int reps) {
    const uint32_t blocks = c.ranges * c.slices;
    const uint32_t lds = (c.rows + 1) * 8u;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gather_kernel<kFloat, kMode, kBases>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gather_kernel<kFloat, kMode, kBases>), dim3(blocks), dim3(kThreads), lds, 0, image, bases, x, out, steps, c.rows);
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gather_kernel<kFloat, kMode, kBases>), dim3(blocks), dim3(kThreads), lds, 0, image, bases, x, out, steps, c.rows);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    CHECK(hipGetLastError());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 50;
    const Case cases[] = {
        // pokec stand-in: 1.63 M x 1.63 M, 30.6 M non-zeros (OWNER24 today: 89 us kernel, 96 us step)
        {"pokec  20K-row ranges x 3 slices", 30600000ull, 1632768, 20000, 82, 3},
        {"pokec  20K-row ranges x 2 slices", 30600000ull, 1632768, 20000, 82, 2},
        {"pokec  13K-row ranges x 2 slices", 30600000ull, 1632768, 12800, 128, 2},
        {"pokec  6.4K-row ranges x 1 slice", 30600000ull, 1632768, 6400, 256, 1},
        // ogbn-products stand-in: 2.45 M x 2.45 M, 123.7 M non-zeros (OWNER24 today: 200 us kernel, 207 us step)
        {"ogbn   20K-row ranges x 2 slices", 123700000ull, 2449408, 20000, 123, 2},
        {"ogbn   9.6K-row ranges x 1 slice", 123700000ull, 2449408, 9600, 256, 1},
        {"ogbn   19K-row ranges x 1 slice (128 wg)", 123700000ull, 2449408, 19200, 128, 1},
        // ogbl-ppa (DELTA today: 54.4 us): x is 2.3 MB
        {"ppa    2.25K-row ranges x 1 slice", 42460000ull, 576384, 2252, 256, 1},
        {"ppa    9K-row ranges x 4 slices", 42460000ull, 576384, 9006, 64, 4},
    };
    uint32_t* x; uint32_t* out; uint8_t* image; uint32_t* bases;
    const size_t cap = 1100ull << 20;
    CHECK(hipMalloc(&image, cap));
    CHECK(hipMalloc(&bases, cap / 128));
    CHECK(hipMalloc(&x, 16 << 20));
    CHECK(hipMalloc(&out, size_t(256) * 3 * 20001 * 4 + (64 << 20)));
    std::vector<uint32_t> hx(4 << 20);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0x3f800000u + uint32_t(i & 0xffff);
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    printf("%-44s %8s %7s | %9s %9s %9s %9s | %9s %9s | elements per 128-B line of x\n", "case", "blocks", "MB", "fixed all", "no LDS", "no gather", "stream", "float all", "no LDS");
    for (const Case& c : cases) {
        const uint32_t blocks = c.ranges * c.slices;
        const uint32_t steps = (uint32_t((c.nnz / blocks + kWaves * kLanes - 1) / (kWaves * kLanes)) + 7u) & ~7u;
        const size_t bytes = size_t(blocks) * steps * kWaves * kChunkBytes;
        if (bytes > cap) { printf("%s: image too large\n", c.name); continue; }
        const uint32_t span = c.ncols / c.slices;
        hipLaunchKernelGGL(generate, dim3(blocks), dim3(1024), 0, 0, image, bases, steps, c.rows, 0u, span, c.slices, 12345u);
        CHECK(hipDeviceSynchronize());
        const float f0 = run_mode<false, 0>(c, image, bases, x, out, steps, reps), f1 = run_mode<false, 1>(c, image, bases, x, out, steps, reps);
        const float f2 = run_mode<false, 2>(c, image, bases, x, out, steps, reps), f3 = run_mode<false, 3>(c, image, bases, x, out, steps, reps);
        const float g0 = run_mode<true, 0>(c, image, bases, x, out, steps, reps), g1 = run_mode<true, 1>(c, image, bases, x, out, steps, reps);
        const double per_line = double(c.nnz) / blocks / (double(span) * 4 / 128);
        printf("%-44s %8u %7.1f | %9.1f %9.1f %9.1f %9.1f | %9.1f %9.1f | %.1f\n", c.name, blocks, bytes / 1e6, f0, f1, f2, f3, g0, g1, per_line);
        const float v0 = run_mode<false, 0, 1>(c, image, bases, x, out, steps, reps), v1 = run_mode<false, 1, 1>(c, image, bases, x, out, steps, reps);
        const float v2 = run_mode<false, 2, 1>(c, image, bases, x, out, steps, reps), v3 = run_mode<false, 3, 1>(c, image, bases, x, out, steps, reps);
        const float w0 = run_mode<true, 0, 1>(c, image, bases, x, out, steps, reps), w1 = run_mode<true, 1, 1>(c, image, bases, x, out, steps, reps);
        printf("%-44s %8s %7s | %9.1f %9.1f %9.1f %9.1f | %9.1f %9.1f | = %.2f TB/s of stream (fixed, all)\n", "   ... bases by vector load", "", "", v0, v1, v2, v3, w0, w1, bytes / (v0 * 1e-6) / 1e12);
        fflush(stdout);
    }
    return 0;
}
