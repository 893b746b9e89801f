// tools/record_stream_bench.hip — how fast do per-wavefront record streams read, by record shape?
// Every consumer wavefront (14 of 16 per workgroup, one workgroup per CU) walks its own contiguous stream with a ring
// of kDepth records in flight, exactly like spmv_rowblock_kernel's consumer loop; only the loads differ:
//   shape 0: dwordx2 per lane                      (512-byte records: value + packed row/col)
//   shape 1: dword + ushort per lane               (384-byte records: value + 16-bit gap)
//   shape 2: dwordx2 + dword per lane              (768 bytes  = 2 records of shape 1)
//   shape 3: dwordx4 + dwordx2 per lane            (1536 bytes = 4 records of shape 1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int kShape> struct Shape;
template <> struct Shape<0> { static constexpr uint32_t bytes = 512; };
template <> struct Shape<1> { static constexpr uint32_t bytes = 384; };
template <> struct Shape<2> { static constexpr uint32_t bytes = 768; };
template <> struct Shape<3> { static constexpr uint32_t bytes = 1536; };

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int kShape>
struct Rec {
    u32x4 a; u32x2 b;
    __device__ __forceinline__ void load(const uint8_t* rec, uint32_t lane) {
        const uint64_t p = reinterpret_cast<uint64_t>(rec);
        const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(p >> 32)), lo = __builtin_amdgcn_readfirstlane(uint32_t(p));
        const uint64_t base = (uint64_t(hi) << 32) | lo;
        if (kShape == 0) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 nt" : "=&v"(b) : "v"(lane * 8u), "s"(base) : "memory");
        if (kShape == 1) asm volatile("s_nop 4\n\tglobal_load_dword %0, %2, %4 nt\n\tglobal_load_ushort %1, %3, %4 offset:256 nt" : "=&v"(a.x), "=&v"(a.y) : "v"(lane * 4u), "v"(lane * 2u), "s"(base) : "memory");
        if (kShape == 2) asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %2, %4 nt\n\tglobal_load_dword %1, %3, %4 offset:512 nt" : "=&v"(b), "=&v"(a.x) : "v"(lane * 8u), "v"(lane * 4u), "s"(base) : "memory");
        if (kShape == 3) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %4 nt\n\tglobal_load_dwordx2 %1, %3, %4 offset:1024 nt" : "=&v"(a), "=&v"(b) : "v"(lane * 16u), "v"(lane * 8u), "s"(base) : "memory");
    }
    template <int kOutstanding> __device__ __forceinline__ void wait() { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(kOutstanding) : "memory"); }
    __device__ __forceinline__ uint32_t fold() const { return a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y; }
};

template <int kShape, int kDepth, int kWaves>
__global__ __launch_bounds__(1024) void ring_kernel(const uint8_t* __restrict__ src, size_t bytes_per_wave, uint32_t* sink) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63;
    if (wave >= kWaves) return;
    constexpr uint32_t kBytes = Shape<kShape>::bytes;
    constexpr int kPer = kShape == 0 ? 1 : 2;     // vector-memory instructions per record
    const uint8_t* p = src + (size_t(blockIdx.x) * kWaves + wave) * bytes_per_wave;
    const uint32_t total = uint32_t(bytes_per_wave / kBytes), last = total - 1;
    Rec<kShape> r[kDepth];
#pragma unroll
    for (int k = 0; k < kDepth; ++k) r[k].load(p + size_t(min(uint32_t(k), last)) * kBytes, lane);
    uint32_t acc = 0;
    for (uint32_t base = 0; base < total; base += kDepth) {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            r[k].template wait<kPer * (kDepth - 1)>();
            acc ^= r[k].fold();
            r[k].load(p + size_t(min(base + k + kDepth, last)) * kBytes, lane);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int kShape, int kDepth, int kWaves>
void run(const char* name, const uint8_t* d, size_t total, uint32_t* sink) {
    constexpr uint32_t kBytes = Shape<kShape>::bytes;
    const int wgs = 256;
    const size_t per = total / wgs / kWaves / (kBytes * kDepth) * (kBytes * kDepth);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring_kernel<kShape, kDepth, kWaves>), dim3(wgs), dim3(1024), 0, 0, d, per, sink);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ring_kernel<kShape, kDepth, kWaves>), dim3(wgs), dim3(1024), 0, 0, d, per, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = double(per) * wgs * kWaves;
    printf("%-44s %6.1f MB  %7.1f us  %7.1f GB/s  (%u B x %d in flight per wave)\n", name, bytes / 1e6, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e9, kBytes, kDepth);
}

int main() {
    const size_t cap = 1ull << 30;
    uint8_t* d; uint32_t* sink;
    hipMalloc(&d, cap); hipMalloc(&sink, 64);
    hipMemset(d, 1, cap);
    const size_t big = 341000000, small = 280000000;
    run<0, 8, 14>("dwordx2 512 B, depth 8", d, big, sink);
    run<0, 8, 14>("dwordx2 512 B, depth 8", d, small, sink);
    run<1, 8, 14>("dword+ushort 384 B, depth 8", d, small, sink);
    run<1, 12, 14>("dword+ushort 384 B, depth 12", d, small, sink);
    run<1, 16, 14>("dword+ushort 384 B, depth 16", d, small, sink);
    run<2, 4, 14>("dwordx2+dword 768 B, depth 4", d, small, sink);
    run<2, 6, 14>("dwordx2+dword 768 B, depth 6", d, small, sink);
    run<2, 8, 14>("dwordx2+dword 768 B, depth 8", d, small, sink);
    run<3, 2, 14>("dwordx4+dwordx2 1536 B, depth 2", d, small, sink);
    run<3, 4, 14>("dwordx4+dwordx2 1536 B, depth 4", d, small, sink);
    run<1, 8, 16>("dword+ushort 384 B, depth 8, 16 waves", d, small, sink);
    run<2, 6, 16>("dwordx2+dword 768 B, depth 6, 16 waves", d, small, sink);
    return 0;
}
