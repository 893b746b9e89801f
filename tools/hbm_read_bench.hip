// tools/hbm_read_bench.hip — what does a plain streaming READ reach on this MI355X?  (context for roofline.frac)
// Variants: dwordx2 / dwordx4 per lane, default / nt cache policy, 256 / 512 / 2048 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int kBytes, bool kNt>
__global__ __launch_bounds__(1024) void read_kernel(const uint8_t* __restrict__ src, size_t bytes_per_wg, uint32_t* sink) {
    const uint8_t* p = src + (size_t)blockIdx.x * bytes_per_wg + threadIdx.x * kBytes;
    const size_t stride = 1024 * kBytes;
    uint32_t acc = 0;
    for (size_t off = 0; off + stride * 4 <= bytes_per_wg; off += stride * 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (kBytes == 8) {
                uint64_t v;
                if (kNt) v = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(p + off + j * stride));
                else v = *reinterpret_cast<const uint64_t*>(p + off + j * stride);
                acc ^= (uint32_t)v ^ (uint32_t)(v >> 32);
            } else {
                typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                u32x4 v;
                if (kNt) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p + off + j * stride));
                else v = *reinterpret_cast<const u32x4*>(p + off + j * stride);
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int kBytes, bool kNt>
void run(const char* name, const uint8_t* d, size_t total, int wgs, uint32_t* sink) {
    size_t per = total / wgs / (1024 * kBytes * 4) * (1024 * kBytes * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((read_kernel<kBytes, kNt>), dim3(wgs), dim3(1024), 0, 0, d, per, sink);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((read_kernel<kBytes, kNt>), dim3(wgs), dim3(1024), 0, 0, d, per, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s wgs %5d  %8.1f us per pass  %7.1f GB/s\n", name, wgs, ms / reps * 1e3, (double)per * wgs / (ms / reps * 1e-3) / 1e9);
}


// the consumer loop of spmv_rowblock_kernel in isolation: kWaves of the 16 wavefronts stream 512-byte chunks that are
// interleaved across the wavefronts, 8 asm loads in flight per lane, one s_waitcnt vmcnt(7) + one reload per step
// kPolicy: the cache-policy bits of the stream loads: 0 = nt (what the kernels use), 1 = none, 2 = sc1, 3 = sc0 sc1, 4 = sc0 sc1 nt, 5 = sc1 nt
template <int kPolicy>
__device__ __forceinline__ void ld(uint64_t& dst, const void* addr) {
    if (kPolicy == 0) asm volatile("global_load_dwordx2 %0, %1, off nt" : "=v"(dst) : "v"(addr) : "memory");
    if (kPolicy == 1) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(addr) : "memory");
    if (kPolicy == 2) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(dst) : "v"(addr) : "memory");
    if (kPolicy == 3) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(dst) : "v"(addr) : "memory");
    if (kPolicy == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt" : "=v"(dst) : "v"(addr) : "memory");
    if (kPolicy == 5) asm volatile("global_load_dwordx2 %0, %1, off sc1 nt" : "=v"(dst) : "v"(addr) : "memory");
}
template <int N> __device__ __forceinline__ void wt(uint64_t& v) { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(N) : "memory"); }
template <int kWaves, int kPolicy = 0>
__global__ __launch_bounds__(1024) void ring_kernel(const uint8_t* __restrict__ src, size_t bytes_per_wg, uint32_t* sink) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63;
    if (wave >= kWaves) return;
    const uint8_t* p = src + (size_t)blockIdx.x * bytes_per_wg + wave * 512 + lane * 8;
    const uint32_t total = uint32_t(bytes_per_wg / (512 * kWaves)), last = total - 1;
    const size_t stride = 512 * kWaves;
    uint64_t buf[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) ld<kPolicy>(buf[k], p + (size_t)min((uint32_t)k, last) * stride);
    uint32_t acc = 0;
    for (uint32_t base = 0; base < total; base += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            wt<7>(buf[k]);
            acc ^= (uint32_t)buf[k] ^ (uint32_t)(buf[k] >> 32);
            ld<kPolicy>(buf[k], p + (size_t)min(base + k + 8, last) * stride);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int kWaves, int kPolicy = 0>
void run_ring(const char* name, const uint8_t* d, size_t total, int wgs, uint32_t* sink) {
    size_t per = total / wgs / (512 * kWaves * 8) * (512 * kWaves * 8);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ring_kernel<kWaves, kPolicy>), dim3(wgs), dim3(1024), 0, 0, d, per, sink);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ring_kernel<kWaves, kPolicy>), dim3(wgs), dim3(1024), 0, 0, d, per, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s wgs %5d  %8.1f us per pass  %7.1f GB/s\n", name, wgs, ms / reps * 1e3, (double)per * wgs / (ms / reps * 1e-3) / 1e9);
}

int main() {
    const size_t total = 1ull << 30;   // 1 GiB: well past the 256 MiB Infinity Cache
    uint8_t* d; uint32_t* sink;
    hipMalloc(&d, total); hipMalloc(&sink, 64);
    hipMemset(d, 1, total);
    for (int wgs : {256, 512, 2048}) {
        run<16, false>("dwordx4", d, total, wgs, sink);
        run<16, true>("dwordx4 nt", d, total, wgs, sink);
        run<8, false>("dwordx2", d, total, wgs, sink);
        run<8, true>("dwordx2 nt", d, total, wgs, sink);
    }
    // 341 MB like the ogbl-ppa element stream (partly Infinity-Cache resident between passes)
    run<16, true>("dwordx4 nt, 341 MB", d, 341ull << 20, 256, sink);
    run<8, true>("dwordx2 nt, 341 MB", d, 341ull << 20, 256, sink);
    run_ring<14>("ring8 asm, 14 waves, 341 MB", d, 341ull << 20, 256, sink);
    run_ring<16>("ring8 asm, 16 waves, 341 MB", d, 341ull << 20, 256, sink);
    run_ring<14>("ring8 asm, 14 waves, 1 GiB", d, total, 256, sink);
    run_ring<8>("ring8 asm, 8 waves, 341 MB", d, 341ull << 20, 256, sink);
    // cache-policy bits of the stream loads, 1 GiB (nothing survives in the Infinity Cache between passes)
    run_ring<14, 0>("ring8 14 waves 1 GiB nt", d, total, 256, sink);
    run_ring<14, 1>("ring8 14 waves 1 GiB (none)", d, total, 256, sink);
    run_ring<14, 2>("ring8 14 waves 1 GiB sc1", d, total, 256, sink);
    run_ring<14, 3>("ring8 14 waves 1 GiB sc0 sc1", d, total, 256, sink);
    run_ring<14, 4>("ring8 14 waves 1 GiB sc0 sc1 nt", d, total, 256, sink);
    run_ring<14, 5>("ring8 14 waves 1 GiB sc1 nt", d, total, 256, sink);
    return 0;
}
