// tools/record_check_bench.hip — do counted waits hold for the DELTA record loads?  Every value word / gap of a
// 280 MB record image encodes its own index; 14 wavefronts per workgroup stream their records with a ring of kDepth
// loads in flight and verify each record right after its s_waitcnt vmcnt(2*(kDepth-1)).  Any error = data consumed
// before it landed (or loaded from the wrong address).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr uint32_t kRec = 384;

template <bool kSaddr>
__device__ __forceinline__ void rload(uint32_t& val, uint32_t& gap, const uint8_t* rec, uint32_t lane) {
    if (kSaddr) {
        const uint64_t p = reinterpret_cast<uint64_t>(rec);
        const uint32_t hi = __builtin_amdgcn_readfirstlane(uint32_t(p >> 32)), lo = __builtin_amdgcn_readfirstlane(uint32_t(p));
        const uint64_t base = (uint64_t(hi) << 32) | lo;
        asm volatile("s_nop 4\n\tglobal_load_dword %0, %2, %4 nt\n\tglobal_load_ushort %1, %3, %4 offset:256 nt"
                     : "=&v"(val), "=&v"(gap) : "v"(lane * 4u), "v"(lane * 2u), "s"(base) : "memory");
    } else {
        asm volatile("global_load_dword %0, %1, off nt" : "=v"(val) : "v"(rec + lane * 4u) : "memory");
        asm volatile("global_load_ushort %0, %1, off nt" : "=v"(gap) : "v"(rec + 256 + lane * 2u) : "memory");
    }
}
template <int N> __device__ __forceinline__ void rwait(uint32_t& val, uint32_t& gap) { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(val), "+v"(gap) : "n"(N) : "memory"); }

template <bool kSaddr, int kDepth, int kSlack>
__global__ __launch_bounds__(1024) void check_kernel(const uint8_t* __restrict__ src, uint32_t recs_per_wave, uint32_t* errors) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / 64), lane = threadIdx.x & 63;
    if (wave >= 14) return;
    const uint32_t stream = blockIdx.x * 14 + wave;
    const uint8_t* p = src + size_t(stream) * recs_per_wave * kRec;
    const uint32_t last = recs_per_wave - 1;
    uint32_t val[kDepth], gap[kDepth];
#pragma unroll
    for (int k = 0; k < kDepth; ++k) rload<kSaddr>(val[k], gap[k], p + size_t(min(uint32_t(k), last)) * kRec, lane);
    uint32_t bad = 0;
    for (uint32_t base = 0; base < recs_per_wave; base += kDepth) {
#pragma unroll
        for (int k = 0; k < kDepth; ++k) {
            const uint32_t s = base + k;
            rwait<2 * (kDepth - 1) - kSlack>(val[k], gap[k]);
            const uint32_t id = (stream * recs_per_wave + min(s, last)) * 64u + lane;
            bad += (val[k] != id) + (gap[k] != (id & 0xffffu));
            rload<kSaddr>(val[k], gap[k], p + size_t(min(s + kDepth, last)) * kRec, lane);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (bad) atomicAdd(errors, bad);
}

template <bool kSaddr, int kDepth, int kSlack>
void run(const char* name, const uint8_t* d, uint32_t recs_per_wave, uint32_t* d_err) {
    uint32_t total = 0;
    for (int rep = 0; rep < 20; ++rep) {
        hipMemset(d_err, 0, 4);
        hipLaunchKernelGGL((check_kernel<kSaddr, kDepth, kSlack>), dim3(256), dim3(1024), 0, 0, d, recs_per_wave, d_err);
        uint32_t e = 0;
        hipMemcpy(&e, d_err, 4, hipMemcpyDeviceToHost);
        total += e;
    }
    printf("%-40s errors over 20 passes: %u\n", name, total);
}

int main() {
    const uint32_t streams = 256 * 14, recs_per_wave = 200;
    const size_t bytes = size_t(streams) * recs_per_wave * kRec;
    std::vector<uint8_t> h(bytes);
    for (uint32_t r = 0; r < streams * recs_per_wave; ++r)
        for (uint32_t l = 0; l < 64; ++l) {
            const uint32_t id = r * 64 + l;
            reinterpret_cast<uint32_t*>(h.data() + size_t(r) * kRec)[l] = id;
            reinterpret_cast<uint16_t*>(h.data() + size_t(r) * kRec + 256)[l] = uint16_t(id);
        }
    uint8_t* d; uint32_t* d_err;
    hipMalloc(&d, bytes + 4096); hipMalloc(&d_err, 4);
    hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    printf("%zu MB of records\n", bytes / 1000000);
    run<true, 8, 0>("saddr, depth 8, vmcnt(14)", d, recs_per_wave, d_err);
    run<false, 8, 0>("vaddr, depth 8, vmcnt(14)", d, recs_per_wave, d_err);
    run<true, 8, 1>("saddr, depth 8, vmcnt(13)", d, recs_per_wave, d_err);
    run<true, 8, 2>("saddr, depth 8, vmcnt(12)", d, recs_per_wave, d_err);
    run<true, 16, 0>("saddr, depth 16, vmcnt(30)", d, recs_per_wave, d_err);
    run<true, 4, 0>("saddr, depth 4, vmcnt(6)", d, recs_per_wave, d_err);
    run<false, 4, 0>("vaddr, depth 4, vmcnt(6)", d, recs_per_wave, d_err);
    return 0;
}
