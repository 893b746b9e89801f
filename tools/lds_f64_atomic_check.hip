// tools/lds_f64_atomic_check.hip — are LDS double atomic adds exact under same-address contention?
// 16 wavefronts add 1.0 (and 1u) `iters` times to kRows accumulators chosen so that many lanes of one instruction collide.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T, int kRows>
__global__ __launch_bounds__(1024) void k(uint32_t iters, T* out, int pattern) {
    __shared__ T acc[kRows];
    for (uint32_t i = threadIdx.x; i < kRows; i += 1024) acc[i] = T(0);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = 0; i < iters; ++i) {
        uint32_t row;
        if (pattern == 0) row = (lane / 8 + i) % kRows;              // 8 lanes per address, neighbouring rows
        else if (pattern == 1) row = (i + wave) % kRows;             // whole wavefront on one address
        else row = (lane * 7 + i * 13 + wave) % kRows;               // scattered
        atomicAdd(&acc[row], T(1));
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kRows; i += 1024) out[blockIdx.x * kRows + i] = acc[i];
}

template <typename T>
void run(const char* name) {
    constexpr int kRows = 16, kBlocks = 64;
    T* d; hipMalloc(&d, sizeof(T) * kRows * kBlocks);
    for (int pattern = 0; pattern < 3; ++pattern) {
        const uint32_t iters = 1000;
        hipLaunchKernelGGL((k<T, kRows>), dim3(kBlocks), dim3(1024), 0, 0, iters, d, pattern);
        T h[kRows * kBlocks];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0; double worst = 0;
        for (int b = 0; b < kBlocks; ++b) {
            double sum = 0;
            for (int r = 0; r < kRows; ++r) sum += double(h[b * kRows + r]);
            if (sum != 1024.0 * iters) { ++bad; worst = sum; }
        }
        printf("%-6s pattern %d: %d of %d workgroups lost updates (e.g. total %.0f instead of %.0f)\n", name, pattern, bad, kBlocks, worst, 1024.0 * iters);
    }
}

int main() {
    run<unsigned long long>("u64");
    run<double>("f64");
    run<float>("f32");
    return 0;
}
