// tools/lds_atomic_bench.hip — throughput of LDS atomic adds by type, 16 wavefronts per CU, pseudo-random rows
// (what the row accumulators of spmv_rowblock_kernel see in sparse blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T> __device__ __forceinline__ void lds_add(T* p, T v) { atomicAdd(p, v); }

template <typename T, int kRows>
__global__ __launch_bounds__(1024) void k(uint32_t iters, T* sink) {
    __shared__ T acc[kRows];
    for (uint32_t i = threadIdx.x; i < kRows; i += 1024) acc[i] = T(0);
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (uint32_t i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        lds_add(&acc[(h >> 8) % kRows], T(1));
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = acc[0];
}

// round 4 (SWEEP's fixed-point accumulators): a RETURNING 32-bit add whose old value gives the carry, + the rare ds_or_b32 of a carry bit
template <int kRows, int kMode>      // 0: returning add, result unused late; 1: + carry check and ds_or_b32; 2: carry check deferred by one iteration
__global__ __launch_bounds__(1024) void k_rtn(uint32_t iters, uint32_t* sink) {
    __shared__ uint32_t acc[kRows + kRows / 32];
    for (uint32_t i = threadIdx.x; i < kRows + kRows / 32; i += 1024) acc[i] = 0;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u, keep = 0, prev_old = 0, prev_v = 0, prev_row = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t row = (h >> 8) % kRows, v = (h >> 3) | 1u;
        const uint32_t old = atomicAdd(&acc[row], v);
        if (kMode == 0) keep ^= old;
        if (kMode == 1 && old + v < old) atomicOr(&acc[kRows + (row >> 5)], 1u << (row & 31u));
        if (kMode == 2) {
            if (prev_old + prev_v < prev_old) atomicOr(&acc[kRows + (prev_row >> 5)], 1u << (prev_row & 31u));
            prev_old = old; prev_v = v; prev_row = row;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = acc[0] ^ keep ^ prev_old;
}
// round 5: is the RETURNING float add as slow as the plain one (0.33 lanes/clk/CU)?  kMode 0: ds_add_rtn_f32, result kept; 1: the same sum as a
// 32-bit compare-and-swap loop
template <int kRows, int kMode>
__global__ __launch_bounds__(1024) void k_f32_rtn(uint32_t iters, float* sink) {
    __shared__ float acc[kRows];
    for (uint32_t i = threadIdx.x; i < kRows; i += 1024) acc[i] = 0.f;
    __syncthreads();
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float keep = 0.f;
    for (uint32_t i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t row = (h >> 8) % kRows;
        const float v = __uint_as_float(0x3f800000u | (h & 0xffffu));
        if (kMode == 0) keep += atomicAdd(&acc[row], v);
        else {
            uint32_t* w = reinterpret_cast<uint32_t*>(&acc[row]);
            uint32_t old = *w, assumed;
            do { assumed = old; old = atomicCAS(w, assumed, __float_as_uint(__uint_as_float(assumed) + v)); } while (old != assumed);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = acc[0] + keep;
}
template <int kMode>
void run_f32_rtn(const char* name) {
    float* sink; hipMalloc(&sink, 256 * 4);
    const uint32_t iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_f32_rtn<8192, kMode>), dim3(256), dim3(1024), 0, 0, iters, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_f32_rtn<8192, kMode>), dim3(256), dim3(1024), 0, 0, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_cu = double(iters) * 1024;
    printf("%-26s %8.1f us  %6.2f ns per wave-instruction (64 lanes)  %5.2f lanes/clk/CU @2.4GHz\n", name, ms * 1e3, ms * 1e6 / (per_cu / 64), per_cu / (ms * 1e-3 * 2.4e9));
}

template <int kMode>
void run_rtn(const char* name) {
    uint32_t* sink; hipMalloc(&sink, 256 * 4);
    const uint32_t iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_rtn<8192, kMode>), dim3(256), dim3(1024), 0, 0, iters, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_rtn<8192, kMode>), dim3(256), dim3(1024), 0, 0, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_cu = double(iters) * 1024;
    printf("%-26s %8.1f us  %6.2f ns per wave-instruction (64 lanes)  %5.2f lanes/clk/CU @2.4GHz\n", name, ms * 1e3, ms * 1e6 / (per_cu / 64), per_cu / (ms * 1e-3 * 2.4e9));
}

template <typename T>
void run(const char* name) {
    T* sink; hipMalloc(&sink, 256 * sizeof(T));
    const uint32_t iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, 8192>), dim3(256), dim3(1024), 0, 0, iters, sink);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, 8192>), dim3(256), dim3(1024), 0, 0, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_cu = double(iters) * 1024;   // atomics per CU
    printf("%-10s %8.1f us  %6.2f ns per wave-instruction (64 lanes)  %5.2f lanes/clk/CU @2.4GHz\n", name, ms * 1e3, ms * 1e6 / (per_cu / 64), per_cu / (ms * 1e-3 * 2.4e9));
}

int main() {
    run<unsigned int>("u32");
    run<unsigned long long>("u64");
    run<float>("f32");
    run<double>("f64");
    run_rtn<0>("u32 rtn");
    run_rtn<1>("u32 rtn + carry bit");
    run_rtn<2>("u32 rtn + deferred carry");
    run_f32_rtn<0>("f32 rtn");
    run_f32_rtn<1>("f32 by 32-bit CAS loop");
    return 0;
}
