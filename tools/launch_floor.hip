// launch_floor.hip -- what a kernel costs before it does anything (round 5): K back-to-back launches on one stream between two events,
// for an empty kernel and for kernels that are nothing but a chain of 1 .. 4 DEPENDENT global loads (each load's address comes out of
// the previous one) followed by one store, at several grid shapes.  The chain of the LIGHT kernel is descriptor -> elements -> x -> store.
//   hipcc -O3 --offload-arch=gfx950 -o tools/launch_floor.bin tools/launch_floor.hip && tools/launch_floor.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int kDepth>
__global__ void chain_kernel(const uint32_t* __restrict__ table, uint32_t* __restrict__ out, uint32_t mask) {
    uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) & mask;
#pragma unroll
    for (int d = 0; d < kDepth; ++d) i = table[i] & mask;      // dependent: address from the previous load
    if (kDepth == 0) { if (threadIdx.x == 1024) out[0] = i; return; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = i;
}
__global__ void lds_barrier_kernel(const uint32_t* __restrict__ table, uint32_t* __restrict__ out, uint32_t mask) {
    __shared__ uint32_t acc[2048];
    for (uint32_t k = threadIdx.x; k < 2048; k += blockDim.x) acc[k] = 0;
    uint32_t i = table[(blockIdx.x * blockDim.x + threadIdx.x) & mask] & mask;
    __syncthreads();
    atomicAdd(&acc[i & 2047], table[i]);
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[threadIdx.x];
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const uint32_t n = 1u << 22, mask = n - 1;
    std::vector<uint32_t> h(n);
    uint32_t s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = s >> 8; }
    uint32_t *table, *out;
    CHECK(hipMalloc(&table, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemcpy(table, h.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int K = 2000;
    auto timed = [&](const char* what, auto launch) {
        for (int i = 0; i < 200; ++i) launch();
        hipStreamSynchronize(st);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, st);
            for (int i = 0; i < K; ++i) launch();
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            best = ms < best ? ms : best;
        }
        printf("%-64s %6.2f us per launch\n", what, best * 1e3f / K);
    };
    for (int shape = 0; shape < 4; ++shape) {
        const dim3 grid(shape == 0 ? 64 : shape == 1 ? 256 : shape == 2 ? 1024 : 256), block(shape == 3 ? 1024 : 256);
        char name[128];
#define RUN(D) snprintf(name, sizeof name, "%4u x %4u threads, %d dependent loads + store", grid.x, block.x, D); \
        timed(name, [&] { hipLaunchKernelGGL(chain_kernel<D>, grid, block, 0, st, table, out, mask); });
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
        snprintf(name, sizeof name, "%4u x %4u threads, 2 dependent loads, LDS zero + atomic + 2 barriers", grid.x, block.x);
        timed(name, [&] { hipLaunchKernelGGL(lds_barrier_kernel, grid, block, 0, st, table, out, mask); });
    }
    return 0;
}
