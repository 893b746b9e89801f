// tools/lds_accum_bench.hip — how fast can a CU accumulate 4-byte FLOAT row sums in LDS?
//
// The float modes keep 8-byte (double) row accumulators because ds_add_f32 measured 9x slower than ds_add_f64
// (tools/lds_atomic_bench.hip).  Half-size accumulators would double the rows a workgroup owns and halve the x volume
// staged per SpMV on hyper-sparse matrices (ogbn-products), so this probes every way to get them:
//   add_f32           ds_add_f32, no return (the baseline: slow)
//   add_rtn_f32       ds_add_rtn_f32 (returning form)
//   add_f32_ftz       ds_add_f32 with MODE.FP_DENORM(single) = flush (does the LDS take a slow path for denormals?)
//   add_f64, add_u64, add_u32   reference points
//   rmw_f32           ds_read_b32 + v_add_f32 + ds_write_b32 on rows PRIVATE to the wavefront (LDS executes one
//                     wavefront's instructions in order, so no atomic is needed when no other wavefront touches the row
//                     and the lanes of one instruction hold distinct rows)
//   gather            ds_read_b32 of random words alone (the x gather)
//   gather+add_f64    what the kernel's float path does per element today
//   gather+rmw_f32    the candidate
// Rows are pseudo-random per lane; 16 wavefronts per CU, 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

enum Mode { kAddF32, kAddRtnF32, kAddF32Ftz, kAddF64, kAddU64, kAddU32, kRmwF32, kGather, kGatherAddF64, kGatherRmwF32, kGatherAddU64 };

constexpr int kRows = 8192;       // accumulators
constexpr int kXWords = 8192;     // x sub-tile

template <int kMode>
__global__ __launch_bounds__(1024) void k(uint32_t iters, float* sink) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[kRows * 8 + kXWords * 4];
    float* acc32 = reinterpret_cast<float*>(lds);
    double* acc64 = reinterpret_cast<double*>(lds);
    unsigned long long* accu64 = reinterpret_cast<unsigned long long*>(lds);
    uint32_t* accu32 = reinterpret_cast<uint32_t*>(lds);
    float* xs = reinterpret_cast<float*>(lds + kRows * 8);
    for (uint32_t i = threadIdx.x; i < (kRows * 8 + kXWords * 4) / 4; i += 1024) reinterpret_cast<uint32_t*>(lds)[i] = 0;
    __syncthreads();
    if (kMode == kAddF32Ftz) __builtin_amdgcn_s_setreg((4 << 6) | (1 << 11) | 1, 0);   // hwreg(MODE, offset 4, size 2) = 0: flush f32 denormals
    const uint32_t wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    uint32_t h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    float keep = 0.0f;
    for (uint32_t i = 0; i < iters; ++i) {
        h = h * 1664525u + 1013904223u;
        const uint32_t row = (h >> 8) % kRows;
        // wave-private rows with distinct rows per instruction: 512 rows per wavefront, lane picks row 8*lane' + r
        const uint32_t prow = wave * (kRows / 16) + ((lane * 8 + ((h >> 20) & 7)) % (kRows / 16));
        const uint32_t col = (h >> 4) % kXWords;
        float xv = 1.0f;
        if (kMode == kGather || kMode == kGatherAddF64 || kMode == kGatherRmwF32 || kMode == kGatherAddU64) xv = xs[col];
        switch (kMode) {
            case kAddF32: case kAddF32Ftz: atomicAdd(&acc32[row], 1.0f); break;
            case kAddRtnF32: keep += atomicAdd(&acc32[row], 1.0f); break;
            case kAddF64: atomicAdd(&acc64[row], 1.0); break;
            case kAddU64: atomicAdd(&accu64[row], 1ull); break;
            case kAddU32: atomicAdd(&accu32[row], 1u); break;
            case kRmwF32: { float a = acc32[prow]; a += 1.0f; acc32[prow] = a; break; }
            case kGather: keep += xv; break;
            case kGatherAddF64: atomicAdd(&acc64[row], static_cast<double>(xv + 1.0f)); break;
            case kGatherAddU64: atomicAdd(&accu64[row], static_cast<unsigned long long>(__float_as_uint(xv)) + 1ull); break;
            case kGatherRmwF32: { float a = acc32[prow]; a += xv + 1.0f; acc32[prow] = a; break; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = acc32[0] + keep;
    if (keep == 12345.678f) sink[blockIdx.x + 1] = keep;
}

template <int kMode>
void run(const char* name) {
    float* sink;
    hipMalloc(&sink, 1024 * sizeof(float));
    const uint32_t iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<kMode>), dim3(256), dim3(1024), 0, 0, iters, sink);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<kMode>), dim3(256), dim3(1024), 0, 0, iters, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double per_cu = double(iters) * 1024;   // element updates per CU
    printf("%-16s %8.1f us  %7.2f ns per wave-step (64 lanes)  %5.2f lanes/clk/CU @2.4GHz\n", name, best * 1e3, best * 1e6 / (per_cu / 64),
           per_cu / (best * 1e-3 * 2.4e9));
    hipFree(sink);
}

int main() {
    run<kAddU32>("add_u32");
    run<kAddU64>("add_u64");
    run<kAddF64>("add_f64");
    run<kAddF32>("add_f32");
    run<kAddRtnF32>("add_rtn_f32");
    run<kAddF32Ftz>("add_f32_ftz");
    run<kRmwF32>("rmw_f32");
    run<kGather>("gather");
    run<kGatherAddU64>("gather+add_u64");
    run<kGatherAddF64>("gather+add_f64");
    run<kGatherRmwF32>("gather+rmw_f32");
    return 0;
}
