// hs_api.cpp — implementation of the drop-in C-ABI (include/hisparse_hip.h) on the HIP runtime.
//
// The reference reaches its device through OpenCL/XRT objects created in sw/benchmark.cpp:228-298 and
// launches five kernels per row partition (:318-338).  Here one context owns one HIP device, one
// stream and the device-resident data; hs_run launches the whole SpMV (all row partitions) as ONE
// kernel, spmv_rowblock_kernel<fixed|float>, which writes the packed y directly.
// There is no CPU fallback anywhere in this file: without a usable gfx950 device every call fails.
#include "hisparse_hip.h"

#include <hip/hip_runtime_api.h>

#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "hisparse/channel_packets.h"
#include "hisparse/common.h"
#include "spmv_kernels.h"
#include "gpu_tiles.h"
#include "stream_tiles.h"
#include "tiles_common.h"

using hisparse::Geometry;
using hisparse::dev::Block;
using hisparse::dev::Unit;

struct hs_context {
    hisparse::dev::detail::OptionMap options;    // hs_set_option: "HISPARSE_<KEY>" -> value
    int device = -1;
    int impl = 0;
    Geometry geom;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int compute_units = 0;

    bool matrix_loaded = false;
    bool vector_loaded = false;
    uint32_t num_rows = 0, num_cols = 0, row_parts = 0, col_parts = 0;
    uint8_t* d_image = nullptr;
    Block* d_blocks = nullptr;
    Unit* d_units = nullptr;
    uint32_t* d_part_heads = nullptr;
    uint32_t num_workgroups = 0;
    uint32_t lds_bytes = 0;
    uint32_t bitmap_x_groups = 0;
    uint32_t col_slices = 1;
    uint32_t ring_buffers = 4;
    uint32_t format = 0;           // StreamFormat of d_image
    bool light = false;            // the LIGHT plan: d_image is a PAIRS image run by spmv_light_kernel (stream_tiles.h)
    uint32_t* d_partial = nullptr;  // col_slices > 1: per-slice partial results, col_slices x num_rows words
    // hs_run_batch with `batch_graph`: the captured step sequence, kept while nothing it bakes in changes
    hipGraph_t batch_graph = nullptr;
    hipGraphExec_t batch_exec = nullptr;
    uint32_t batch_steps = 0;
    const void* batch_x = nullptr;
    void* batch_y = nullptr;
    hipStream_t batch_stream = nullptr;
    // Column-sliced plans, hs_run after hs_run on the library's own stream: the combine pass of a step is CARRIED into the SpMV kernel of the
    // next one (spmv_device.h: CarriedCombine) -- enqueue() below.  `pending`: the set of partial vectors whose sum has not been written to
    // its y yet; flush_combine() launches the stand-alone combine for it, and every entry point that could observe y, its target or the
    // stream does that first, so the stream-order contract of hisparse_hip.h holds unchanged.
    bool stream_resident = false;   // plan-time decision (load_matrix_impl): SWEEP stream loads without the non-temporal hint (the image fits the Infinity Cache)
    bool carry_combine = false;     // plan-time decision (load_matrix_impl): two sets of partial vectors exist
    bool in_batch = false;          // inside hs_run_batch: the batch settles its own last step before it returns, whatever stream it runs on
    bool stream_shared = false;     // hs_get_stream was called: somebody else may order work against the stream -- every step completes in itself
    int pending = -1;
    uint32_t* pending_y = nullptr;
    uint32_t carry_turn = 0;
    bool crossing_blocks = false;   // some row block reaches over a row-partition border (tiles_common.h: Layout::cross_parts)
    uint32_t* d_partition_y = nullptr;   // hs_run_partition on a one-slice plan with such blocks: the kernel writes here (num_rows words,
                                         // allocated on first use), the partition's own rows are then copied into y
    uint32_t max_block_rows = 0;
    // SpMM over a SWEEP image planned for it (spmm_sweep.hip; option spmm_vectors = 4): X interleaved [column][4], the four result columns
    // (per column slice) before the combine pass, and after it
    uint32_t spmm_vectors = 1;
    uint32_t* d_spmm_x4 = nullptr;
    uint32_t* d_spmm_partial = nullptr;
    uint32_t* d_spmm_y = nullptr;
    uint32_t* d_x_interleaved = nullptr;   // fused SpMM over a BITMAP image: 4 columns of X as [column][vector] words (allocated on first use)
    // SpMM on the matrix engine (float BITMAP matrices): the second image + scratch (spmm_mfma.hip)
    uint32_t* d_mfma = nullptr;
    uint64_t mfma_bytes = 0;
    hisparse::dev::MfmaImage mfma_info;    // geometry only (words empty)
    uint32_t* d_mfma_x = nullptr;
    float* d_mfma_partial = nullptr;
    uint32_t* d_mfma_flag = nullptr;
    uint32_t mfma_call = 0;

    // SpMSpV extension: the matrix once more, in CSC form (hs_load_matrix_csc), + scratch
    uint32_t* d_csc_indptr = nullptr;
    uint32_t* d_csc_rows = nullptr;
    uint32_t* d_csc_vals = nullptr;
    hisparse::dev::SpmspvScratch csc_scratch;      // the product list (rows, product words, row blocks) and its counters
    std::vector<uint32_t> csc_col_len;             // host copy of the column lengths: splits a host-side x whose products exceed the list
    uint32_t* d_csc_y = nullptr;                   // max(csc rows, the dense matrix's padded rows) words
    uint32_t csc_y_words = 0;
    hisparse::dev::hs_idx_val_dev* d_sx = nullptr; // hs_spmspv: the caller's IDX_VAL_T pairs on the device ...
    hisparse::dev::hs_idx_val_dev* h_sx = nullptr; // ... = this pinned, mapped staging buffer (two halves of sx_capacity entries, used in turn)
    hipEvent_t sx_read[2] = {nullptr, nullptr};    // half h's kernels have read it: the host may write it again
    uint32_t sx_turn = 0;
    std::vector<uint8_t> sx_seen;                  // one bit per column, all zero between calls (the repeat check of hs_spmspv)
    uint32_t sx_capacity = 0;
    uint32_t* d_x_dense = nullptr;                 // dense dispatch: x scattered into a zero vector (num_cols words)
    uint32_t csc_rows = 0, csc_cols = 0;
    uint64_t csc_nnz = 0;
    double dense_spmv_us = 0.0;                    // the dense SpMV of the loaded matrix, timed once (hs_spmspv's dispatch rule); 0: not yet
    uint64_t spmspv_dense_dispatches = 0;          // calls of hs_spmspv answered by the dense SpMV (hs_get_stats does not carry it: tests read it through hs_last_error)

    uint32_t* d_x = nullptr;       // library-owned packed x
    uint32_t* d_y = nullptr;       // library-owned packed y
    const uint32_t* x_bound = nullptr;
    uint32_t* y_bound = nullptr;
    uint32_t x_capacity = 0;
    uint32_t x_len = 0;            // words of the vector last given to hs_load_vector

    hs_stats stats{};
    std::string error;
};

namespace {

thread_local std::string g_create_error;

constexpr size_t kImageSlackBytes = 16384;  // the clamped prefetches of a wavefront without chunks (offset w * 512 past a short block) stay inside the allocation

int fail(hs_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return code;
}
int hip_fail(hs_context* ctx, hipError_t e, const char* what) {
    return fail(ctx, HS_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HS_HIP(ctx, call)                                       \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call);  \
    } while (0)

// a call-time switch of this context: hs_set_option first, the environment as the fallback for tools
const char* ctx_option(const hs_context* c, const char* name) { return hisparse::dev::detail::option_lookup(&c->options, name); }

// hs_set_option's keys (the HISPARSE_<KEY> environment switches the library understands); plan-time ones take effect at the next load
const char* const kOptionKeys[] = {
    "STREAM_FORMAT", "COL_SLICES", "MAX_ROWS", "CROSS_PARTITIONS", "SPMM_VECTORS", "ROW_RUNS", "AUX_BITS", "XCD_AFFINITY", "STREAM_RESIDENT", "RETILE", "PLAN_DEBUG",
    "BITMAP_SKEW", "BITMAP_X_LDS", "BITMAP_BUILD", "WALK_LANES", "NO_MFMA_IMAGE", "MFMA_CHUNK", "LIGHT", "LIGHT_WGS", "SWEEP",
    "DELTA_DEAL", "POW2_SLICES", "SPMM_FUSED", "SPMM_MFMA", "SPMSPV", "SPMSPV_CROSSOVER", "ITERATE_GRAPH", "BATCH_GRAPH", "CARRY_COMBINE", "AUTOTUNE", "PLAN_CENSUS",
};

void drop_batch_graph(hs_context* c) {
    if (c->batch_exec) (void)hipGraphExecDestroy(c->batch_exec);
    if (c->batch_graph) (void)hipGraphDestroy(c->batch_graph);
    c->batch_exec = nullptr;
    c->batch_graph = nullptr;
    c->batch_steps = 0;
}

void free_matrix(hs_context* c) {
    drop_batch_graph(c);
    if (c->d_image) (void)hipFree(c->d_image);
    if (c->d_blocks) (void)hipFree(c->d_blocks);
    if (c->d_units) (void)hipFree(c->d_units);
    if (c->d_part_heads) (void)hipFree(c->d_part_heads);
    c->d_part_heads = nullptr;
    if (c->d_y) (void)hipFree(c->d_y);
    if (c->d_partial) (void)hipFree(c->d_partial);
    c->d_partial = nullptr;
    c->carry_combine = false;
    c->pending = -1;
    if (c->d_partition_y) (void)hipFree(c->d_partition_y);
    c->d_partition_y = nullptr;
    c->crossing_blocks = false;
    if (c->d_x_interleaved) (void)hipFree(c->d_x_interleaved);
    c->d_x_interleaved = nullptr;
    for (uint32_t** p : {&c->d_spmm_x4, &c->d_spmm_partial, &c->d_spmm_y}) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    c->spmm_vectors = 1;
    for (void* p : {static_cast<void*>(c->d_mfma), static_cast<void*>(c->d_mfma_x), static_cast<void*>(c->d_mfma_partial), static_cast<void*>(c->d_mfma_flag)})
        if (p) (void)hipFree(p);
    c->d_mfma = c->d_mfma_x = c->d_mfma_flag = nullptr;
    c->d_mfma_partial = nullptr;
    c->mfma_info = hisparse::dev::MfmaImage();
    c->d_image = nullptr;
    c->d_blocks = nullptr;
    c->d_units = nullptr;
    c->d_y = nullptr;
    c->y_bound = nullptr;
    c->matrix_loaded = false;
}

void free_csc(hs_context* c) {
    hisparse::dev::SpmspvScratch& w = c->csc_scratch;
    for (void* p : {static_cast<void*>(c->d_csc_indptr), static_cast<void*>(c->d_csc_rows), static_cast<void*>(c->d_csc_vals), static_cast<void*>(c->d_csc_y),
                    static_cast<void*>(w.keys), static_cast<void*>(w.vals), static_cast<void*>(w.bin_base), static_cast<void*>(w.cursors),
                    static_cast<void*>(w.overflow), static_cast<void*>(c->d_x_dense)})
        if (p) (void)hipFree(p);
    if (c->h_sx) (void)hipHostFree(c->h_sx);
    for (hipEvent_t& e : c->sx_read) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    c->h_sx = nullptr;
    c->d_csc_indptr = c->d_csc_rows = c->d_csc_vals = c->d_csc_y = c->d_x_dense = nullptr;
    c->d_sx = nullptr;
    w = hisparse::dev::SpmspvScratch();
    c->csc_col_len.clear();
    c->sx_capacity = 0;
    c->csc_y_words = 0;
    c->csc_rows = c->csc_cols = 0;
    c->csc_nnz = 0;
}

// The result of the last carried step, if any, goes to its y now (one combine_slices_kernel launch on the context's stream).
int flush_combine(hs_context* c) {
    if (c && c->pending >= 0) {
        const uint32_t* partial = c->d_partial + size_t(c->pending) * c->col_slices * c->num_rows;
        const hipError_t e = hisparse::dev::launch_combine_slices(c->impl != HS_IMPL_FIXED, partial, c->pending_y, c->num_rows, c->col_slices, 0, c->num_rows, c->stream);
        c->pending = -1;
        if (e != hipSuccess) return hip_fail(c, e, "combine_slices_kernel");
    }
    return HS_OK;
}
#define HS_FLUSH(ctx)                                  \
    do {                                               \
        if ((ctx) && (ctx)->pending >= 0) {            \
            (void)hipSetDevice((ctx)->device);         \
            const int rc_flush_ = flush_combine(ctx);  \
            if (rc_flush_ != HS_OK) return rc_flush_;  \
        }                                              \
    } while (0)

uint32_t* y_target(hs_context* c) { return c->y_bound ? c->y_bound : c->d_y; }
const uint32_t* x_source(hs_context* c) { return c->x_bound ? c->x_bound : c->d_x; }

int check_ready(hs_context* ctx) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    if (!ctx->vector_loaded && !ctx->x_bound) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_vector has not been called");
    if (!ctx->x_bound && ctx->x_len != ctx->num_cols)
        return fail(ctx, HS_ERR_NOT_LOADED, "the loaded vector does not have this matrix's padded column count: call hs_load_vector again");
    return HS_OK;
}

hisparse::dev::SpmvLaunch launch_args(hs_context* c, int32_t filter) {
    hisparse::dev::SpmvLaunch a;
    a.image = c->d_image;
    a.blocks = c->d_blocks;
    a.units = c->d_units;
    a.part_heads = c->d_part_heads;
    a.x = x_source(c);
    a.out = c->col_slices > 1 ? c->d_partial : y_target(c);
    a.row_part_filter = filter;
    a.ring_buffers = c->ring_buffers;
    a.format = c->format;
    a.num_cols = c->num_cols;
    a.num_workgroups = c->num_workgroups;
    a.lds_bytes = c->lds_bytes;
    a.bitmap_x_groups = c->bitmap_x_groups;
    a.light = c->light;
    a.stream_resident = c->stream_resident;
    return a;
}

// rows [lo, hi) of row partition j
void partition_rows(const hs_context* c, uint32_t j, uint32_t& lo, uint32_t& hi) {
    const uint64_t a = uint64_t(j) * c->geom.logical_ob;
    const uint64_t b = std::min<uint64_t>(a + c->geom.logical_ob, c->num_rows);
    lo = uint32_t(a);
    hi = uint32_t(b);
}

// Enqueue one SpMV (filter < 0) or one row partition; optional events bracket the kernel.  `feedback` (hs_iterate): also
// x = scale (*) y (+) shift afterwards -- folded into the combine launch of a column-sliced matrix, its own launch otherwise.
struct Feedback { uint32_t scale, shift; };
// x [num_cols words) and a result vector [num_rows words) share memory: the carried combine writes y(k) from inside step k+1's kernel while
// other workgroups of that kernel read x -- in-place y = A*y stays well defined only with the stand-alone combine
bool x_aliases(const hs_context* c, const uint32_t* x, const uint32_t* y) {
    if (!x || !y) return false;
    const uintptr_t x0 = reinterpret_cast<uintptr_t>(x), x1 = x0 + size_t(c->num_cols) * 4, y0 = reinterpret_cast<uintptr_t>(y), y1 = y0 + size_t(c->num_rows) * 4;
    return x0 < y1 && y0 < x1;
}
int enqueue(hs_context* c, int32_t filter, hipEvent_t k0, hipEvent_t k1, const Feedback* feedback = nullptr) {
    if (const char* why = hisparse::dev::profiling_switch_error()) return fail(c, HS_ERR_BAD_ARG, why);
    if (c->carry_combine && c->col_slices > 1 && filter < 0 && !k0 && !k1 && !feedback && ((c->stream == c->own_stream && !c->stream_shared) || c->in_batch) &&
        !x_aliases(c, x_source(c), y_target(c)) && !(c->pending >= 0 && x_aliases(c, x_source(c), c->pending_y))) {
        // this step's partial rows go to the set the previous step did NOT use; the previous step's are added up by this launch's
        // workgroups before they start on their blocks; this step's own sum is owed (pending) until the next hs_run or a flush
        hisparse::dev::SpmvLaunch a = launch_args(c, filter);
        const uint32_t b = c->carry_turn++ & 1u;
        const size_t set = size_t(c->col_slices) * c->num_rows;
        a.out = c->d_partial + b * set;
        if (c->pending >= 0) {
            a.carry_partial = c->d_partial + size_t(c->pending) * set;
            a.carry_y = c->pending_y;
            a.carry_rows = c->num_rows;
            a.carry_slices = c->col_slices;
        }
        HS_HIP(c, hisparse::dev::launch_spmv(c->impl != HS_IMPL_FIXED, a, c->stream));
        c->pending = int(b);
        c->pending_y = y_target(c);
        return HS_OK;
    }
    if (int rc = flush_combine(c)) return rc;
    hisparse::dev::SpmvLaunch args = launch_args(c, filter);
    // One partition of a plan whose row blocks reach over partition borders: the blocks that intersect the partition run (Block::next_part)
    // and compute rows of its neighbours too.  "Rows of other partitions keep their previous contents" (hisparse_hip.h): a column-sliced
    // plan combines the partition's rows only (below); a one-slice plan writes to a side buffer and the partition's rows are copied over.
    const bool side_y = filter >= 0 && c->crossing_blocks && c->col_slices == 1;
    if (side_y) {
        if (!c->d_partition_y) HS_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->d_partition_y), size_t(c->num_rows) * 4));
        args.out = c->d_partition_y;
    }
    if (k0) HS_HIP(c, hipEventRecord(k0, c->stream));
    HS_HIP(c, hisparse::dev::launch_spmv(c->impl != HS_IMPL_FIXED, args, c->stream));
    if (k1) HS_HIP(c, hipEventRecord(k1, c->stream));
    if (side_y) {
        uint32_t lo = 0, hi = 0;
        partition_rows(c, uint32_t(filter), lo, hi);
        HS_HIP(c, hipMemcpyAsync(y_target(c) + lo, c->d_partition_y + lo, size_t(hi - lo) * 4, hipMemcpyDeviceToDevice, c->stream));
    }
    const bool is_float = c->impl != HS_IMPL_FIXED;
    uint32_t* x = const_cast<uint32_t*>(x_source(c));
    const uint32_t n_fb = std::min(c->num_rows, c->num_cols);
    if (c->col_slices > 1) {
        uint32_t lo = 0, hi = c->num_rows;
        if (filter >= 0) partition_rows(c, uint32_t(filter), lo, hi);
        HS_HIP(c, hisparse::dev::launch_combine_slices(is_float, c->d_partial, y_target(c), c->num_rows, c->col_slices, lo, hi, c->stream,
                                                       feedback ? x : nullptr, n_fb, feedback ? feedback->scale : 0, feedback ? feedback->shift : 0));
    } else if (feedback) {
        HS_HIP(c, hisparse::dev::launch_feedback(is_float, y_target(c), x, n_fb, feedback->scale, feedback->shift, c->stream));
    }
    return HS_OK;
}

}  // namespace

extern "C" {

const char* hs_strerror(int code) {
    switch (code) {
        case HS_OK: return "ok";
        case HS_ERR_BAD_ARG: return "bad argument";
        case HS_ERR_NO_DEVICE: return "no usable gfx950 device";
        case HS_ERR_HIP: return "HIP runtime error";
        case HS_ERR_BAD_MATRIX: return "channel buffers are not a valid CPSR image";
        case HS_ERR_NOT_LOADED: return "matrix or vector not loaded";
        case HS_ERR_UNSUPPORTED: return "unsupported configuration";
        case HS_ERR_NO_MEMORY: return "out of memory";
        default: return "unknown error";
    }
}

const char* hs_last_error(const hs_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int hs_create(hs_context** out, int device_id, int impl, uint32_t ob_bank, uint32_t vb_bank) {
    if (!out) return fail(nullptr, HS_ERR_BAD_ARG, "null context pointer");
    *out = nullptr;
    if (!hisparse::impl_valid(impl)) return fail(nullptr, HS_ERR_BAD_ARG, "impl must be 0 (fixed), 1 (float_pob) or 2 (float_stall)");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(nullptr, HS_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= count) return fail(nullptr, HS_ERR_BAD_ARG, "device_id out of range");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, HS_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    hs_context* c = new (std::nothrow) hs_context;
    if (!c) return fail(nullptr, HS_ERR_NO_MEMORY, "out of memory");
    c->device = device_id;
    c->impl = impl;
    c->geom = hisparse::make_geometry(impl, ob_bank ? ob_bank : hisparse::impl_default_ob_bank(impl),
                                      vb_bank ? vb_bank : hisparse::impl_default_vb_bank(impl));
    c->compute_units = prop.multiProcessorCount;
    if (c->geom.logical_ob > 0xffffffffull || c->geom.logical_vb > 0xffffffffull || c->geom.logical_ob % c->geom.row_divisor != 0) {
        delete c;
        return fail(nullptr, HS_ERR_BAD_ARG, "ob_bank must make 128*ob_bank a multiple of 128*interleave; bank sizes must fit 32 bits");
    }
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) {
        delete c;
        return hip_fail(nullptr, e, "hipSetDevice/hipStreamCreate");
    }
    c->stream = c->own_stream;
    // "program the device" here, not inside the first hs_load_matrix: the code objects load on first use, and the dynamic-LDS cap
    // is a property of the FUNCTION (always the full 160 KiB, so that no context can lower it under another one's launches)
    if ((e = hisparse::dev::configure_spmv_kernels(hisparse::dev::kMaxLdsBytes)) != hipSuccess || (e = hisparse::dev::warm_gpu_tiler()) != hipSuccess) {
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return hip_fail(nullptr, e, "loading the gfx950 kernels");
    }
    // ... and the runtime's one-time set-up of its pageable-copy paths (7 ms in the first copy of a process, whatever its size:
    // tools/h2d_bench.cpp), once per process
    static std::once_flag copy_paths;
    std::call_once(copy_paths, [c] {
        std::vector<uint8_t> host(4 << 20, 0);
        void* dev = nullptr;
        if (hipMalloc(&dev, host.size()) != hipSuccess) return;
        (void)hipMemcpy(dev, host.data(), host.size(), hipMemcpyHostToDevice);
        (void)hipMemcpy(host.data(), dev, host.size(), hipMemcpyDeviceToHost);
        (void)hipMemcpyAsync(dev, host.data(), host.size(), hipMemcpyHostToDevice, c->own_stream);      // the stream-ordered variants
        (void)hipMemcpyAsync(host.data(), dev, host.size(), hipMemcpyDeviceToHost, c->own_stream);      // have their own set-up
        (void)hipStreamSynchronize(c->own_stream);
        (void)hipFree(dev);
    });
    *out = c;
    return HS_OK;
}

int hs_destroy(hs_context* ctx) {
    if (!ctx) return HS_OK;
    (void)hipSetDevice(ctx->device);
    // only the library's own stream is known to be alive here; a caller-owned stream (hs_set_stream) must have been
    // synchronised by its owner before the context is destroyed (hisparse_hip.h)
    if (ctx->stream == ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    else (void)hipDeviceSynchronize();
    free_matrix(ctx);
    free_csc(ctx);
    if (ctx->d_x) (void)hipFree(ctx->d_x);
    drop_batch_graph(ctx);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return HS_OK;
}

namespace {
// hs_load_matrix (CPSR channel buffers) and hs_load_matrix_csr (`csr` != nullptr, channel / n_packets null) behind one body
int load_matrix_once(hs_context* ctx, const void* const* channel, const uint64_t* n_packets, const hisparse::dev::CsrView* csr,
                     uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions, uint32_t num_col_partitions) {
    const Geometry& g = ctx->geom;
    if (num_rows == 0 || num_cols == 0) return fail(ctx, HS_ERR_BAD_ARG, "empty matrix");
    if (num_rows % g.row_divisor != 0 || num_cols % hisparse::PACK_SIZE != 0)
        return fail(ctx, HS_ERR_BAD_ARG, "dimensions are not padded: rows must divide by " + std::to_string(g.row_divisor) +
                                             " and columns by 8 (util_round_csr_matrix_dim)");
    if (num_row_partitions != (num_rows + g.logical_ob - 1) / g.logical_ob || num_col_partitions != (num_cols + g.logical_vb - 1) / g.logical_vb)
        return fail(ctx, HS_ERR_BAD_ARG, "partition counts do not match the dimensions and the bank sizes of this context");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_matrix(ctx);
    const auto t0 = std::chrono::steady_clock::now();

    hisparse::dev::StreamTiles tiles;
    std::string why;
    auto drop_device_images = [](hisparse::dev::StreamTiles& t) {      // what a builder left on the device for a load that fails after it
        if (t.d_image) (void)hipFree(t.d_image);
        if (t.mfma.d_words) (void)hipFree(t.mfma.d_words);
        t.d_image = nullptr;
        t.mfma.d_words = nullptr;
    };
    // The per-non-zero passes of the re-tiling run on the GPU (gpu_tiles.h) unless HISPARSE_RETILE=host; BITMAP images and matrices
    // with duplicate entries are built by the host code, which also remains the byte-for-byte checker of the GPU path.
    const hisparse::dev::detail::OptionScope option_scope(&ctx->options);      // this context's hs_set_option values rule the planning below
    const char* retile = hisparse::dev::detail::env_switch("HISPARSE_RETILE");
    bool on_gpu = csr || !(retile && std::string(retile) == "host");
    try {
        // one 1024-thread workgroup per CU: its row accumulators and x ring fill the 160 KiB LDS
        bool ok = hisparse::dev::build_stream_tiles(channel, n_packets, g, num_rows, num_cols, num_row_partitions, num_col_partitions,
                                                    uint32_t(ctx->compute_units), tiles, why, ctx->stream, on_gpu, kImageSlackBytes, csr);
        if (!ok && csr && why == "gpu re-tile: duplicate entries") {
            // A (row, column) that occurs twice is legal input for the reference's formatter (csr2cpsr keeps both entries and the PEs add
            // both products), but the device sort has no defined order among equal positions.  Do what a reference driver does instead:
            // format on the host (sw/benchmark.cpp:110-195) and hand the CPSR buffers to the host builder, like hs_load_matrix does for
            // such a matrix.
            drop_device_images(tiles);
            tiles = hisparse::dev::StreamTiles();
            spmv::io::CSRMatrix<float> m;
            m.num_rows = csr->num_rows;
            m.num_cols = csr->num_cols;
            const uint64_t nnz = csr->indptr[csr->num_rows];
            m.adj_indptr.assign(csr->indptr, csr->indptr + csr->num_rows + 1);
            m.adj_indices.assign(csr->indices, csr->indices + nnz);
            m.adj_data.assign(csr->values, csr->values + nnz);
            const hisparse::ChannelPackets packets = hisparse::format_matrix(m, g, /*skip_empty_rows=*/true);
            const void* chan[hisparse::NUM_HBM_CHANNELS];
            uint64_t count[hisparse::NUM_HBM_CHANNELS];
            for (uint32_t c = 0; c < hisparse::NUM_HBM_CHANNELS; ++c) { chan[c] = packets.channel[c].data(); count[c] = packets.channel[c].size(); }
            on_gpu = false;
            ok = packets.num_rows == num_rows && packets.num_cols == num_cols &&
                 hisparse::dev::build_stream_tiles(chan, count, g, num_rows, num_cols, num_row_partitions, num_col_partitions, uint32_t(ctx->compute_units), tiles, why);
        }
        if (!ok && !csr && on_gpu && why.rfind("gpu re-tile:", 0) == 0) {       // duplicates, or a HIP failure on the way: the host path decides
            drop_device_images(tiles);
            tiles = hisparse::dev::StreamTiles();
            on_gpu = false;
            ok = hisparse::dev::build_stream_tiles(channel, n_packets, g, num_rows, num_cols, num_row_partitions, num_col_partitions,
                                                   uint32_t(ctx->compute_units), tiles, why);
        }
        if (!ok) return fail(ctx, HS_ERR_BAD_MATRIX, why);
    } catch (const std::bad_alloc&) {
        drop_device_images(tiles);
        return fail(ctx, HS_ERR_NO_MEMORY, "out of host memory while re-tiling the matrix");
    } catch (const std::exception& e) {      // whatever a builder task threw (WorkerPool rethrows it): never through the C ABI
        drop_device_images(tiles);
        return fail(ctx, HS_ERR_BAD_MATRIX, std::string("re-tiling the matrix failed: ") + e.what());
    } catch (...) {
        drop_device_images(tiles);
        return fail(ctx, HS_ERR_BAD_MATRIX, "re-tiling the matrix failed");
    }
    const bool debug = ctx_option(ctx, "HISPARSE_PLAN_DEBUG") != nullptr;
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    if (debug) std::fprintf(stderr, "load: image built after %.1f ms\n", since());
    // (BITMAP: + the block's stretch of x behind the accumulators when the builder asks for it, spmv_bitmap.hip kXLds)
    const uint32_t lds_bytes = tiles.format == hisparse::dev::kFormatSweep ? hisparse::dev::spmv_sweep_lds_bytes(tiles.max_block_rows, ctx->impl != HS_IMPL_FIXED)
                               : tiles.light ? hisparse::dev::spmv_light_lds_bytes(tiles.max_block_rows)
                                           : hisparse::dev::spmv_lds_bytes(tiles.max_block_rows, tiles.ring_buffers, tiles.format) +
                                                 tiles.bitmap_x_groups * hisparse::dev::kBitmapGroupCols * 4u;
    if (lds_bytes > hisparse::dev::kMaxLdsBytes) {
        drop_device_images(tiles);
        return fail(ctx, HS_ERR_UNSUPPORTED, "row block does not fit the LDS");
    }

    // What the builder left on the device belongs to the context from here on: an early return below then leaks nothing --
    // free_matrix (the next load, hs_destroy) gives it back; matrix_loaded stays false until the end.
    if (tiles.d_image) { ctx->d_image = tiles.d_image; }
    if (tiles.mfma.d_words) { ctx->d_mfma = reinterpret_cast<uint32_t*>(tiles.mfma.d_words); }
    const bool image_on_device = tiles.d_image != nullptr, mfma_on_device = tiles.mfma.d_words != nullptr;
    tiles.d_image = nullptr;
    tiles.mfma.d_words = nullptr;
    // the dynamic-LDS cap is a property of the FUNCTION, not of this context: always raise it to the full 160 KiB, so that a
    // second context with a smaller matrix on the same device cannot lower it under a first one's launches
    HS_HIP(ctx, hisparse::dev::configure_spmv_kernels(hisparse::dev::kMaxLdsBytes));
    auto upload = [&](void** dst, const void* src, size_t bytes, size_t slack) -> hipError_t {
        hipError_t e = hipMalloc(dst, std::max<size_t>(bytes + slack, 256));
        if (e != hipSuccess || bytes == 0) return e;
        return hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    };
    if (!image_on_device)      // (built on the device: adopted above, slack included)
        HS_HIP(ctx, upload(reinterpret_cast<void**>(&ctx->d_image), tiles.image.data(), tiles.image.size(), kImageSlackBytes));
    HS_HIP(ctx, upload(reinterpret_cast<void**>(&ctx->d_blocks), tiles.blocks.data(), tiles.blocks.size() * sizeof(Block), 0));
    HS_HIP(ctx, upload(reinterpret_cast<void**>(&ctx->d_units), tiles.units.data(), tiles.units.size() * sizeof(Unit), 0));
    HS_HIP(ctx, upload(reinterpret_cast<void**>(&ctx->d_part_heads), tiles.part_heads.data(), tiles.part_heads.size() * sizeof(uint32_t), 0));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_y), size_t(num_rows) * 4));
    HS_HIP(ctx, hipMemset(ctx->d_y, 0, size_t(num_rows) * 4));  // the host zero-initialises y (sw/benchmark.cpp:217-222)
    {
        // SWEEP images that fit the 256 MiB Infinity Cache are streamed WITHOUT the non-temporal hint: repeated SpMVs of one matrix -- the
        // reference's benchmark loop, an iterative caller -- then read most of the image from the cache (profiles/r05_sweep_stream_policy.txt: one
        // rank's slab of ogbn-products split 8 ways, 124 MB: 39.5 -> 32.0 us; pokec, 247 MB: 61.1 -> 59.6-60.8 fixed, 69.3-70.3 -> 66.3-67.7 float_pob).
        // Larger images keep `nt` (a plain read loop over 1 GiB: 7.1 TB/s with it, 6.0 without: profiles/r02_hbm_read_bench.txt).  `stream_resident` = 0 | 1 decides otherwise.
        // Round 6: the row-block kernels' PAIRS / DELTA streams too (spmv_kernels.hip: Ring<kRing | 4>), by their own rule (stream_tiles.h:
        // kRowblockResidentMaxImageBytes): up to the size of the cache where the blocks walk several units, tiny images whatever their shape; pure one-unit
        // streams and everything larger keep `nt` (hollywood: +13 % without it).  OWNER / OWNER24 / BITMAP / LIGHT images are not affected.
        const char* opt = ctx_option(ctx, "HISPARSE_STREAM_RESIDENT");
        const uint64_t image_bytes = image_on_device ? tiles.image_bytes : uint64_t(tiles.image.size());
        const bool rowblock_stream = (tiles.format == hisparse::dev::kFormatPairs || tiles.format == hisparse::dev::kFormatDelta) && !tiles.light;
        ctx->stream_resident = opt ? std::atoi(opt) != 0
                               : rowblock_stream ? image_bytes <= hisparse::dev::kRowblockResidentMaxImageBytes &&
                                                       (tiles.units.size() > tiles.blocks.size() || image_bytes <= hisparse::dev::kRowblockResidentSmallImageBytes)
                                                 : image_bytes <= hisparse::dev::kResidentMaxImageBytes;
    }
    if (tiles.col_slices > 1) {
        // the combine pass carried into the next step's kernel (hs_context::carry_combine); `carry_combine` = 0 | 1 decides otherwise
        // Measured (profiles/r05_carry_combine_ab.txt, three boxes, whole step): where a step is a few microseconds -- one rank's slab of
        // mouse_gene split 8 ways: 10.1 -> 8.6 us, the second launch WAS a third of it -- carrying wins every time.  On the large images it
        // is a wash that depends on the box and the run (ogbl-ppa 55.4 -> 53.5 / 54.2 / 57.0 us, hollywood 137.0 -> 135.1 / 139.1, the R-MAT
        // stand-in 60.0 -> 57.8 / 62.0, ogbn-products 206 -> 204; pokec's SWEEP kernel 73.5 -> 76.0: 33 MB of partial rows in front of every
        // launch): the partial rows a workgroup adds up were written by OTHER XCDs and come back from the memory side while nothing else of
        // the workgroup can start.  Hence: on by itself for images below 160 MiB (48 MiB until the middle was measured in round 6: stream_tiles.h, kCarryMaxImageBytes), the launch-bound regime, and for OWNER images; `carry_combine` = 0 | 1 decides otherwise.
        const char* opt = ctx_option(ctx, "HISPARSE_CARRY_COMBINE");
        const uint64_t image_bytes = image_on_device ? tiles.image_bytes : uint64_t(tiles.image.size());
        // (OWNER / OWNER24 images of any size too: ogbn-products gained 1-1.5 % in every one of four A/B pairs on two boxes (profiles/r05_carry_combine_ab.txt) -- its workgroups
        // run two blocks each and the carried rows ride on the first block's long prologue)
        const bool owner_image = tiles.format == hisparse::dev::kFormatOwner || tiles.format == hisparse::dev::kFormatOwner24;
        const bool carry = opt ? std::atoi(opt) != 0 : (image_bytes < hisparse::dev::kCarryMaxImageBytes || owner_image);
        HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_partial), size_t(carry ? 2 : 1) * tiles.col_slices * num_rows * 4));
        ctx->carry_combine = carry;
    }
    if (mfma_on_device || (tiles.mfma.words_bytes != 0 && !tiles.mfma.words.empty())) {      // float BITMAP matrix: the second image for the SpMM on the matrix engine + its scratch
        // OPTIONAL: SpMV works without it.  If the image or its scratch cannot be had (out of memory), the matrix loads without a second
        // image and hs_spmm takes the fused 4-column kernel instead.
        const hisparse::dev::MfmaImage& mi = tiles.mfma;
        bool ok = mfma_on_device || upload(reinterpret_cast<void**>(&ctx->d_mfma), mi.words.data(), mi.words.size(), 0) == hipSuccess;
        ok = ok && hipMalloc(reinterpret_cast<void**>(&ctx->d_mfma_x), hisparse::dev::spmm_mfma_x_words(mi.groups) * 4) == hipSuccess &&
             hipMalloc(reinterpret_cast<void**>(&ctx->d_mfma_partial), hisparse::dev::spmm_mfma_partial_words(mi.tiles, mi.chunks) * 4) == hipSuccess &&
             hipMalloc(reinterpret_cast<void**>(&ctx->d_mfma_flag), 64) == hipSuccess && hipMemset(ctx->d_mfma_flag, 0, 64) == hipSuccess;
        if (ok) {
            ctx->mfma_bytes = mi.words_bytes;
            ctx->mfma_info.tiles = mi.tiles; ctx->mfma_info.groups = mi.groups; ctx->mfma_info.chunk = mi.chunk; ctx->mfma_info.chunks = mi.chunks;
            ctx->mfma_info.offsets_word = mi.offsets_word; ctx->mfma_info.values_word = mi.values_word;
        } else {
            (void)hipGetLastError();
            for (void* p : {static_cast<void*>(ctx->d_mfma), static_cast<void*>(ctx->d_mfma_x), static_cast<void*>(ctx->d_mfma_partial), static_cast<void*>(ctx->d_mfma_flag)})
                if (p) (void)hipFree(p);
            ctx->d_mfma = ctx->d_mfma_x = ctx->d_mfma_flag = nullptr;
            ctx->d_mfma_partial = nullptr;
        }
    }
    if (debug) std::fprintf(stderr, "load: descriptors + result buffers on the device after %.1f ms\n", since());
    ctx->num_rows = num_rows;
    ctx->num_cols = num_cols;
    ctx->row_parts = num_row_partitions;
    ctx->col_parts = num_col_partitions;
    ctx->num_workgroups = tiles.num_workgroups;
    ctx->lds_bytes = lds_bytes;
    ctx->bitmap_x_groups = tiles.bitmap_x_groups;
    ctx->col_slices = tiles.col_slices;
    ctx->spmm_vectors = tiles.spmm_vectors;
    if (tiles.spmm_vectors == 4) {
        const uint32_t need = hisparse::dev::spmm_sweep_lds_bytes(tiles.max_block_rows, ctx->impl != HS_IMPL_FIXED);
        if (need > hisparse::dev::kMaxLdsBytes) ctx->spmm_vectors = 1;      // (cannot happen with the planner's row cap; the k-SpMV path then)
        else HS_HIP(ctx, hisparse::dev::configure_spmm_sweep_kernels(hisparse::dev::kMaxLdsBytes));
    }
    for (const Block& b : tiles.blocks) ctx->crossing_blocks = ctx->crossing_blocks || b.last_part != b.row_part;
    ctx->max_block_rows = tiles.max_block_rows;
    ctx->ring_buffers = tiles.ring_buffers;
    ctx->format = tiles.format;
    ctx->light = tiles.light;
    ctx->dense_spmv_us = 0.0;
    ctx->matrix_loaded = true;

    hs_stats& s = ctx->stats;
    s = hs_stats{};
    s.nnz = tiles.nnz;
    for (int c = 0; n_packets && c < HS_NUM_CHANNELS; ++c) s.cpsr_bytes += n_packets[c] * sizeof(hisparse::MatPkt);
    s.stream_bytes = tiles.image_bytes;
    s.stream_elements = tiles.elements;
    s.num_blocks = uint32_t(tiles.blocks.size());
    s.num_units = uint32_t(tiles.units.size());
    s.col_slices = tiles.col_slices;
    s.ring_buffers = tiles.ring_buffers;
    s.stream_format = tiles.format;
    s.num_workgroups = tiles.num_workgroups;
    s.lds_bytes = lds_bytes;
    s.num_compute_units = uint32_t(ctx->compute_units);
    s.load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    s.retiled_on_gpu = image_on_device;
    s.light_kernel = tiles.light ? 1u : 0u;
    s.stream_resident = ctx->stream_resident && (tiles.format == hisparse::dev::kFormatSweep || ((tiles.format == hisparse::dev::kFormatPairs || tiles.format == hisparse::dev::kFormatDelta) && !tiles.light)) ? 1u : 0u;
    return HS_OK;
}
// EXTENSION, opt-in (hs_set_option "autotune" = 1): the plan by MEASUREMENT.  The planner's model is within 10 % of the best plan that can be forced on 23 of 24
// + 9 of 12 out-of-sample matrices (tools/planner_check.py); what is left are close calls no statistic it has separates (a fixed-point one-slice plan that OWNER24
// would run 1.3 x faster next to others of the same shape it would slow down; hollywood: OWNER24 3-6 % ahead of the DELTA image the gap rule picks).  With the
// option set the load builds the planner's own image, times a few SpMVs of it on a zero vector (the step time does not depend on the values), does the same for
// every other element format the matrix can take, and keeps the fastest -- a caller that will run thousands of SpMVs of one matrix trades a few more loads
// (each tens of milliseconds) for it.  The reference's analogue is its design-space sweep (performance_model/design_space_exp.cpp:496-547), done there by a
// model because a bitstream cannot be rebuilt per matrix; an image can.
double time_loaded_plan(hs_context* ctx, int runs) {
    uint32_t* zero_x = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&zero_x), size_t(ctx->num_cols) * 4 + 64) != hipSuccess) { (void)hipGetLastError(); return -1.0; }
    double us = -1.0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    const uint32_t* saved_x = ctx->x_bound;
    uint32_t* saved_y = ctx->y_bound;
    ctx->x_bound = zero_x;
    ctx->y_bound = nullptr;
    if (hipMemsetAsync(zero_x, 0, size_t(ctx->num_cols) * 4, ctx->stream) == hipSuccess && hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess) {
        int rc = HS_OK;
        for (int i = 0; i < 3 && rc == HS_OK; ++i) rc = enqueue(ctx, -1, nullptr, nullptr);
        for (int rep = 0; rep < 2 && rc == HS_OK; ++rep) {      // best of two regions
            (void)hipEventRecord(t0, ctx->stream);
            for (int i = 0; i < runs && rc == HS_OK; ++i) rc = enqueue(ctx, -1, nullptr, nullptr);
            if (rc == HS_OK) rc = flush_combine(ctx);
            (void)hipEventRecord(t1, ctx->stream);
            float ms = 0.0f;
            if (rc == HS_OK && hipEventSynchronize(t1) == hipSuccess && hipEventElapsedTime(&ms, t0, t1) == hipSuccess) {
                const double one = double(ms) * 1000.0 / runs;
                us = us < 0.0 ? one : std::min(us, one);
            }
        }
        if (rc != HS_OK) { ctx->pending = -1; us = -1.0; }
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    ctx->x_bound = saved_x;
    ctx->y_bound = saved_y;
    (void)hipFree(zero_x);
    return us;
}

int load_matrix_impl(hs_context* ctx, const void* const* channel, const uint64_t* n_packets, const hisparse::dev::CsrView* csr,
                     uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions, uint32_t num_col_partitions) {
    int rc = load_matrix_once(ctx, channel, n_packets, csr, num_rows, num_cols, num_row_partitions, num_col_partitions);
    const char* tune = ctx_option(ctx, "HISPARSE_AUTOTUNE");
    if (rc != HS_OK || !(tune && std::atoi(tune) != 0) || ctx_option(ctx, "HISPARSE_STREAM_FORMAT")) return rc;      // (a forced format is the caller's decision)
    const bool debug = ctx_option(ctx, "HISPARSE_PLAN_DEBUG") != nullptr;
    const char* const names[] = {"pairs", "delta", "bitmap", "owner", "pairs24", "owner24", "sweep"};      // StreamFormat order (stream_tiles.h)
    const std::string own = ctx->light ? "light" : names[ctx->format < 7 ? ctx->format : 0];
    const uint64_t nnz = ctx->stats.nnz;
    const int runs = int(std::max<uint64_t>(5, std::min<uint64_t>(50, (uint64_t(40) << 20) / std::max<uint64_t>(1, nnz))));      // ~ 1-3 ms of SpMVs per candidate
    double best_us = time_loaded_plan(ctx, runs);
    if (best_us <= 0.0) return HS_OK;                       // could not time: the planner's plan stands
    std::string best = own;
    if (debug) std::fprintf(stderr, "autotune: planner's plan %s x%u: %.2f us\n", own.c_str(), ctx->col_slices, best_us);
    const double own_us = best_us;
    const auto light_it = ctx->options.find("HISPARSE_LIGHT");      // the caller's own setting, put back at the end
    const bool had_light = light_it != ctx->options.end();
    const std::string caller_light = had_light ? light_it->second : std::string();
    auto restore = [&]() {
        ctx->options.erase("HISPARSE_STREAM_FORMAT");
        if (had_light) ctx->options["HISPARSE_LIGHT"] = caller_light; else ctx->options.erase("HISPARSE_LIGHT");
    };
    for (const char* fmt : {"delta", "pairs", "owner24", "sweep"}) {
        if (own == fmt) continue;
        ctx->options["HISPARSE_STREAM_FORMAT"] = fmt;
        ctx->options["HISPARSE_LIGHT"] = "0";
        const int rc2 = load_matrix_once(ctx, channel, n_packets, csr, num_rows, num_cols, num_row_partitions, num_col_partitions);
        double us = -1.0;
        if (rc2 == HS_OK && std::string(names[ctx->format < 7 ? ctx->format : 0]) == fmt) us = time_loaded_plan(ctx, runs);
        if (debug) std::fprintf(stderr, "autotune: %s x%u: %s\n", fmt, rc2 == HS_OK ? ctx->col_slices : 0u, us > 0.0 ? (std::to_string(us) + " us").c_str() : "not available");
        if (us > 0.0 && us < 0.97 * best_us) { best_us = us; best = fmt; }      // (3 %: below that it is the box's noise, and the planner's plan wins ties)
    }
    restore();
    if (best != own) {
        ctx->options["HISPARSE_STREAM_FORMAT"] = best;
        ctx->options["HISPARSE_LIGHT"] = "0";
    }
    rc = load_matrix_once(ctx, channel, n_packets, csr, num_rows, num_cols, num_row_partitions, num_col_partitions);      // the winner (or the planner's own plan again)
    restore();
    if (debug) std::fprintf(stderr, "autotune: kept %s (%.2f us against the planner's %.2f)\n", best.c_str(), best_us, own_us);
    return rc;
}
}  // namespace

int hs_load_matrix(hs_context* ctx, const void* const channel[HS_NUM_CHANNELS], const uint64_t n_packets[HS_NUM_CHANNELS],
                   uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions, uint32_t num_col_partitions) {
    if (!ctx || !channel || !n_packets) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    return load_matrix_impl(ctx, channel, n_packets, nullptr, num_rows, num_cols, num_row_partitions, num_col_partitions);
}

int hs_load_matrix_csr(hs_context* ctx, uint32_t num_rows, uint32_t num_cols, const uint32_t* indptr, const uint32_t* indices, const float* values,
                       uint32_t* padded_rows, uint32_t* padded_cols) {
    if (!ctx || !indptr) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    const Geometry& g = ctx->geom;
    if (num_rows == 0 || num_cols == 0) return fail(ctx, HS_ERR_BAD_ARG, "empty matrix");
    // util_round_csr_matrix_dim (sw/data_formatter.h:15-29): rows up to a multiple of P*C*F, columns to a multiple of 8
    const uint64_t rows = (uint64_t(num_rows) + g.row_divisor - 1) / g.row_divisor * g.row_divisor;
    const uint64_t cols = (uint64_t(num_cols) + hisparse::PACK_SIZE - 1) / hisparse::PACK_SIZE * hisparse::PACK_SIZE;
    if (rows > 0xffffffffull || cols > 0xffffffffull) return fail(ctx, HS_ERR_BAD_ARG, "padded dimensions exceed 32 bits");
    hisparse::dev::CsrView view;
    view.num_rows = num_rows;
    view.num_cols = num_cols;
    view.indptr = indptr;
    view.indices = indices;
    view.values = values;
    const int rc = load_matrix_impl(ctx, nullptr, nullptr, &view, uint32_t(rows), uint32_t(cols), uint32_t((rows + g.logical_ob - 1) / g.logical_ob),
                                    uint32_t((cols + g.logical_vb - 1) / g.logical_vb));
    if (rc == HS_OK) {
        if (padded_rows) *padded_rows = uint32_t(rows);
        if (padded_cols) *padded_cols = uint32_t(cols);
    }
    return rc;
}

int hs_debug_read_tiles(hs_context* ctx, void* image, uint64_t image_capacity, void* blocks, void* units) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    const hs_stats& s = ctx->stats;
    if (image && image_capacity < s.stream_bytes) return fail(ctx, HS_ERR_BAD_ARG, "image buffer too small");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (image && s.stream_bytes) HS_HIP(ctx, hipMemcpy(image, ctx->d_image, s.stream_bytes, hipMemcpyDeviceToHost));
    if (blocks && s.num_blocks) HS_HIP(ctx, hipMemcpy(blocks, ctx->d_blocks, size_t(s.num_blocks) * sizeof(Block), hipMemcpyDeviceToHost));
    if (units && s.num_units) HS_HIP(ctx, hipMemcpy(units, ctx->d_units, size_t(s.num_units) * sizeof(Unit), hipMemcpyDeviceToHost));
    return HS_OK;
}

int hs_debug_read_mfma_image(hs_context* ctx, void* words, uint64_t capacity, uint64_t* bytes) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    const uint64_t n = ctx->d_mfma ? ctx->mfma_bytes : 0;
    if (bytes) *bytes = n;
    if (!words || !n) return HS_OK;
    if (capacity < n) return fail(ctx, HS_ERR_BAD_ARG, "buffer too small");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    HS_HIP(ctx, hipMemcpy(words, ctx->d_mfma, n, hipMemcpyDeviceToHost));
    return HS_OK;
}

int hs_load_vector(hs_context* ctx, const void* packed_x, uint32_t num_cols) {
    if (!ctx || !packed_x) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (ctx->matrix_loaded && num_cols != ctx->num_cols) return fail(ctx, HS_ERR_BAD_ARG, "vector length must equal the padded column count");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    if (num_cols > ctx->x_capacity) {
        HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_x) (void)hipFree(ctx->d_x);
        ctx->d_x = nullptr;
        ctx->x_capacity = 0;
        HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_x), size_t(num_cols) * 4 + 64));
        ctx->x_capacity = num_cols;
    }
    HS_HIP(ctx, hipMemcpyAsync(ctx->d_x, packed_x, size_t(num_cols) * 4, hipMemcpyHostToDevice, ctx->stream));
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller may reuse packed_x immediately
    ctx->vector_loaded = true;
    ctx->x_len = num_cols;
    return HS_OK;
}

int hs_run(hs_context* ctx) {
    int rc = check_ready(ctx);
    if (rc != HS_OK) return rc;
    HS_HIP(ctx, hipSetDevice(ctx->device));
    return enqueue(ctx, -1, nullptr, nullptr);
}

int hs_run_batch(hs_context* ctx, uint32_t steps) {
    int rc = check_ready(ctx);
    if (rc != HS_OK) return rc;
    if (steps == 0) return HS_OK;
    HS_HIP(ctx, hipSetDevice(ctx->device));
    const char* opt = ctx_option(ctx, "HISPARSE_BATCH_GRAPH");
    // A batch is one unit in stream order: inside it the steps carry each other's combine pass (enqueue), and the last step's is launched
    // before the call returns -- also on a caller-owned stream, where single hs_run calls must each complete in themselves.
    struct InBatch {
        hs_context* c;
        const bool own;
        explicit InBatch(hs_context* ctx) : c(ctx), own(ctx->stream == ctx->own_stream && !ctx->stream_shared) { c->in_batch = true; }
        ~InBatch() { c->in_batch = false; }
        int settle() { return own ? HS_OK : flush_combine(c); }      // (on the library's own stream the sum may stay owed: every entry point settles it)
    } batch(ctx);
    if (!(opt && std::atoi(opt) != 0)) {      // plain: the launches of `steps` SpMVs enqueued from this C loop
        for (uint32_t i = 0; i < steps; ++i)
            if ((rc = enqueue(ctx, -1, nullptr, nullptr)) != HS_OK) {
                (void)batch.settle();      // a caller-owned stream is never left owing a sum, also not on the error path
                return rc;
            }
        return batch.settle();
    }
    // graph replay: the same launches captured once into a hipGraph (per step count, vector, result target and stream) and replayed
    // with ONE runtime call -- what the step costs when the host's enqueue rate is out of the picture
    if (const char* why = hisparse::dev::profiling_switch_error()) return fail(ctx, HS_ERR_BAD_ARG, why);
    if (!ctx->batch_exec || ctx->batch_steps != steps || ctx->batch_x != x_source(ctx) || ctx->batch_y != y_target(ctx) || ctx->batch_stream != ctx->stream) {
        drop_batch_graph(ctx);
        HS_FLUSH(ctx);                                       // the graph owes nothing when it begins ...
        hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) return hip_fail(ctx, e, "hipStreamBeginCapture (the legacy default stream cannot be captured)");
        for (uint32_t i = 0; i < steps && rc == HS_OK; ++i) rc = enqueue(ctx, -1, nullptr, nullptr);
        if (rc == HS_OK) rc = flush_combine(ctx);            // ... and nothing when it ends (a carried plan: K kernels + one combine)
        e = hipStreamEndCapture(ctx->stream, &ctx->batch_graph);
        // nothing was EXECUTED during the capture: whatever the captured steps recorded as owed does not exist (a dropped graph must not
        // leave a stale `pending` for the next entry point's flush to combine)
        if (rc != HS_OK) { ctx->pending = -1; drop_batch_graph(ctx); return rc; }
        if (e != hipSuccess || !ctx->batch_graph) { ctx->pending = -1; drop_batch_graph(ctx); return hip_fail(ctx, e, "hipStreamEndCapture"); }
        e = hipGraphInstantiate(&ctx->batch_exec, ctx->batch_graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { ctx->pending = -1; drop_batch_graph(ctx); return hip_fail(ctx, e, "hipGraphInstantiate"); }
        ctx->batch_steps = steps;
        ctx->batch_x = x_source(ctx);
        ctx->batch_y = y_target(ctx);
        ctx->batch_stream = ctx->stream;
    }
    HS_FLUSH(ctx);
    HS_HIP(ctx, hipGraphLaunch(ctx->batch_exec, ctx->stream));
    return HS_OK;
}

int hs_run_partition(hs_context* ctx, uint32_t row_part_id, uint32_t part_len) {
    int rc = check_ready(ctx);
    if (rc != HS_OK) return rc;
    if (row_part_id >= ctx->row_parts) return fail(ctx, HS_ERR_BAD_ARG, "row_part_id out of range");
    uint32_t lo, hi;
    partition_rows(ctx, row_part_id, lo, hi);
    if (part_len != (hi - lo) / hisparse::NUM_HBM_CHANNELS)
        return fail(ctx, HS_ERR_BAD_ARG, "part_len must be the partition's rows / 16 (sw/benchmark.cpp:301-322): expected " +
                                             std::to_string((hi - lo) / hisparse::NUM_HBM_CHANNELS));
    HS_HIP(ctx, hipSetDevice(ctx->device));
    return enqueue(ctx, int32_t(row_part_id), nullptr, nullptr);
}

int hs_feedback(hs_context* ctx, uint32_t scale_word, uint32_t shift_word) {
    int rc = check_ready(ctx);
    if (rc != HS_OK) return rc;
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    HS_HIP(ctx, hisparse::dev::launch_feedback(ctx->impl != HS_IMPL_FIXED, y_target(ctx), const_cast<uint32_t*>(x_source(ctx)),
                                               std::min(ctx->num_rows, ctx->num_cols), scale_word, shift_word, ctx->stream));
    return HS_OK;
}

int hs_iterate(hs_context* ctx, uint32_t iterations, uint32_t scale_word, uint32_t shift_word) {
    int rc = check_ready(ctx);
    if (rc != HS_OK) return rc;
    if (iterations == 0) return HS_OK;
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    const Feedback feedback{scale_word, shift_word};
    auto one_iteration = [&]() -> int { return enqueue(ctx, -1, nullptr, nullptr, &feedback); };
    // One iteration = 2-3 small launches, enqueued from this C loop far faster than the GPU retires them, so plain
    // stream-ordered launches are the default.  HISPARSE_ITERATE_GRAPH=1 captures chunks of 32 iterations into one
    // hipGraph and replays them instead; measured on ROCm 7.2 that is no faster (1k x 1k: 8.4 vs 8.6 us per iteration)
    // and slower for large matrices (ogbl-ppa 64.7 vs 60.9 us: gaps between graph nodes), so it stays opt-in.
    const char* graph_env = ctx_option(ctx, "HISPARSE_ITERATE_GRAPH");
    const bool use_graph = graph_env && std::atoi(graph_env) != 0;
    const uint32_t chunk = use_graph ? std::min<uint32_t>(iterations, 32) : 1;
    uint32_t done = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (chunk > 1 && hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (uint32_t i = 0; i < chunk && rc == HS_OK; ++i) rc = one_iteration();
        const hipError_t end = hipStreamEndCapture(ctx->stream, &graph);
        if (rc == HS_OK && end == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            hipError_t e = hipSuccess;
            for (; done + chunk <= iterations && e == hipSuccess; done += chunk) e = hipGraphLaunch(exec, ctx->stream);
            (void)hipGraphExecDestroy(exec);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) return fail(ctx, HS_ERR_HIP, std::string("hipGraphLaunch: ") + hipGetErrorString(e));
        } else {
            if (graph) (void)hipGraphDestroy(graph);
            (void)hipGetLastError();   // capture was refused (e.g. the legacy default stream): plain launches below
            if (rc != HS_OK) return rc;
        }
    } else {
        (void)hipGetLastError();
    }
    for (; done < iterations; ++done)
        if ((rc = one_iteration()) != HS_OK) return rc;
    return HS_OK;
}

// ---- SpMSpV extension (spmspv.hip) --------------------------------------------------------------------
int hs_load_matrix_csc(hs_context* ctx, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, uint32_t num_rows,
                       uint32_t num_cols) {
    if (!ctx || !indptr || num_rows == 0 || num_cols == 0) return fail(ctx, HS_ERR_BAD_ARG, "null argument or empty matrix");
    const uint64_t nnz = indptr[num_cols];
    if (indptr[0] != 0 || (nnz && (!row_indices || !value_words))) return fail(ctx, HS_ERR_BAD_MATRIX, "indptr must start at 0; arrays missing");
    for (uint32_t c = 0; c < num_cols; ++c)
        if (indptr[c + 1] < indptr[c]) return fail(ctx, HS_ERR_BAD_MATRIX, "CSC indptr must be non-decreasing");
    for (uint64_t e = 0; e < nnz; ++e)
        if (row_indices[e] >= num_rows) return fail(ctx, HS_ERR_BAD_MATRIX, "CSC row index out of range");
    const uint32_t bins = hisparse::dev::spmspv_bins(num_rows), block_bits = hisparse::dev::spmspv_block_bits(num_rows);
    if (bins > hisparse::dev::spmspv_max_bins()) return fail(ctx, HS_ERR_UNSUPPORTED, "more than 16 384 row blocks of 16 384 rows (268 M rows)");
    if (nnz > 0xfffffff0ull) return fail(ctx, HS_ERR_UNSUPPORTED, "more than 2^32 non-zeros");
    // a bin per row block, as large as the block's share of the matrix: products of an x that names every column at most once always fit
    std::vector<uint32_t> bin_base(size_t(bins) + 1, 0);
    for (uint64_t e = 0; e < nnz; ++e) bin_base[(row_indices[e] >> block_bits) + 1]++;
    for (uint32_t b = 0; b < bins; ++b) bin_base[b + 1] += bin_base[b];
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_csc(ctx);
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_csc_indptr), (size_t(num_cols) + 1) * 4));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_csc_rows), std::max<size_t>(nnz, 1) * 4));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_csc_vals), std::max<size_t>(nnz, 1) * 4));
    hisparse::dev::SpmspvScratch& w = ctx->csc_scratch;
    w.capacity = std::max<uint64_t>(nnz, 1);
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&w.keys), size_t(w.capacity) * 4));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&w.vals), size_t(w.capacity) * 4));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&w.bin_base), (size_t(bins) + 1) * 4));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&w.cursors), size_t(bins) * 4));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&w.overflow), 4));
    HS_HIP(ctx, hipMemcpy(w.bin_base, bin_base.data(), bin_base.size() * 4, hipMemcpyHostToDevice));
    HS_HIP(ctx, hipMemset(w.cursors, 0, size_t(bins) * 4));
    HS_HIP(ctx, hipMemset(w.overflow, 0, 4));
    // y: also large enough for the dense SpMV's padded rows (dense dispatch of hs_spmspv writes it directly)
    ctx->csc_y_words = std::max(num_rows, ctx->matrix_loaded ? ctx->num_rows : 0u);
    ctx->csc_y_words = std::max<uint32_t>(ctx->csc_y_words, uint32_t((uint64_t(num_rows) + ctx->geom.row_divisor - 1) / ctx->geom.row_divisor * ctx->geom.row_divisor));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_csc_y), size_t(ctx->csc_y_words) * 4));
    HS_HIP(ctx, hipMemcpy(ctx->d_csc_indptr, indptr, (size_t(num_cols) + 1) * 4, hipMemcpyHostToDevice));
    if (nnz) {
        HS_HIP(ctx, hipMemcpy(ctx->d_csc_rows, row_indices, nnz * 4, hipMemcpyHostToDevice));
        HS_HIP(ctx, hipMemcpy(ctx->d_csc_vals, value_words, nnz * 4, hipMemcpyHostToDevice));
    }
    HS_HIP(ctx, hipMemset(ctx->d_csc_y, 0, size_t(ctx->csc_y_words) * 4));
    ctx->csc_col_len.resize(num_cols);
    for (uint32_t c = 0; c < num_cols; ++c) ctx->csc_col_len[c] = indptr[c + 1] - indptr[c];
    ctx->csc_rows = num_rows;
    ctx->csc_cols = num_cols;
    ctx->csc_nnz = nnz;
    ctx->dense_spmv_us = 0.0;
    return HS_OK;
}

namespace {

// one pass: expand + accumulate over `count` device-resident entries
int spmspv_pass(hs_context* ctx, const hisparse::dev::hs_idx_val_dev* x_dev, uint32_t count, bool add_to_y) {
    HS_HIP(ctx, hisparse::dev::launch_spmspv(ctx->impl != HS_IMPL_FIXED, ctx->d_csc_indptr, ctx->d_csc_rows, ctx->d_csc_vals, x_dev, count, ctx->csc_rows,
                                             ctx->csc_cols, ctx->csc_scratch, add_to_y, ctx->d_csc_y, ctx->stream));
    return HS_OK;
}

// The dense SpMV instead (hs_spmspv above the crossover): x scattered into a zero vector, the context's loaded matrix, y into the SpMSpV
// result buffer.  Only when hs_load_matrix holds a matrix of the CSC matrix's shape (padded) -- the caller's contract is that it is the
// SAME matrix (hisparse_hip.h) -- and x names no column twice.
bool dense_dispatch_possible(const hs_context* ctx) {
    if (!ctx->matrix_loaded) return false;
    const Geometry& g = ctx->geom;
    const uint64_t rows = (uint64_t(ctx->csc_rows) + g.row_divisor - 1) / g.row_divisor * g.row_divisor, cols = (uint64_t(ctx->csc_cols) + 7) / 8 * 8;
    return rows == ctx->num_rows && cols == ctx->num_cols && ctx->csc_y_words >= ctx->num_rows;
}
int spmspv_dense(hs_context* ctx, const hisparse::dev::hs_idx_val_dev* x_dev, uint32_t count) {
    if (!ctx->d_x_dense) HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_x_dense), size_t(ctx->num_cols) * 4));
    HS_HIP(ctx, hisparse::dev::launch_spmspv_scatter_x(x_dev, count, ctx->num_cols, ctx->d_x_dense, ctx->stream));
    const uint32_t* saved_x = ctx->x_bound;
    uint32_t* saved_y = ctx->y_bound;
    ctx->x_bound = ctx->d_x_dense;
    ctx->y_bound = ctx->d_csc_y;
    int rc = enqueue(ctx, -1, nullptr, nullptr);
    if (rc == HS_OK) rc = flush_combine(ctx);      // never leave a sum owed to a transient target
    ctx->x_bound = saved_x;
    ctx->y_bound = saved_y;
    if (rc == HS_OK) ++ctx->spmspv_dense_dispatches;
    return rc;
}

}  // namespace

int hs_spmspv_device(hs_context* ctx, const hs_idx_val* x_entries_dev, uint32_t count) {
    if (!ctx || (count && !x_entries_dev)) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (!ctx->d_csc_indptr) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix_csc has not been called");
    if (reinterpret_cast<uintptr_t>(x_entries_dev) & 7u) return fail(ctx, HS_ERR_BAD_ARG, "device entries must be 8-byte aligned");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    return spmspv_pass(ctx, reinterpret_cast<const hisparse::dev::hs_idx_val_dev*>(x_entries_dev), count, false);
}

int hs_spmspv(hs_context* ctx, const hs_idx_val* x_entries, uint32_t count) {
    if (!ctx || (count && !x_entries)) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (!ctx->d_csc_indptr) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix_csc has not been called");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    // One pass over the entries: range check, repeat check (a bit per column, kept zero between calls) and this call's product count.
    // A bin of the product list holds as many products as the matrix has non-zeros in that row block: an x that names a column more than once
    // can ask for more.  Such a call is cut into passes of UNIQUE columns -- pass i takes the i-th occurrence of every column: each pass fits by
    // construction -- y = first pass, y += the others (the sums are order-free: exact in fixed point, tolerance in float).  Repeats also rule
    // the dense dispatch out (two entries of one column are two separately rounded products, not one product of their sum).
    std::vector<uint32_t> pass_end;      // entries [pass_end[i-1], pass_end[i]) of the STAGED order form pass i
    bool repeats = false;
    uint64_t products = 0;
    {
        if (ctx->sx_seen.size() != (size_t(ctx->csc_cols) + 7) / 8) ctx->sx_seen.assign((size_t(ctx->csc_cols) + 7) / 8, 0);
        uint8_t* seen = ctx->sx_seen.data();
        uint32_t k = 0;
        for (; k < count; ++k) {
            const uint32_t col = x_entries[k].index;
            if (col >= ctx->csc_cols) break;
            repeats |= (seen[col >> 3] >> (col & 7)) & 1u;
            seen[col >> 3] |= uint8_t(1u << (col & 7));
            products += ctx->csc_col_len[col];
        }
        for (uint32_t j = 0; j < k; ++j) seen[x_entries[j].index >> 3] = 0;
        if (k < count) return fail(ctx, HS_ERR_BAD_ARG, "sparse vector index out of range");
    }
    if (count > ctx->sx_capacity) {
        HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_sx) (void)hipHostFree(ctx->h_sx);
        ctx->d_sx = ctx->h_sx = nullptr;
        ctx->sx_capacity = 0;
        const uint32_t cap = std::max<uint32_t>(count, 1024);
        // pinned AND mapped: the expand kernel reads the entries straight out of host memory (8 bytes per entry, once, coalesced) -- no copy
        // command in front of it (an H2D of 46 KB put ~15 us between the call and its first kernel).  Two halves used in turn, so that the
        // host fills one while the previous call's kernels may still be reading the other.
        HS_HIP(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_sx), size_t(cap) * 2 * 8, hipHostMallocMapped));
        HS_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_sx), ctx->h_sx, 0));
        for (hipEvent_t& e : ctx->sx_read)
            if (!e) HS_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->sx_capacity = cap;
        ctx->sx_turn = 0;
    }
    const uint32_t half = ctx->sx_turn & 1u;
    ctx->sx_turn++;
    if (count) HS_HIP(ctx, hipEventSynchronize(ctx->sx_read[half]));      // the call before last read this half (long done, normally; a never-recorded event is complete)
    hisparse::dev::hs_idx_val_dev* const staged = ctx->h_sx + size_t(half) * ctx->sx_capacity;
    const hisparse::dev::hs_idx_val_dev* const staged_dev = ctx->d_sx + size_t(half) * ctx->sx_capacity;
    if (!repeats) {
        if (count) std::memcpy(staged, x_entries, size_t(count) * sizeof(hs_idx_val));
        pass_end.push_back(count);
    } else {
        // occurrence number of every entry, then a stable counting sort by it into the staging buffer
        std::vector<uint32_t> occurrence(count), seen_times(ctx->csc_cols, 0), per_pass;
        for (uint32_t k = 0; k < count; ++k) {
            occurrence[k] = seen_times[x_entries[k].index]++;
            if (occurrence[k] >= per_pass.size()) per_pass.resize(occurrence[k] + 1, 0);
            per_pass[occurrence[k]]++;
        }
        std::vector<uint32_t> at(per_pass.size(), 0);
        for (size_t i = 1; i < per_pass.size(); ++i) at[i] = at[i - 1] + per_pass[i - 1];
        for (size_t i = 0; i < per_pass.size(); ++i) pass_end.push_back(at[i] + per_pass[i]);
        for (uint32_t k = 0; k < count; ++k) {
            staged[at[occurrence[k]]].index = x_entries[k].index;
            staged[at[occurrence[k]]].val = x_entries[k].val;
            ++at[occurrence[k]];
        }
    }
    // Above the crossover the dense SpMV is faster (it reads every non-zero once, coalesced, at 6-8 bytes; the sparse path reads a column
    // entry, writes its product into a bin and reads it again): hisparse_hip.h.  The host knows this call's product count exactly (the
    // columns' lengths), the sparse path costs ~14 us + products / 45 G/s (profiles/r04_spmspv_binned.txt: ogbl-ppa, mouse_gene, pokec), and the dense SpMV of the
    // loaded matrix is TIMED once, on the first call that could use it (three launches on a zero vector and one synchronisation; hyper-
    // sparse matrices run at a third of the roofline, so no formula over the non-zero count would do).  `spmspv_crossover` (a fraction of
    // the columns) overrides the rule; `spmspv` = sparse | dense forces a path.
    const char* force = ctx_option(ctx, "HISPARSE_SPMSPV");
    const bool possible = !repeats && dense_dispatch_possible(ctx);
    bool want_dense = false;
    // The dense dispatch is OPT-IN: the CSC matrix is independent of the matrix hs_load_matrix holds -- a caller may keep
    // A for SpMV and A^T (or anything else of the same shape) as CSC -- and only the caller knows that the two are the same matrix.  It says
    // so with `spmspv` = auto (the rule below) | dense (always), or by setting `spmspv_crossover`; without one of them every call takes the
    // sparse path over the CSC arrays, whatever its size.
    const bool automatic = force && std::string(force) == "auto";
    if (force && !automatic) {
        want_dense = std::string(force) == "dense";
    } else if (const char* v = ctx_option(ctx, "HISPARSE_SPMSPV_CROSSOVER")) {
        const double crossover = std::atof(v);
        want_dense = crossover > 0.0 && double(count) > crossover * double(ctx->csc_cols);
    } else if (automatic && possible && 14.0 + double(products) / 45000.0 > 30.0) {      // (below 30 us no dense SpMV of a matrix worth a CSC copy competes)
        if (ctx->dense_spmv_us <= 0.0) {
            if (!ctx->d_x_dense) {
                HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_x_dense), size_t(ctx->num_cols) * 4));
                HS_HIP(ctx, hipMemsetAsync(ctx->d_x_dense, 0, size_t(ctx->num_cols) * 4, ctx->stream));
            }
            const uint32_t* saved_x = ctx->x_bound;
            uint32_t* saved_y = ctx->y_bound;
            ctx->x_bound = ctx->d_x_dense;
            ctx->y_bound = ctx->d_csc_y;
            int rc = enqueue(ctx, -1, nullptr, nullptr);      // warm, then three timed
            hipEvent_t t0 = nullptr, t1 = nullptr;
            if (rc == HS_OK && hipEventCreate(&t0) == hipSuccess && hipEventCreate(&t1) == hipSuccess) {
                (void)hipEventRecord(t0, ctx->stream);
                for (int i = 0; i < 3 && rc == HS_OK; ++i) rc = enqueue(ctx, -1, nullptr, nullptr);
                // A carried plan owes the last step's sum to d_csc_y here; if the rule below then takes the SPARSE path, that combine would
                // run after spmspv_pass and overwrite its y with A*0: settle it inside the timed region
                if (rc == HS_OK) rc = flush_combine(ctx);
                (void)hipEventRecord(t1, ctx->stream);
                float ms = 0.0f;
                if (rc == HS_OK && hipEventSynchronize(t1) == hipSuccess && hipEventElapsedTime(&ms, t0, t1) == hipSuccess) ctx->dense_spmv_us = std::max(1.0, double(ms) * 1000.0 / 3.0);
            }
            if (t0) (void)hipEventDestroy(t0);
            if (t1) (void)hipEventDestroy(t1);
            if (rc == HS_OK) rc = flush_combine(ctx);      // (the event creation failed: the warm step's sum is still owed)
            else ctx->pending = -1;
            ctx->x_bound = saved_x;
            ctx->y_bound = saved_y;
            if (rc != HS_OK) return rc;
        }
        // + the scatter of x into the zero vector (a memset and a small kernel: ~8 us)
        want_dense = ctx->dense_spmv_us > 0.0 && 14.0 + double(products) / 45000.0 > ctx->dense_spmv_us + 8.0;
    }
    int rc = HS_OK;
    if (want_dense && possible) {
        rc = spmspv_dense(ctx, staged_dev, count);
    } else {
        uint32_t begin = 0;
        for (size_t i = 0; i < pass_end.size() && rc == HS_OK; ++i) {
            rc = spmspv_pass(ctx, staged_dev + begin, pass_end[i] - begin, i != 0);
            begin = pass_end[i];
        }
    }
    if (count && rc == HS_OK) HS_HIP(ctx, hipEventRecord(ctx->sx_read[half], ctx->stream));      // the kernels that read the staging buffer are behind this
    return rc;
}

int hs_read_spmspv_result(hs_context* ctx, void* packed_y, uint32_t num_rows) {
    if (!ctx || !packed_y) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (!ctx->d_csc_indptr) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix_csc has not been called");
    if (num_rows != ctx->csc_rows) return fail(ctx, HS_ERR_BAD_ARG, "result length must equal the CSC matrix's row count");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    uint32_t overflow = 0;
    HS_HIP(ctx, hipMemcpyAsync(packed_y, ctx->d_csc_y, size_t(num_rows) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HS_HIP(ctx, hipMemcpyAsync(&overflow, ctx->csc_scratch.overflow, sizeof(overflow), hipMemcpyDeviceToHost, ctx->stream));
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (overflow) {      // only hs_spmspv_device can get here: hs_spmspv cuts such a call into passes
        HS_HIP(ctx, hipMemset(ctx->csc_scratch.overflow, 0, sizeof(overflow)));
        HS_HIP(ctx, hipMemset(ctx->csc_scratch.cursors, 0, size_t(hisparse::dev::spmspv_bins(ctx->csc_rows)) * 4));
        return fail(ctx, HS_ERR_BAD_ARG, "hs_spmspv_device: the entries asked for more products than the matrix has non-zeros (columns named more than "
                                         "once): the result is incomplete; hs_spmspv with host entries splits such a call");
    }
    return HS_OK;
}

int hs_spmspv_status(hs_context* ctx, uint32_t* overflowed, void** overflow_word_dev) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->d_csc_indptr) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix_csc has not been called");
    if (overflow_word_dev) *overflow_word_dev = ctx->csc_scratch.overflow;
    if (overflowed) {
        HS_HIP(ctx, hipSetDevice(ctx->device));
        HS_FLUSH(ctx);
        HS_HIP(ctx, hipMemcpyAsync(overflowed, ctx->csc_scratch.overflow, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return HS_OK;
}

int hs_sync(hs_context* ctx) {
    if (!ctx) return HS_ERR_BAD_ARG;
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HS_OK;
}

int hs_read_result(hs_context* ctx, void* packed_y, uint32_t num_rows) {
    if (!ctx || !packed_y) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    if (num_rows != ctx->num_rows) return fail(ctx, HS_ERR_BAD_ARG, "result length must equal the padded row count");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    HS_HIP(ctx, hipMemcpyAsync(packed_y, y_target(ctx), size_t(num_rows) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HS_OK;
}

int hs_set_option(hs_context* ctx, const char* key, const char* value) {
    if (!ctx || !key) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    std::string k(key);
    for (char& ch : k) ch = char(std::toupper(static_cast<unsigned char>(ch)));
    if (k.rfind("HISPARSE_", 0) == 0) k = k.substr(9);
    bool known = false;
    for (const char* name : kOptionKeys) known = known || k == name;
    if (k == "ABLATE" || k == "DEPTH" || k == "TIMELINE_OUT")
        return fail(ctx, HS_ERR_BAD_ARG, "'" + k + "' is a profiling switch of libhisparse_hip_prof.so (wrong results by design), not an option of this library");
    if (!known) return fail(ctx, HS_ERR_BAD_ARG, "unknown option '" + std::string(key) + "'");
    if (value && *value) ctx->options["HISPARSE_" + k] = value;
    else ctx->options.erase("HISPARSE_" + k);
    drop_batch_graph(ctx);      // the captured batch bakes in whatever enqueue() read at capture time: any option change invalidates it
    return HS_OK;
}

int hs_set_stream(hs_context* ctx, void* hip_stream) {
    if (!ctx) return HS_ERR_BAD_ARG;
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    // work already enqueued must not be overtaken by work on the new stream; a caller-owned stream may be gone by now,
    // so only the library's own stream is synchronised by handle
    if (ctx->stream == ctx->own_stream) HS_HIP(ctx, hipStreamSynchronize(ctx->own_stream));
    else HS_HIP(ctx, hipDeviceSynchronize());
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return HS_OK;
}

int hs_get_stream(hs_context* ctx, void** hip_stream) {
    if (!ctx || !hip_stream) return HS_ERR_BAD_ARG;
    HS_FLUSH(ctx);
    ctx->stream_shared = true;      // whoever holds the handle may order work against it: from now on every step completes in itself
    *hip_stream = ctx->stream;
    return HS_OK;
}

int hs_device_vector(hs_context* ctx, void** x_dev) {
    if (!ctx || !x_dev) return HS_ERR_BAD_ARG;
    if (!ctx->d_x) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_vector has not been called");
    *x_dev = ctx->d_x;
    return HS_OK;
}

int hs_device_result(hs_context* ctx, void** y_dev) {
    if (!ctx || !y_dev) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    HS_FLUSH(ctx);
    *y_dev = ctx->d_y;
    return HS_OK;
}

int hs_bind_device_vector(hs_context* ctx, const void* x_dev) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (x_dev && (reinterpret_cast<uintptr_t>(x_dev) & 15u)) return fail(ctx, HS_ERR_BAD_ARG, "device vector must be 16-byte aligned");
    if (static_cast<const uint32_t*>(x_dev) != ctx->x_bound) drop_batch_graph(ctx);
    ctx->x_bound = static_cast<const uint32_t*>(x_dev);
    return HS_OK;
}

int hs_bind_device_result(hs_context* ctx, void* y_dev) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (y_dev && (reinterpret_cast<uintptr_t>(y_dev) & 15u)) return fail(ctx, HS_ERR_BAD_ARG, "device result must be 16-byte aligned");
    if (static_cast<uint32_t*>(y_dev) != ctx->y_bound) {
        HS_FLUSH(ctx);      // (an owed sum belongs to the old target)
        drop_batch_graph(ctx);
    }
    ctx->y_bound = static_cast<uint32_t*>(y_dev);
    return HS_OK;
}

int hs_push_result(hs_context* ctx, void* const* dst, uint32_t n_dst, uint32_t num_words) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    if (n_dst == 0) return HS_OK;
    if (!dst || n_dst > hisparse::dev::kMaxPushTargets) return fail(ctx, HS_ERR_BAD_ARG, "1 .. 8 destinations");
    if (num_words > ctx->num_rows || (num_words & 3u)) return fail(ctx, HS_ERR_BAD_ARG, "num_words must be a multiple of 4 and at most the padded row count");
    for (uint32_t k = 0; k < n_dst; ++k)
        if (!dst[k] || (reinterpret_cast<uintptr_t>(dst[k]) & 15u)) return fail(ctx, HS_ERR_BAD_ARG, "destinations must be 16-byte aligned device pointers");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    HS_HIP(ctx, hisparse::dev::launch_push_result(y_target(ctx), dst, n_dst, num_words, ctx->stream));
    return HS_OK;
}

// SpMM as k SpMVs over the resident image (hisparse_hip.h): every column of X through the same kernels, so every column of Y is
// exactly what hs_run gives for it.
int hs_spmm_device(hs_context* ctx, const void* x_dev, uint64_t ldx, void* y_dev, uint64_t ldy, uint32_t k) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    if (k == 0) return HS_OK;
    if (!x_dev || !y_dev) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if ((reinterpret_cast<uintptr_t>(x_dev) & 15u) || (reinterpret_cast<uintptr_t>(y_dev) & 15u) || (ldx & 3u) || (ldy & 3u))
        return fail(ctx, HS_ERR_BAD_ARG, "device matrices must be 16-byte aligned with leading dimensions that are multiples of 4 words");
    if (ldx < ctx->num_cols || ldy < ctx->num_rows) return fail(ctx, HS_ERR_BAD_ARG, "leading dimensions must cover the padded column / row counts");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t* x_saved = ctx->x_bound;
    uint32_t* y_saved = ctx->y_bound;
    int rc = HS_OK;
    const bool is_float = ctx->impl != HS_IMPL_FIXED;
    uint32_t j = 0;
    // BITMAP images (dense rows: pruned-NN layers, which are multiplied with batches in practice): 4, then 2 columns at a time through
    // the fused kernel of spmm_bitmap.hip -- masks and values are streamed once for them.  Everything else, and a last odd column:
    // one SpMV per column.
    const char* fused_env = ctx_option(ctx, "HISPARSE_SPMM_FUSED");      // read per call (a test may change it)
    const bool fused_enabled = !(fused_env && std::string(fused_env) == "0");
    const char* mfma_env = ctx_option(ctx, "HISPARSE_SPMM_MFMA");
    // float BITMAP matrices, 16 columns at a time on the matrix engine: the matrix is streamed once per 16 columns and every x word is
    // shared by 16 rows in registers (spmm_mfma.hip)
    if (fused_enabled && !(mfma_env && std::string(mfma_env) == "0") && ctx->d_mfma && is_float) {
        // (5 .. 15 columns left over: still one pass -- 25 us on transformer-50 whatever it carries, against 25 us per FOUR columns of the fused kernel)
        while (k - j >= 5) {
            const uint32_t vectors = std::min<uint32_t>(16, k - j);
            hisparse::dev::SpmmMfmaLaunch a;
            a.vectors = vectors;
            a.words = ctx->d_mfma;
            a.offsets_word = ctx->mfma_info.offsets_word; a.values_word = ctx->mfma_info.values_word;
            a.tiles = ctx->mfma_info.tiles; a.groups = ctx->mfma_info.groups; a.chunk = ctx->mfma_info.chunk; a.chunks = ctx->mfma_info.chunks;
            a.x = static_cast<const uint32_t*>(x_dev) + size_t(j) * ldx;
            a.ldx = ldx;
            a.x_interleaved = ctx->d_mfma_x;
            a.partial = ctx->d_mfma_partial;
            a.flag = ctx->d_mfma_flag;
            a.call = ++ctx->mfma_call ? ctx->mfma_call : ++ctx->mfma_call;
            a.y = static_cast<uint32_t*>(y_dev) + size_t(j) * ldy;
            a.ldy = ldy;
            a.num_rows = ctx->num_rows;
            a.num_cols = ctx->num_cols;
            HS_HIP(ctx, hisparse::dev::launch_spmm_mfma(a, ctx->stream));
            j += vectors;
        }
    }
    if (fused_enabled && ctx->format == hisparse::dev::kFormatBitmap && ctx->col_slices == 1) {
        for (uint32_t group : {4u, 2u}) {
            if (ctx->max_block_rows > hisparse::dev::spmm_bitmap_max_block_rows(is_float, group)) continue;
            while (k - j >= group) {
                if (!ctx->d_x_interleaved) HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_x_interleaved), size_t(ctx->num_cols) * 4 * 4 + 64));
                hisparse::dev::SpmmLaunch a;
                a.image = ctx->d_image;
                a.blocks = ctx->d_blocks;
                a.units = ctx->d_units;
                a.x = static_cast<const uint32_t*>(x_dev) + size_t(j) * ldx;
                a.ldx = ldx;
                a.x_interleaved = ctx->d_x_interleaved;
                a.y = static_cast<uint32_t*>(y_dev) + size_t(j) * ldy;
                a.ldy = ldy;
                a.vectors = group;
                a.num_cols = ctx->num_cols;
                a.num_workgroups = ctx->num_workgroups;
                a.max_block_rows = ctx->max_block_rows;
                HS_HIP(ctx, hisparse::dev::launch_spmm_bitmap(is_float, a, ctx->stream));
                j += group;
            }
        }
    }
    // SWEEP images planned for it (option spmm_vectors = 4 at load time): four columns per pass through the matrix (spmm_sweep.hip); the
    // last pass may carry fewer (its missing columns are zero vectors whose results are not copied out)
    if (fused_enabled && ctx->format == hisparse::dev::kFormatSweep && ctx->spmm_vectors == 4 && k - j >= 2) {
        if (int frc = flush_combine(ctx)) return frc;
        const size_t rows = ctx->num_rows;
        if (!ctx->d_spmm_x4) HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_spmm_x4), size_t(ctx->num_cols) * 16 + 64));
        if (!ctx->d_spmm_y) HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_spmm_y), rows * 16));
        if (ctx->col_slices > 1 && !ctx->d_spmm_partial) HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_spmm_partial), size_t(ctx->col_slices) * rows * 16));
        while (j < k) {
            const uint32_t vectors = std::min<uint32_t>(4, k - j);
            hisparse::dev::SpmmSweepLaunch a;
            a.image = ctx->d_image;
            a.blocks = ctx->d_blocks;
            a.x = static_cast<const uint32_t*>(x_dev) + size_t(j) * ldx;
            a.ldx = ldx;
            a.x4 = ctx->d_spmm_x4;
            a.out = ctx->col_slices > 1 ? ctx->d_spmm_partial : ctx->d_spmm_y;
            a.vectors = vectors;
            a.num_rows = ctx->num_rows;
            a.num_cols = ctx->num_cols;
            a.num_workgroups = ctx->num_workgroups;
            a.max_block_rows = ctx->max_block_rows;
            HS_HIP(ctx, hisparse::dev::launch_spmm_sweep(is_float, a, ctx->stream));
            if (ctx->col_slices > 1)      // the four vectors' partial rows lie back to back inside a slice: ONE combine over 4 x rows "rows"
                HS_HIP(ctx, hisparse::dev::launch_combine_slices(is_float, ctx->d_spmm_partial, ctx->d_spmm_y, uint32_t(4 * rows), ctx->col_slices, 0,
                                                                 uint32_t(4 * rows), ctx->stream));
            HS_HIP(ctx, hipMemcpy2DAsync(static_cast<uint32_t*>(y_dev) + size_t(j) * ldy, size_t(ldy) * 4, ctx->d_spmm_y, rows * 4, rows * 4, vectors,
                                         hipMemcpyDeviceToDevice, ctx->stream));
            j += vectors;
        }
    }
    for (; j < k && rc == HS_OK; ++j) {
        ctx->x_bound = static_cast<const uint32_t*>(x_dev) + size_t(j) * ldx;
        ctx->y_bound = static_cast<uint32_t*>(y_dev) + size_t(j) * ldy;
        rc = enqueue(ctx, -1, nullptr, nullptr);
    }
    // the last column's sum is not left owed to the caller's memory: an event or a device-wide synchronisation then completes Y, and the
    // caller may free y_dev
    if (rc == HS_OK) rc = flush_combine(ctx);
    else ctx->pending = -1;
    ctx->x_bound = x_saved;
    ctx->y_bound = y_saved;
    return rc;
}

int hs_spmm(hs_context* ctx, const void* packed_x, uint32_t num_cols, uint32_t k, void* packed_y, uint32_t num_rows) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    if (num_cols != ctx->num_cols || num_rows != ctx->num_rows) return fail(ctx, HS_ERR_BAD_ARG, "dimensions must equal the matrix's padded column / row counts");
    if (k == 0) return HS_OK;
    if (!packed_x || !packed_y) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t ldx = (uint64_t(num_cols) + 3u) & ~uint64_t(3), ldy = (uint64_t(num_rows) + 3u) & ~uint64_t(3);
    struct Buffers {   // freed on every return path
        uint32_t *x = nullptr, *y = nullptr;
        ~Buffers() { if (x) (void)hipFree(x); if (y) (void)hipFree(y); }
    } b;
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&b.x), size_t(ldx) * k * 4 + 64));
    HS_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&b.y), size_t(ldy) * k * 4));
    HS_HIP(ctx, hipMemcpy2DAsync(b.x, size_t(ldx) * 4, packed_x, size_t(num_cols) * 4, size_t(num_cols) * 4, k, hipMemcpyHostToDevice, ctx->stream));
    int rc = hs_spmm_device(ctx, b.x, ldx, b.y, ldy, k);
    if (rc != HS_OK) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    HS_FLUSH(ctx);
    HS_HIP(ctx, hipMemcpy2DAsync(packed_y, size_t(num_rows) * 4, b.y, size_t(ldy) * 4, size_t(num_rows) * 4, k, hipMemcpyDeviceToHost, ctx->stream));
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return HS_OK;
}

int hs_get_stats(const hs_context* ctx, hs_stats* stats) {
    if (!ctx || !stats) return HS_ERR_BAD_ARG;
    *stats = ctx->stats;
    return HS_OK;
}

int hs_time_runs(hs_context* ctx, int warmup, int runs, float* total_ms, float* kernel_ms) {
    int rc = check_ready(ctx);
    if (rc != HS_OK) return rc;
    if (warmup < 0 || runs <= 0) return fail(ctx, HS_ERR_BAD_ARG, "need warmup >= 0 and runs > 0");
    HS_HIP(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < warmup; ++i)
        if ((rc = enqueue(ctx, -1, nullptr, nullptr)) != HS_OK) return rc;
    HS_FLUSH(ctx);
    HS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    struct Events {   // destroyed on every return path
        std::vector<hipEvent_t> ev;
        ~Events() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } events;
    events.ev.assign(2 + (kernel_ms ? size_t(runs) * 2 : 0), nullptr);
    for (auto& ev : events.ev) HS_HIP(ctx, hipEventCreate(&ev));
    const hipEvent_t begin = events.ev[0], end = events.ev[1];
    hipEvent_t* k = events.ev.data() + 2;
    HS_HIP(ctx, hipEventRecord(begin, ctx->stream));
    for (int i = 0; i < runs; ++i)
        if ((rc = enqueue(ctx, -1, kernel_ms ? k[size_t(i) * 2] : nullptr, kernel_ms ? k[size_t(i) * 2 + 1] : nullptr)) != HS_OK) return rc;
    HS_FLUSH(ctx);                                     // (the last step's sum may still be owed)
    HS_HIP(ctx, hipEventRecord(end, ctx->stream));
    HS_HIP(ctx, hipEventSynchronize(end));
    float ms = 0.0f;
    HS_HIP(ctx, hipEventElapsedTime(&ms, begin, end));
    if (total_ms) *total_ms = ms;
    if (kernel_ms) {
        float sum = 0.0f;
        for (int i = 0; i < runs; ++i) {
            float one = 0.0f;
            HS_HIP(ctx, hipEventElapsedTime(&one, k[size_t(i) * 2], k[size_t(i) * 2 + 1]));
            sum += one;
        }
        *kernel_ms = sum;
    }
    return HS_OK;
}

int hs_time_kernel(hs_context* ctx, int warmup, int runs, float* kernel_ms) {
    int rc = check_ready(ctx);
    if (rc != HS_OK) return rc;
    if (warmup < 0 || runs <= 0 || !kernel_ms) return fail(ctx, HS_ERR_BAD_ARG, "need warmup >= 0, runs > 0 and an output pointer");
    if (const char* why = hisparse::dev::profiling_switch_error()) return fail(ctx, HS_ERR_BAD_ARG, why);
    HS_HIP(ctx, hipSetDevice(ctx->device));
    HS_FLUSH(ctx);
    const bool is_float = ctx->impl != HS_IMPL_FIXED;
    const hisparse::dev::SpmvLaunch args = launch_args(ctx, -1);
    // a plan whose combine pass is carried into the next step's kernel: the kernel as it runs in consecutive steps, i.e. with that work in it
    const bool carried = ctx->carry_combine && ctx->col_slices > 1 && ctx->stream == ctx->own_stream && !ctx->stream_shared;
    auto launch = [&]() -> int {
        if (carried) return enqueue(ctx, -1, nullptr, nullptr);
        HS_HIP(ctx, hisparse::dev::launch_spmv(is_float, args, ctx->stream));
        return HS_OK;
    };
    for (int i = 0; i < warmup; ++i)
        if ((rc = launch()) != HS_OK) return rc;
    hipEvent_t ev[2] = {nullptr, nullptr};
    struct Guard { hipEvent_t* e; ~Guard() { for (int i = 0; i < 2; ++i) if (e[i]) (void)hipEventDestroy(e[i]); } } guard{ev};
    HS_HIP(ctx, hipEventCreate(&ev[0]));
    HS_HIP(ctx, hipEventCreate(&ev[1]));
    HS_HIP(ctx, hipEventRecord(ev[0], ctx->stream));
    for (int i = 0; i < runs; ++i)
        if ((rc = launch()) != HS_OK) return rc;
    HS_HIP(ctx, hipEventRecord(ev[1], ctx->stream));
    HS_HIP(ctx, hipEventSynchronize(ev[1]));
    HS_HIP(ctx, hipEventElapsedTime(kernel_ms, ev[0], ev[1]));
    // column-sliced plans: the launches above left partial sums only; one whole step (or the owed combine) puts y back in order
    if (carried) return flush_combine(ctx);
    return enqueue(ctx, -1, nullptr, nullptr);
}

}  // extern "C"
