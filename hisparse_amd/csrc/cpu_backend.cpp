// cpu_backend.cpp — libhisparse_cpu.so: the drop-in boundary of include/hisparse_hip.h on host threads, for machines WITHOUT a GPU
// ("a CPU backend with the same symbols for config (1)" -- the role spmv_csim plays for the reference:
// sw/Makefile builds the same driver against csim or against the xclbin).
//
// This is a DIFFERENT LIBRARY that a driver links or loads INSTEAD of libhisparse_hip.so.  Nothing in libhisparse_hip.so, in
// hisparse_amd/device.py or in bench.py falls back to it: without a usable gfx950 device they fail (HS_ERR_NO_DEVICE).  It is its own
// code -- the CPSR decode of tiles_common.h (shared with the GPU library's load path) into plain CSR rows, then one pass per row --
// and does not touch oracle/ (which is test infrastructure and restates the reference's dataflow cluster by cluster instead).
//
// Arithmetic = the GPU kernels' (spmv_device.h), so the two libraries agree bit for bit in fixed point:
//   fixed: every product narrowed to Q8.24 with AP_RND / AP_SAT (spmv/libfpga/pe.h:64), summed in 64 bits, clamped once (pe.h:72;
//          saturating adds of non-negative terms are order free);
//   float: one fp32 multiply per product (pe-pob.h:63-65, pe-stall.h:52), summed in double in storage order, rounded to fp32 once.
// Core entry points only (create / load_matrix / load_matrix_csr / load_vector / run / run_partition / sync / read_result / stats /
// time_runs / errors); the device-memory hooks and the extensions answer HS_ERR_UNSUPPORTED.
#include "hisparse_hip.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "hisparse/common.h"
#include "hisparse/q8_24.h"
#include "tiles_common.h"

using hisparse::Geometry;

struct hs_context {
    int impl = 0;
    Geometry geom{};
    bool matrix_loaded = false, vector_loaded = false;
    uint32_t num_rows = 0, num_cols = 0, row_parts = 0, col_parts = 0;
    std::vector<uint64_t> indptr;      // CSR of the padded matrix: absolute columns, value words
    std::vector<uint32_t> indices, words;
    std::vector<uint32_t> x, y;
    hs_stats stats{};
    std::string error;
};

namespace {

thread_local std::string g_create_error;

int fail(hs_context* ctx, int code, const std::string& msg) {
    if (ctx) ctx->error = msg; else g_create_error = msg;
    return code;
}

// rows [lo, hi): one pass per row
void multiply_rows(const hs_context& c, uint32_t lo, uint32_t hi, std::vector<uint32_t>& y) {
    const bool fixed = c.impl == HS_IMPL_FIXED;
    hisparse::dev::detail::parallel_for((size_t(hi - lo) + 1023) / 1024, [&](size_t chunk) {
        const uint32_t r0 = lo + uint32_t(chunk) * 1024, r1 = uint32_t(std::min<uint64_t>(hi, uint64_t(r0) + 1024));
        for (uint32_t r = r0; r < r1; ++r) {
            if (fixed) {
                uint64_t sum = 0;
                for (uint64_t e = c.indptr[r]; e < c.indptr[r + 1]; ++e) sum += hisparse::q8_24_mul_raw(c.words[e], c.x[c.indices[e]]);
                y[r] = sum > hisparse::Q8_24_MAX_RAW ? hisparse::Q8_24_MAX_RAW : uint32_t(sum);
            } else {
                double sum = 0.0;
                for (uint64_t e = c.indptr[r]; e < c.indptr[r + 1]; ++e) {
                    float a, b;
                    std::memcpy(&a, &c.words[e], 4);
                    std::memcpy(&b, &c.x[c.indices[e]], 4);
                    const float prod = a * b;          // built with -ffp-contract=off: a multiply, then the add
                    sum += double(prod);
                }
                const float out = float(sum);
                std::memcpy(&y[r], &out, 4);
            }
        }
    });
}

int check_ready(hs_context* ctx) {
    if (!ctx) return HS_ERR_BAD_ARG;
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    if (!ctx->vector_loaded || ctx->x.size() != ctx->num_cols) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_vector has not been called for this matrix");
    return HS_OK;
}

int check_dims(hs_context* ctx, uint32_t num_rows, uint32_t num_cols, uint32_t rp, uint32_t cp) {
    const Geometry& g = ctx->geom;
    if (num_rows == 0 || num_cols == 0) return fail(ctx, HS_ERR_BAD_ARG, "empty matrix");
    if (num_rows % g.row_divisor != 0 || num_cols % hisparse::PACK_SIZE != 0)
        return fail(ctx, HS_ERR_BAD_ARG, "dimensions are not padded: rows must divide by " + std::to_string(g.row_divisor) + " and columns by 8 (util_round_csr_matrix_dim)");
    if (rp != (num_rows + g.logical_ob - 1) / g.logical_ob || cp != (num_cols + g.logical_vb - 1) / g.logical_vb)
        return fail(ctx, HS_ERR_BAD_ARG, "partition counts do not match the dimensions and the bank sizes of this context");
    return HS_OK;
}

void loaded(hs_context* ctx, uint32_t rows, uint32_t cols, uint32_t rp, uint32_t cp, std::chrono::steady_clock::time_point t0) {
    ctx->num_rows = rows; ctx->num_cols = cols; ctx->row_parts = rp; ctx->col_parts = cp;
    ctx->y.assign(rows, 0);           // the host zero-initialises y (sw/benchmark.cpp:217-222)
    ctx->matrix_loaded = true;
    ctx->stats.nnz = ctx->indptr[rows];
    ctx->stats.load_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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
        case HS_ERR_UNSUPPORTED: return "not supported by the CPU backend";
        case HS_ERR_NO_MEMORY: return "out of memory";
        default: return "unknown error";
    }
}
const char* hs_last_error(const hs_context* ctx) { return ctx ? ctx->error.c_str() : g_create_error.c_str(); }

int hs_create(hs_context** out, int device_id, int impl, uint32_t ob_bank, uint32_t vb_bank) {
    (void)device_id;
    if (!out) return fail(nullptr, HS_ERR_BAD_ARG, "null context pointer");
    *out = nullptr;
    if (!hisparse::impl_valid(impl)) return fail(nullptr, HS_ERR_BAD_ARG, "impl must be 0 (fixed), 1 (float_pob) or 2 (float_stall)");
    hs_context* c = new (std::nothrow) hs_context;
    if (!c) return fail(nullptr, HS_ERR_NO_MEMORY, "out of memory");
    c->impl = impl;
    c->geom = hisparse::make_geometry(impl, ob_bank ? ob_bank : hisparse::impl_default_ob_bank(impl), vb_bank ? vb_bank : hisparse::impl_default_vb_bank(impl));
    if (c->geom.logical_ob > 0xffffffffull || c->geom.logical_vb > 0xffffffffull || c->geom.logical_ob % c->geom.row_divisor != 0) {
        delete c;
        return fail(nullptr, HS_ERR_BAD_ARG, "ob_bank must make 128*ob_bank a multiple of 128*interleave; bank sizes must fit 32 bits");
    }
    *out = c;
    return HS_OK;
}

int hs_destroy(hs_context* ctx) {
    delete ctx;
    return HS_OK;
}

int hs_load_matrix(hs_context* ctx, const void* const channel[HS_NUM_CHANNELS], const uint64_t n_packets[HS_NUM_CHANNELS], uint32_t num_rows,
                   uint32_t num_cols, uint32_t num_row_partitions, uint32_t num_col_partitions) {
    if (!ctx || !channel || !n_packets) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (int rc = check_dims(ctx, num_rows, num_cols, num_row_partitions, num_col_partitions)) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    using namespace hisparse::dev::detail;
    ctx->matrix_loaded = false;
    Layout L;
    L.g = &ctx->geom;
    L.num_rows = num_rows; L.num_cols = num_cols; L.row_parts = num_row_partitions; L.col_parts = num_col_partitions;
    L.F = ctx->geom.interleave;
    L.sub_width = uint32_t(ctx->geom.logical_vb);
    L.subs_per_cp = 1;
    const uint64_t header_pkts = uint64_t(num_row_partitions) * num_col_partitions * (1 + L.F);
    for (uint32_t c = 0; c < HS_NUM_CHANNELS; ++c) {
        if (!channel[c] && n_packets[c]) return fail(ctx, HS_ERR_BAD_MATRIX, "null channel buffer");
        if (n_packets[c] < header_pkts) return fail(ctx, HS_ERR_BAD_MATRIX, "channel " + std::to_string(c) + " is shorter than its partition headers");
    }
    try {
        // two walks of the image (rows of different physical channels are disjoint): count, then fill in column-partition order
        std::vector<uint32_t> row_nnz(num_rows, 0);
        std::vector<WalkResult> res(size_t(num_row_partitions) * HS_NUM_CHANNELS);
        auto walk_all = [&](auto visit) {
            parallel_for(res.size(), [&](size_t w) {
                const uint32_t rp = uint32_t(w / HS_NUM_CHANNELS), pc = uint32_t(w % HS_NUM_CHANNELS);
                for (uint32_t cp = 0; cp < num_col_partitions && res[w].ok; ++cp) {
                    const uint32_t col_base = uint32_t(uint64_t(cp) * ctx->geom.logical_vb);
                    WalkResult r = walk_channel_partition(L, static_cast<const hisparse::MatPkt*>(channel[pc]), n_packets[pc], pc, rp, cp,
                                                          [&](uint32_t row, uint32_t col, uint32_t val) { visit(row, col_base + col, val); });
                    if (!r.ok) res[w] = r;
                }
            });
            for (const auto& r : res)
                if (!r.ok) return r.error;
            return std::string();
        };
        std::string why = walk_all([&](uint32_t row, uint32_t, uint32_t) { row_nnz[row]++; });
        if (!why.empty()) return fail(ctx, HS_ERR_BAD_MATRIX, why);
        ctx->indptr.assign(size_t(num_rows) + 1, 0);
        for (uint32_t r = 0; r < num_rows; ++r) ctx->indptr[r + 1] = ctx->indptr[r] + row_nnz[r];
        ctx->indices.assign(ctx->indptr[num_rows], 0);
        ctx->words.assign(ctx->indptr[num_rows], 0);
        std::vector<uint32_t> cursor(num_rows, 0);
        why = walk_all([&](uint32_t row, uint32_t col, uint32_t val) {
            const uint64_t at = ctx->indptr[row] + cursor[row]++;
            ctx->indices[at] = col;
            ctx->words[at] = val;
        });
        if (!why.empty()) return fail(ctx, HS_ERR_BAD_MATRIX, why);
    } catch (const std::bad_alloc&) {
        return fail(ctx, HS_ERR_NO_MEMORY, "out of host memory");
    }
    ctx->stats = hs_stats{};
    for (int c = 0; c < HS_NUM_CHANNELS; ++c) ctx->stats.cpsr_bytes += n_packets[c] * sizeof(hisparse::MatPkt);
    loaded(ctx, num_rows, num_cols, num_row_partitions, num_col_partitions, t0);
    return HS_OK;
}

int hs_load_matrix_csr(hs_context* ctx, uint32_t num_rows, uint32_t num_cols, const uint32_t* indptr, const uint32_t* indices, const float* values,
                       uint32_t* padded_rows, uint32_t* padded_cols) {
    if (!ctx || !indptr) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    const Geometry& g = ctx->geom;
    if (num_rows == 0 || num_cols == 0) return fail(ctx, HS_ERR_BAD_ARG, "empty matrix");
    const uint64_t rows = (uint64_t(num_rows) + g.row_divisor - 1) / g.row_divisor * g.row_divisor;
    const uint64_t cols = (uint64_t(num_cols) + hisparse::PACK_SIZE - 1) / hisparse::PACK_SIZE * hisparse::PACK_SIZE;
    if (rows > 0xffffffffull || cols > 0xffffffffull) return fail(ctx, HS_ERR_BAD_ARG, "padded dimensions exceed 32 bits");
    if (indptr[0] != 0) return fail(ctx, HS_ERR_BAD_MATRIX, "CSR indptr must start at 0");
    for (uint32_t r = 0; r < num_rows; ++r)
        if (indptr[r + 1] < indptr[r]) return fail(ctx, HS_ERR_BAD_MATRIX, "CSR indptr decreases at row " + std::to_string(r));
    const uint64_t nnz = indptr[num_rows];
    if (nnz && (!indices || !values)) return fail(ctx, HS_ERR_BAD_ARG, "CSR arrays missing");
    for (uint64_t e = 0; e < nnz; ++e)
        if (indices[e] >= num_cols) return fail(ctx, HS_ERR_BAD_MATRIX, "CSR column index outside the matrix");
    const auto t0 = std::chrono::steady_clock::now();
    ctx->matrix_loaded = false;
    ctx->indptr.assign(size_t(rows) + 1, nnz);
    for (uint32_t r = 0; r <= num_rows; ++r) ctx->indptr[r] = indptr[r];
    ctx->indices.assign(indices, indices + nnz);
    ctx->words.resize(nnz);
    for (uint64_t e = 0; e < nnz; ++e) {
        if (ctx->impl == HS_IMPL_FIXED) ctx->words[e] = hisparse::q8_24_raw_from_double(double(values[e]));     // csr_matrix_convert_from_float
        else std::memcpy(&ctx->words[e], &values[e], 4);
    }
    ctx->stats = hs_stats{};
    loaded(ctx, uint32_t(rows), uint32_t(cols), uint32_t((rows + g.logical_ob - 1) / g.logical_ob), uint32_t((cols + g.logical_vb - 1) / g.logical_vb), t0);
    if (padded_rows) *padded_rows = uint32_t(rows);
    if (padded_cols) *padded_cols = uint32_t(cols);
    return HS_OK;
}

int hs_load_vector(hs_context* ctx, const void* packed_x, uint32_t num_cols) {
    if (!ctx || !packed_x) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (ctx->matrix_loaded && num_cols != ctx->num_cols) return fail(ctx, HS_ERR_BAD_ARG, "vector length must equal the padded column count");
    ctx->x.assign(static_cast<const uint32_t*>(packed_x), static_cast<const uint32_t*>(packed_x) + num_cols);
    ctx->vector_loaded = true;
    return HS_OK;
}

int hs_run(hs_context* ctx) {
    if (int rc = check_ready(ctx)) return rc;
    multiply_rows(*ctx, 0, ctx->num_rows, ctx->y);
    return HS_OK;
}

int hs_run_partition(hs_context* ctx, uint32_t row_part_id, uint32_t part_len) {
    if (int rc = check_ready(ctx)) return rc;
    if (row_part_id >= ctx->row_parts) return fail(ctx, HS_ERR_BAD_ARG, "row_part_id out of range");
    const uint64_t lo = uint64_t(row_part_id) * ctx->geom.logical_ob, hi = std::min<uint64_t>(lo + ctx->geom.logical_ob, ctx->num_rows);
    if (part_len != (hi - lo) / hisparse::NUM_HBM_CHANNELS)
        return fail(ctx, HS_ERR_BAD_ARG, "part_len must be the partition's rows / 16 (sw/benchmark.cpp:301-322): expected " + std::to_string((hi - lo) / hisparse::NUM_HBM_CHANNELS));
    multiply_rows(*ctx, uint32_t(lo), uint32_t(hi), ctx->y);
    return HS_OK;
}

int hs_sync(hs_context* ctx) { return ctx ? HS_OK : HS_ERR_BAD_ARG; }

int hs_read_result(hs_context* ctx, void* packed_y, uint32_t num_rows) {
    if (!ctx || !packed_y) return fail(ctx, HS_ERR_BAD_ARG, "null argument");
    if (!ctx->matrix_loaded) return fail(ctx, HS_ERR_NOT_LOADED, "hs_load_matrix has not been called");
    if (num_rows != ctx->num_rows) return fail(ctx, HS_ERR_BAD_ARG, "result length must equal the padded row count");
    std::memcpy(packed_y, ctx->y.data(), size_t(num_rows) * 4);
    return HS_OK;
}

int hs_get_stats(const hs_context* ctx, hs_stats* stats) {
    if (!ctx || !stats) return HS_ERR_BAD_ARG;
    *stats = ctx->stats;
    return HS_OK;
}

int hs_time_runs(hs_context* ctx, int warmup, int runs, float* total_ms, float* kernel_ms) {
    if (int rc = check_ready(ctx)) return rc;
    if (warmup < 0 || runs <= 0) return fail(ctx, HS_ERR_BAD_ARG, "need warmup >= 0 and runs > 0");
    for (int i = 0; i < warmup; ++i) multiply_rows(*ctx, 0, ctx->num_rows, ctx->y);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < runs; ++i) multiply_rows(*ctx, 0, ctx->num_rows, ctx->y);
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (total_ms) *total_ms = ms;
    if (kernel_ms) *kernel_ms = ms;
    return HS_OK;
}

int hs_run_batch(hs_context* ctx, uint32_t steps) {
    for (uint32_t i = 0; i < steps; ++i)
        if (int rc = hs_run(ctx)) return rc;
    return HS_OK;
}
int hs_time_kernel(hs_context* ctx, int warmup, int runs, float* kernel_ms) { return hs_time_runs(ctx, warmup, runs, nullptr, kernel_ms); }

// ---- not part of this backend: device-memory hooks and the extensions ---------------------------------------------------------
#define HS_CPU_UNSUPPORTED(ctx) return fail(ctx, HS_ERR_UNSUPPORTED, std::string(__func__) + " is not part of the CPU backend (libhisparse_cpu.so)")
int hs_set_option(hs_context* ctx, const char*, const char*) { HS_CPU_UNSUPPORTED(ctx); }
int hs_set_stream(hs_context* ctx, void*) { HS_CPU_UNSUPPORTED(ctx); }
int hs_get_stream(hs_context* ctx, void**) { HS_CPU_UNSUPPORTED(ctx); }
int hs_device_vector(hs_context* ctx, void**) { HS_CPU_UNSUPPORTED(ctx); }
int hs_device_result(hs_context* ctx, void**) { HS_CPU_UNSUPPORTED(ctx); }
int hs_bind_device_vector(hs_context* ctx, const void*) { HS_CPU_UNSUPPORTED(ctx); }
int hs_bind_device_result(hs_context* ctx, void*) { HS_CPU_UNSUPPORTED(ctx); }
int hs_push_result(hs_context* ctx, void* const*, uint32_t, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_feedback(hs_context* ctx, uint32_t, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_iterate(hs_context* ctx, uint32_t, uint32_t, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_load_matrix_csc(hs_context* ctx, const uint32_t*, const uint32_t*, const uint32_t*, uint32_t, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_spmspv(hs_context* ctx, const hs_idx_val*, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_spmspv_device(hs_context* ctx, const hs_idx_val*, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_read_spmspv_result(hs_context* ctx, void*, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_spmspv_status(hs_context* ctx, uint32_t*, void**) { HS_CPU_UNSUPPORTED(ctx); }
int hs_spmm_device(hs_context* ctx, const void*, uint64_t, void*, uint64_t, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_spmm(hs_context* ctx, const void*, uint32_t, uint32_t, void*, uint32_t) { HS_CPU_UNSUPPORTED(ctx); }
int hs_debug_read_tiles(hs_context* ctx, void*, uint64_t, void*, void*) { HS_CPU_UNSUPPORTED(ctx); }
int hs_debug_read_mfma_image(hs_context* ctx, void*, uint64_t, uint64_t*) { HS_CPU_UNSUPPORTED(ctx); }

}  // extern "C"
