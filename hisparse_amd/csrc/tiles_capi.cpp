// tiles_capi.cpp — C-ABI introspection of the load-time re-tiling (no GPU needed).
// Lets tests inspect exactly what hs_load_matrix uploads; declared in include/hisparse_hip.h.
#include <cstring>
#include <new>
#include <string>

#include "hisparse_hip.h"
#include "stream_tiles.h"

struct hs_tiles {
    hisparse::dev::StreamTiles t;
};

namespace {
thread_local std::string g_tiles_error;
}

extern "C" {

const char* hs_tiles_last_error(void) { return g_tiles_error.c_str(); }

int hs_tiles_build(const void* const channel[HS_NUM_CHANNELS], const uint64_t n_packets[HS_NUM_CHANNELS], int impl, uint32_t ob_bank,
                   uint32_t vb_bank, uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions,
                   uint32_t num_col_partitions, uint32_t max_workgroups, hs_tiles** out) {
    if (!channel || !n_packets || !out || !hisparse::impl_valid(impl) || ob_bank == 0 || vb_bank == 0) {
        g_tiles_error = "bad argument";
        return HS_ERR_BAD_ARG;
    }
    hs_tiles* h = new (std::nothrow) hs_tiles;
    if (!h) return HS_ERR_NO_MEMORY;
    try {
        if (!hisparse::dev::build_stream_tiles(channel, n_packets, hisparse::make_geometry(impl, ob_bank, vb_bank), num_rows, num_cols,
                                               num_row_partitions, num_col_partitions, max_workgroups, h->t, g_tiles_error)) {
            delete h;
            return HS_ERR_BAD_MATRIX;
        }
    } catch (const std::bad_alloc&) {
        delete h;
        g_tiles_error = "out of memory";
        return HS_ERR_NO_MEMORY;
    }
    *out = h;
    return HS_OK;
}

int hs_tiles_info(const hs_tiles* h, uint64_t* image_bytes, uint32_t* num_blocks, uint32_t* num_units, uint32_t* num_workgroups,
                  uint32_t* max_block_rows, uint64_t* nnz, uint64_t* elements, uint32_t* col_slices, uint32_t* ring_buffers,
                  uint32_t* stream_format) {
    if (!h) return HS_ERR_BAD_ARG;
    if (image_bytes) *image_bytes = h->t.image.size();
    if (num_blocks) *num_blocks = uint32_t(h->t.blocks.size());
    if (num_units) *num_units = uint32_t(h->t.units.size());
    if (num_workgroups) *num_workgroups = h->t.num_workgroups;
    if (max_block_rows) *max_block_rows = h->t.max_block_rows;
    if (nnz) *nnz = h->t.nnz;
    if (elements) *elements = h->t.elements;
    if (col_slices) *col_slices = h->t.col_slices;
    if (ring_buffers) *ring_buffers = h->t.ring_buffers;
    if (stream_format) *stream_format = h->t.format;
    return HS_OK;
}

int hs_tiles_copy(const hs_tiles* h, void* image, void* blocks, void* units, uint32_t* wg_first, uint32_t* block_order) {
    if (!h) return HS_ERR_BAD_ARG;
    if (image && !h->t.image.empty()) std::memcpy(image, h->t.image.data(), h->t.image.size());
    if (blocks && !h->t.blocks.empty()) std::memcpy(blocks, h->t.blocks.data(), h->t.blocks.size() * sizeof(hisparse::dev::Block));
    if (units && !h->t.units.empty()) std::memcpy(units, h->t.units.data(), h->t.units.size() * sizeof(hisparse::dev::Unit));
    if (wg_first) std::memcpy(wg_first, h->t.wg_first.data(), h->t.wg_first.size() * sizeof(uint32_t));
    if (block_order && !h->t.block_order.empty()) std::memcpy(block_order, h->t.block_order.data(), h->t.block_order.size() * sizeof(uint32_t));
    return HS_OK;
}

void hs_tiles_free(hs_tiles* h) { delete h; }

}  // extern "C"
