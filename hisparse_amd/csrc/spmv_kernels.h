// spmv_kernels.h — launch interface of the gfx950 kernels (spmv_kernels.hip).
#ifndef HISPARSE_SPMV_KERNELS_H_
#define HISPARSE_SPMV_KERNELS_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "stream_tiles.h"

namespace hisparse {
namespace dev {

struct SpmvLaunch {
    const uint8_t* image;      // stream tiles
    const Piece* pieces;
    const uint32_t* wg_first;  // num_workgroups + 1 entries
    const uint32_t* x;         // packed vector words, num_cols
    void* accum;               // fixed: uint64_t[rows + slack] (zero on entry); float: the packed y itself
    uint32_t num_cols;
    uint32_t tile_cols;        // LOGICAL_VB_SIZE
    uint32_t row_stride;       // 128 * F
    int32_t row_part_filter;   // -1: every row partition
    uint32_t num_workgroups;
    uint32_t lds_bytes;
};

// One-time per process/device: allow the kernels to use up to `lds_bytes` of dynamic LDS.
hipError_t configure_spmv_kernels(uint32_t lds_bytes);
// The dominant kernel: streams the tiles, gathers x from LDS, accumulates rows.
hipError_t launch_spmv_stream(bool is_float, const SpmvLaunch& a, hipStream_t stream);
// fixed point only: y[r] = min(accum[r], 2^32-1), accum[r] = 0 for r in [row_lo, row_hi).
hipError_t launch_finalize_fixed(uint64_t* accum, uint32_t* y, uint32_t row_lo, uint32_t row_hi, hipStream_t stream);

}  // namespace dev
}  // namespace hisparse

#endif  // HISPARSE_SPMV_KERNELS_H_
