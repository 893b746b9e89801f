// spmv_kernels.h — launch interface of the gfx950 kernel (spmv_kernels.hip).
#ifndef HISPARSE_SPMV_KERNELS_H_
#define HISPARSE_SPMV_KERNELS_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>

#include "stream_tiles.h"

namespace hisparse {
namespace dev {

struct SpmvLaunch {
    const uint8_t* image;         // element streams
    const Block* blocks;          // workgroup g starts at blocks[g] and follows Block::next
    const Unit* units;
    const uint32_t* part_heads;   // [row partition][workgroup] first block or kNoBlock (only read when row_part_filter >= 0)
    const uint32_t* x;            // packed vector words, num_cols
    uint32_t* out;                // packed result words: y itself (one column slice) or slices x num_rows partials
    int32_t row_part_filter;      // -1: every row partition
    uint32_t ring_buffers;        // x sub-tile buffers in the LDS ring (2..4)
    uint32_t format;              // StreamFormat of `image` (stream_tiles.h): PAIRS chunks, DELTA records or BITMAP rows
    uint32_t num_cols;            // length of x in words (BITMAP: the x reads of a row's last group are range-checked against it)
    uint32_t num_workgroups;
    uint32_t lds_bytes;
    uint32_t bitmap_x_groups = 0; // BITMAP: lds_bytes ends with room for this many 64-column groups of x (0: x is read through L2)
    // column-sliced plans, hs_run after hs_run: the PREVIOUS step's partial vectors, added up into carry_y by this launch's workgroups before
    // they start on their own blocks (spmv_device.h: CarriedCombine); carry_partial == nullptr: nothing to carry
    const uint32_t* carry_partial = nullptr;
    uint32_t* carry_y = nullptr;
    uint32_t carry_rows = 0, carry_slices = 0;
    bool stream_resident = false; // the image is small enough to stay in the 256 MiB Infinity Cache from one step to the next: SWEEP stream loads WITHOUT the non-temporal hint (spmv_sweep.hip)
    bool light = false;           // the LIGHT plan (StreamTiles::light): a PAIRS image consumed by spmv_light_kernel, 256-thread workgroups, lds_bytes = spmv_light_lds_bytes
};

// HISPARSE_ABLATE / HISPARSE_DEPTH (environment): profiling switches of libhisparse_hip_prof.so (-DHISPARSE_PROFILING), read per launch.
// In the product library they select nothing: profiling_switches returns false when either is set to a non-default value, and the
// launch functions then fail (hs_api.cpp reports HS_ERR_BAD_ARG through profiling_switch_error) -- an inherited environment variable
// must never change what a production SpMV computes.
bool profiling_switches(int& ablate, int& depth);
bool profiling_depth_given();
const char* profiling_switch_error();      // nullptr, or why the product library will not launch in this environment

// Dynamic LDS a launch needs for blocks of at most `max_block_rows` rows and a ring of `ring_buffers` x buffers.
uint32_t spmv_lds_bytes(uint32_t max_block_rows, uint32_t ring_buffers, uint32_t format = kFormatPairs);
uint32_t spmv_light_lds_bytes(uint32_t max_block_rows);       // the LIGHT kernel: accumulators only
// One-time per device: allow the kernels to use up to `lds_bytes` of dynamic LDS.
hipError_t configure_spmv_kernels(uint32_t lds_bytes);
// The SpMV kernel: row-owner workgroups, x sub-tiles double-buffered in LDS, no global atomics.
hipError_t launch_spmv(bool is_float, const SpmvLaunch& a, hipStream_t stream);
// Fused SpMM over a BITMAP image (spmm_bitmap.hip): `vectors` (2 or 4) columns of X at once, one column slice only.
struct SpmmLaunch {
    const uint8_t* image;
    const Block* blocks;
    const Unit* units;
    const uint32_t* x;            // column j of X at x + j * ldx words
    uint64_t ldx;
    uint32_t* x_interleaved;      // scratch, num_cols * vectors words
    uint32_t* y;                  // column j of Y at y + j * ldy words
    uint64_t ldy;
    uint32_t vectors;
    uint32_t num_cols;
    uint32_t num_workgroups;
    uint32_t max_block_rows;
};
uint32_t spmm_bitmap_max_block_rows(bool is_float, uint32_t vectors);   // rows per block whose accumulators still fit the LDS
hipError_t launch_spmm_bitmap(bool is_float, const SpmmLaunch& a, hipStream_t stream);
// SpMM on the matrix engine over the second image of a float BITMAP matrix (spmm_mfma.hip; stream_tiles.h: MfmaImage): 16 columns of X per
// call.  Leaves y untouched (and sets *flag) when X holds a non-finite word: 0.0 x inf inside an MFMA would poison rows that do not touch
// that column -- the finish kernel (spmm_finish_kernel) then computes those 16 columns from the stored elements only, the way the PEs would.
struct SpmmMfmaLaunch {
    const uint32_t* words;        // the MfmaImage on the device
    uint64_t offsets_word, values_word;
    uint32_t tiles, groups, chunk, chunks;
    const uint32_t* x;            // column j of X at x + j * ldx words
    uint64_t ldx;
    uint32_t* x_interleaved;      // scratch: spmm_mfma_x_words(groups) words
    float* partial;               // scratch: spmm_mfma_partial_words(tiles, chunks) floats
    uint32_t* flag;               // scratch: one word (zero at allocation): becomes `call` when X holds a non-finite word
    uint32_t call;                // a number no earlier call on this context has used (never 0)
    uint32_t* y;                  // column j of Y at y + j * ldy words
    uint64_t ldy;
    uint32_t num_rows, num_cols;
    uint32_t vectors = 16;        // columns of X / Y this pass really has (1 .. 16): the others are zero vectors, their results are not written
};
size_t spmm_mfma_x_words(uint32_t groups);
size_t spmm_mfma_partial_words(uint32_t tiles, uint32_t chunks);
hipError_t launch_spmm_mfma(const SpmmMfmaLaunch& a, hipStream_t stream);
// SpMM over a SWEEP image planned for it (spmm_sweep.hip; hs_set_option "spmm_vectors" = 4 before the load): up to four columns of X per call.
struct SpmmSweepLaunch {
    const uint8_t* image;
    const Block* blocks;
    const uint32_t* x;            // column j of X at x + j * ldx words
    uint64_t ldx;
    uint32_t* x4;                 // scratch: num_cols x 4 words, 16-byte aligned ([column][vector])
    uint32_t* out;                // [column slice][vector][row] words: one slice = the four result columns back to back
    uint32_t vectors;             // 1 .. 4 columns really there (the others are zero vectors)
    uint32_t num_rows, num_cols;
    uint32_t num_workgroups;
    uint32_t max_block_rows;
};
uint32_t spmm_sweep_lds_bytes(uint32_t max_block_rows, bool is_float);
hipError_t configure_spmm_sweep_kernels(uint32_t lds_bytes);
hipError_t launch_spmm_sweep(bool is_float, const SpmmSweepLaunch& a, hipStream_t stream);
// BITMAP images (spmv_bitmap.hip); launch_spmv forwards to it when a.format == kFormatBitmap.
hipError_t configure_bitmap_kernels(uint32_t lds_bytes);
hipError_t launch_spmv_bitmap(bool is_float, const SpmvLaunch& a, hipStream_t stream);
// SWEEP images (spmv_sweep.hip); launch_spmv forwards to it when a.format == kFormatSweep.
uint32_t spmv_sweep_lds_bytes(uint32_t max_block_rows, bool is_float);       // accumulators only: doubles / 32-bit sums + a carry bit per row
hipError_t configure_sweep_kernels(uint32_t lds_bytes);
hipError_t launch_spmv_sweep(bool is_float, const SpmvLaunch& a, hipStream_t stream);
// Column-sliced matrices only: y[r] = (saturating / fp32) sum of the `slices` partial vectors, rows [row_lo, row_hi);
// with x_fb also x_fb[r] = scale (*) y[r] (+) shift for r < n_fb (hs_iterate's feedback folded into the same launch).
hipError_t launch_combine_slices(bool is_float, const uint32_t* partial, uint32_t* y, uint32_t num_rows, uint32_t slices, uint32_t row_lo,
                                 uint32_t row_hi, hipStream_t stream, uint32_t* x_fb = nullptr, uint32_t n_fb = 0, uint32_t scale = 0,
                                 uint32_t shift = 0);

// SpMSpV extension (spmspv.hip): y = A x for x given as x_count IDX_VAL_T pairs ON THE DEVICE over a CSC matrix.  Two launches: EXPAND writes
// the selected columns' products straight into per-row-block BINS (an LDS histogram per workgroup of 64 entries, one global atomic per workgroup
// and non-empty bin claims the room), ACCUMULATE has one workgroup per row block add ITS bin in LDS and write its rows of y.  No scan, no sort,
// no host synchronisation.  Scratch, owned by the caller (hs_api.cpp):
struct hs_idx_val_dev { uint32_t index, val; };      // == hs_idx_val (hisparse_hip.h), IDX_VAL_T of spmv/libfpga/common.h:54
struct SpmspvScratch {
    uint32_t* keys = nullptr;                   // [nnz] rows of the products, bin by bin
    uint32_t* vals = nullptr;                   // [nnz] product words
    uint32_t* bin_base = nullptr;               // [bins + 1]: bin b = [bin_base[b], bin_base[b + 1]) = as many entries as the matrix has non-zeros in row block b
    uint32_t* cursors = nullptr;                // [bins]: products in the bin (zero between calls: the accumulate kernel re-arms them)
    uint32_t* overflow = nullptr;               // one word: a bin was asked for more than it holds (x named columns more than once)
    uint64_t capacity = 0;                      // = the matrix's non-zeros
};
uint32_t spmspv_block_bits(uint32_t num_rows);  // log2 of the rows per block (13, less for matrices of few rows)
uint32_t spmspv_bins(uint32_t num_rows);
uint32_t spmspv_max_bins();
// add_to_y: y += instead of y = (further passes of one call).
hipError_t launch_spmspv(bool is_float, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, const hs_idx_val_dev* x_entries,
                         uint32_t x_count, uint32_t num_rows, uint32_t num_cols, const SpmspvScratch& scratch, bool add_to_y, uint32_t* y, hipStream_t stream);
// x_dense[0, num_cols) = 0, then x_dense[index] = val for every entry (hs_spmspv's dense dispatch)
hipError_t launch_spmspv_scatter_x(const hs_idx_val_dev* x_entries, uint32_t x_count, uint32_t num_cols, uint32_t* x_dense, hipStream_t stream);

// Multi-GPU gather without a collective: y[0, words) into n_dst <= kMaxPushTargets other buffers (peers' memory over xGMI) with plain stores.
constexpr uint32_t kMaxPushTargets = 8;
hipError_t launch_push_result(const uint32_t* y, void* const* dst, uint32_t n_dst, uint32_t words, hipStream_t stream);

// Iterative callers: x[i] = scale (*) y[i] (+) shift, i < n, in Q8.24 (AP_RND, AP_SAT) or fp32 arithmetic.
hipError_t launch_feedback(bool is_float, const uint32_t* y, uint32_t* x, uint32_t n, uint32_t scale, uint32_t shift, hipStream_t stream);

}  // namespace dev
}  // namespace hisparse

#endif  // HISPARSE_SPMV_KERNELS_H_
