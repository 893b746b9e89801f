// gpu_tiles.h — the data-touching passes of the load-time re-tiling, on the GPU.
//
// stream_tiles.cpp keeps ALL the planning (format, tile plan, row ranges, column slices, stream layout, workgroup assignment: small,
// sequential, a few milliseconds) and asks a "source" for the six things that touch every non-zero:
//     rows' non-zero counts  ->  (row range, sub-tile) counts  ->  every unit's elements sorted by position
//     ->  DELTA slot counts / OWNER shares  ->  the emitted image.
// The host source is the multi-threaded code of round 1 (three walks of the CPSR image, per-unit std::sort, emit loops).  GpuTiler is
// the same on the device: the 16 channel buffers are uploaded once, one thread per SEGMENT of a lane stream replays the reference loader's decode
// (running row index = sum of in-band markers, spmv_cluster.h:73-98 / fp :95-117; sw/data_formatter.h:410,432 for the row <-> lane
// mapping), (unit, position) keys are radix-sorted with hipCUB, and one emit kernel per format writes the image straight into the
// buffer the SpMV kernel will read -- the image never exists on the host.  The host builder stays as the byte-for-byte checker
// (tests/test_gpu_retile.py compares image, Block[] and Unit[] of both).
#ifndef HISPARSE_GPU_TILES_H_
#define HISPARSE_GPU_TILES_H_

#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <string>
#include <vector>

#include "tiles_common.h"

namespace hisparse {
namespace dev {

hipError_t warm_gpu_tiler();

class GpuTiler {
  public:
    GpuTiler(const detail::Layout& layout, const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS], hipStream_t stream);
    // the source is a CSR matrix (hs_load_matrix_csr): rows come from indptr, the element passes run one thread per kCsrSegment
    // consecutive non-zeros, value words are made on the device (csr_matrix_convert_from_float, sw/data_loader.h:76-84)
    GpuTiler(const detail::Layout& layout, const CsrView& csr, hipStream_t stream);
    ~GpuTiler();
    GpuTiler(const GpuTiler&) = delete;
    GpuTiler& operator=(const GpuTiler&) = delete;

    const std::string& error() const { return error_; }
    // false + error(): not a valid CPSR image (same conditions the host walk reports) or a HIP failure.
    // non-zeros per row; total
    bool count_rows(std::vector<uint32_t>& row_nnz, uint64_t& nnz);
    // totals per (row range, column partition, sub-tile): cnt[(b * CP + cp) * S + s]
    bool count_tiles(const std::vector<uint32_t>& block_of_row, uint32_t num_ranges, std::vector<uint32_t>& cnt);
    // Every unit's elements sorted by position, unit u at [plans[u].scratch, + plans[u].n) of the device arrays.  unit_of as in
    // stream_tiles.cpp ((range, cp, s) -> unit or 0xffffffff); row0 of every range.  `duplicates` = some (row, column) occurs twice
    // (the host's order among equal positions is by value word; the caller falls back to the host builder then).
    bool sort_elements(const std::vector<uint32_t>& block_of_row, const std::vector<uint32_t>& range_row0, const std::vector<uint32_t>& unit_of,
                       const std::vector<detail::UnitPlan>& plans, bool& duplicates);
    // DELTA: slots (elements + bridges) of every unit
    bool delta_slots(std::vector<detail::UnitPlan>& plans);
    // OWNER: plans[u].own_begin / own_row / own_last: every unit's elements cut into the 14 wavefronts' shares
    // (detail::balanced_owner_shares; max_span = kOwnerShareRows for OWNER24, 0xffffffff otherwise)
    bool owner_shares(std::vector<detail::UnitPlan>& plans, uint32_t max_span);
    // The image (image_bytes + slack, zero-filled first).  format: the final StreamFormat; block_of_unit / blocks: pre-reorder indices.
    bool emit(StreamFormat format, uint64_t image_bytes, uint64_t slack_bytes, const std::vector<detail::UnitPlan>& plans,
              const std::vector<uint32_t>& block_of_unit, const std::vector<Block>& blocks, bool is_float);
    // ---- BITMAP (bitmap_tiles.cpp plans; masks, values, run heads and the matrix-engine image are made here) -------------------------
    // non-zeros per (row, column slice of groups): cnt[row * slices + k]; slice k = groups [k GR / slices, (k + 1) GR / slices)
    bool bitmap_slice_counts(uint32_t slices, uint32_t groups_per_row, std::vector<uint32_t>& cnt);
    struct BitmapBlock {            // one (row range, column slice); layout as bitmap_tiles.cpp lays it out
        uint32_t row0, nrows, gs0, GS, pieces, stride;
        uint64_t mask_word0;        // 8-byte word index of the block's first mask
        uint64_t prefix0;           // first entry of the block in the per-mask prefix array (nrows x stride entries per block)
        uint32_t piece_cut[kBitmapWaves + 1];   // first group of every piece of a row (pieces + 1 entries; the shares are weighted, bitmap_tiles.cpp)
    };
    struct BitmapRun {              // a wavefront's run: where its masks start (8-byte word index) and in the prefix array; groups in its first row
        uint64_t mask_word, prefix_at;
        uint32_t steps, pad;        // pad = 1: the run starts inside a row (its value offset needs the row's prefix)
    };
    // Builds the image (image_bytes + slack, zero-filled first): every element sets its bit (a bit found set = the (row, column) occurs
    // twice -> duplicates, nothing else is valid then), a per-row prefix count of the masks gives every element its place among the
    // compacted values.  run_prefix[r] = values of the run's row in front of its first group; run_heads[r * 32 ..] = its first 32 masks.
    // mfma (may be null; its layout fields filled by the caller): the second image, left on the device (release_mfma).
    bool bitmap_emit(uint32_t slices, uint32_t groups_per_row, const std::vector<uint32_t>& range_of_row, const std::vector<BitmapBlock>& blocks,
                     const std::vector<uint64_t>& row_value_base, uint64_t image_bytes, uint64_t slack_bytes, const std::vector<BitmapRun>& runs,
                     std::vector<uint32_t>& run_prefix, std::vector<uint64_t>& run_heads, const MfmaImage* mfma, bool& duplicates);
    uint8_t* release_mfma() { uint8_t* p = d_mfma_; d_mfma_ = nullptr; return p; }
    // ---- SWEEP (sweep_tiles.cpp plans: row ranges, column slices, the layout of streams and chunk-base tables; what touches every non-zero is here) ----
    // non-zeros per 128-byte line of x (32 columns): the slice boundaries are cut at equal modelled cost
    bool sweep_line_counts(uint32_t lines, std::vector<uint64_t>& line_nnz);
    // Every element keyed (block = range x slices + slice) << 48 | column << 16 | local row and radix-sorted; block b then holds the sorted
    // elements [block_start[b], block_start[b + 1]).  unsupported = some (row, column) occurs twice (the host orders those by value word) or a
    // 64-element chunk would span more than 65535 columns (the host cuts such chunks short): the caller builds on the host then.
    bool sweep_sort(const std::vector<uint32_t>& range_of_row, const std::vector<uint32_t>& range_row0, const std::vector<uint32_t>& slice_col,
                    uint32_t num_blocks, std::vector<uint64_t>& block_start, bool& unsupported);
    struct SweepBlock {             // one (row range, column slice) as sweep_tiles.cpp lays it out
        uint64_t first;             // its first sorted element
        uint64_t stream_at, table_at;   // byte offsets of its chunks / its chunk-base table in the image
        uint64_t chunk0;            // chunks of all blocks in front of it
        uint32_t count, steps, nrows, pad_col;
    };
    bool sweep_emit(const std::vector<SweepBlock>& blocks, uint64_t image_bytes, uint64_t slack_bytes);
    // hands the device image over (hipFree by the new owner)
    uint8_t* release_image() { uint8_t* p = d_image_; d_image_ = nullptr; return p; }

  private:
    bool fail(const std::string& what);
    bool check(hipError_t e, const char* what);
    bool upload_channels();
    bool upload_csr(std::vector<uint32_t>& row_nnz);
    bool decode_error(const char* pass);

    detail::Layout L_;
    Geometry geom_;
    const void* const* channel_;
    const uint64_t* n_packets_;
    hipStream_t stream_;
    std::string error_;

    const CsrView* csr_ = nullptr;         // CSR source (then channel_ / n_packets_ are null)
    uint32_t* d_indptr_ = nullptr;         // CSR source: indptr of the PADDED matrix (num_rows + 1), indices, values
    uint32_t* d_indices_ = nullptr;
    float* d_values_ = nullptr;
    uint8_t* d_channels_ = nullptr;        // the 16 channel buffers back to back
    void* d_groups_ = nullptr;             // StreamGroup[num_groups_]: the 8 lane streams of one (partition, virtual channel)
    uint32_t num_groups_ = 0;
    uint32_t total_slots_ = 0;             // segments of all lane streams (gpu_tiles.hip: kSegment entries each)
    uint64_t* d_advance_ = nullptr;        // per segment: marker counts in front of it (exclusive scan)
    uint64_t* d_base_ = nullptr;           // per segment: non-zeros in front of it (exclusive scan; [total_slots_] = all)
    uint32_t* d_scalar_ = nullptr;         // error / flag words
    uint32_t* d_block_of_row_ = nullptr;
    uint64_t total_ = 0;                   // non-zeros
    uint64_t* d_keys_ = nullptr;           // sorted: unit << 28 | position
    uint32_t* d_vals_ = nullptr;           // sorted value words
    uint64_t* d_bridges_ = nullptr;        // DELTA: inclusive scan of the bridge slots in front of every element
    uint8_t* d_image_ = nullptr;
    uint8_t* d_mfma_ = nullptr;            // BITMAP, float modes: the matrix-engine image (stream_tiles.h: MfmaImage)
};

}  // namespace dev
}  // namespace hisparse

#endif  // HISPARSE_GPU_TILES_H_
