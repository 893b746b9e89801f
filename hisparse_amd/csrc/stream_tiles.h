// stream_tiles.h — device-private re-tiling of a CPSR image for the gfx950 SpMV kernel.
//
// The FPGA consumes each HBM channel as 8 lock-step lane streams whose row index is a running sum of
// in-band markers (spmv/libfpga/spmv_cluster.h:34-107; fp variant spmv-fp/libfpga/spmv_cluster.h:39-129):
// 128 sequential streams per (row partition, column partition), each padded to the longest lane of its
// channel (sw/data_formatter.h:421-428, sw/benchmark.cpp:148-163).  That is far too few, too
// sequential and too unbalanced for 256 CUs.  What DOES carry over is the architecture:
//
//   FPGA (16 clusters)                                       MI355X (256 workgroups)
//   -------------------------------------------------------  -------------------------------------------
//   a cluster's 8 PEs own a fixed set of rows; their sums    a workgroup owns a contiguous ROW BLOCK
//   live in on-chip output banks (pe.h:121-135)              (<= 4095 rows); the sums live in its LDS
//   the current column partition of x sits in 8 on-chip      the current x SUB-TILE (<= 8192 columns) sits
//   vector banks, double-buffered (vecbuf_access_unit.h)     in LDS, in a ring of four buffers
//   the matrix streams past, one packet per cycle            the block's non-zeros stream past as
//   (spmv_cluster.h:73-98)                                   coalesced 8- or 6-byte elements
//
// So at hs_load_matrix time the CPSR image is decoded ONCE on the host and re-cut:
//   * rows are split into row blocks of roughly equal non-zero count (never across a row partition);
//   * the non-zeros of a block are grouped into UNITS, one per x sub-tile that the block touches
//     (a sub-tile is a <= 8192-column slice of one column partition);
//   * when x is large relative to the work per workgroup, the columns are additionally cut into 2-8
//     COLUMN SLICES (sub-tiles dealt round-robin): a block is then (row range, slice), row ranges get
//     proportionally longer, every workgroup pulls only 1/slices of x through its CU, and a small
//     combine pass adds the per-slice partial results (saturating sums compose: DESIGN.md §4);
//   * a unit's elements are sorted by position (local_row * 8192 + local_col) and dealt to the 14 consumer
//     wavefronts of the workgroup in chunks of 64 slots, in one of two STREAM FORMATS chosen per matrix:
//
//     PAIRS (8 bytes per element): element = { u32 value word, u32 (local_row << 16 | local_col) }, one
//       512-byte chunk per wavefront step, the chunks of a block stored in dealing order so that the
//       workgroup sweeps one contiguous region.  Normally lane l of chunk c takes element l*chunks + c
//       (neighbouring lanes far apart: no same-row LDS atomics in one instruction).  Blocks of <= 32 rows
//       (pruned-NN layers: 512 rows x 16 K non-zeros) would put all 64 lanes on the same one or two
//       accumulators; there the unit is dealt linearly, so a chunk almost always holds ONE row and the
//       wavefront adds it up in registers.  Bytes read per SpMV = the reference's "8 bytes per non-zero"
//       throughput definition (sw/benchmark.cpp:312-314) plus < 2 % chunk padding.
//
//     DELTA (6 bytes per slot): lane l of wavefront w owns run_len consecutive slots of the sorted unit -- the runs are dealt
//       lane-major (run l * 14 + w), so the lanes of one wavefront sit a 64th of the unit apart, like the lanes of a PAIRS chunk
//       (round 5; a contiguous 1/14 of the unit per wavefront put all 64 lanes on ONE accumulator wherever a graph has hub rows).
//       A slot = a u32 value word + a u16 GAP: the distance from the lane's previous position.  A 768-byte record holds TWO
//       consecutive slots of every lane (kRecordBytes below).  Every (unit, wavefront) run starts with a HEAD slot whose value
//       words are the lanes' absolute start positions.  Gap 0xffff = BRIDGE: no element, advance 65535 (value word 0);
//       it carries a lane across distances that do not fit 16 bits and pads the float formats' tails
//       (fixed-point tails are padded with gap 0 / value 0).  Each wavefront's records are contiguous.
//       On ogbl-ppa this is 6.85 bytes per non-zero all in (heads, bridges, padding, dead slots) instead of 8.03.
//
//     DELTA is tried when the mean position gap rows*cols/nnz lies in [kDeltaMinMeanGap, kDeltaMaxMeanGap] (hyper-sparse matrices
//     would need a bridge for every other gap) and kept when, after the sort, it needs at most 5 % bridge slots AND saves more
//     than kDeltaMinSavedBytes of stream against PAIRS: its slots cost more instructions and a head record per unit and
//     wavefront, which only pays when the stream bounds the kernel (stream_tiles.cpp has the measurements).  Inside a DELTA matrix,
//     blocks whose rows are long (block gap < kDenseMeanGap) are flagged kBlockDenseRows: there every lane sums its run in a
//     register and touches its LDS accumulator only when its row changes (otherwise the lanes of one instruction collide on
//     the few rows there are: 55.6 vs 38.9 us on mouse_gene).
//     HISPARSE_STREAM_FORMAT=pairs|delta overrides (the parity tests run both on every case).
//
// Markers, lane padding and partition headers of the CPSR image are gone in both formats.
#ifndef HISPARSE_STREAM_TILES_H_
#define HISPARSE_STREAM_TILES_H_

#include <cstdint>
#include <string>
#include <memory>
#include <utility>
#include <vector>

#include "hisparse/common.h"

namespace hisparse {
namespace dev {

constexpr uint32_t kWaveLanes = 64;
constexpr uint32_t kWavesPerWorkgroup = 16;                   // 1024 threads: one workgroup per CU
constexpr uint32_t kConsumerWaves = 14;                       // stream elements, gather x, accumulate rows
constexpr uint32_t kLoaderWaves = kWavesPerWorkgroup - kConsumerWaves;  // refill the idle x buffer
constexpr uint32_t kSubTileCols = 8192;                       // 32 KiB of x per LDS buffer ...
constexpr uint32_t kMaxXBuffers = 4;                          // ... in a ring of up to four: refills run up to three sub-tiles ahead
constexpr uint32_t kMinXBuffers = 2;
// Rows per block: + 1 spare slot, 8-byte accumulators in both numeric modes (fixed point: 64-bit integer sums; float:
// DOUBLE sums of the float products, because LDS float atomics are slow on this part: measured 0.33 lanes/clk/CU for
// ds_add_f32 against 3.0 for ds_add_f64 and 5.6 for ds_add_u64, tools/lds_atomic_bench.hip).
//   one column slice:      32 KiB of accumulators, ring of 4;   several column slices: 96 KiB of accumulators, ring of 2
constexpr uint32_t kAccumulatorBytes = 8;
constexpr uint32_t max_block_rows(bool sliced) { return (sliced ? 96u : 32u) * 1024u / kAccumulatorBytes - 1u; }
constexpr uint32_t kMaxColSlices = 8;
constexpr uint32_t kMaxForcedColSlices = 16;   // HISPARSE_COL_SLICES / col_slices may ask the row-block planner for up to this many (the combine pass is instantiated for 2 .. 16)
constexpr uint32_t kMaxSweepSlices = 16;      // SWEEP images (round 5): a short, wide matrix -- one rank's slab -- wants few row ranges (every range sweeps all of x) and many slices
// Column-sliced plans whose image stays below this carry the combine pass of a step into the next step's kernel (hs_api.cpp: one launch per
// step in a run of hs_run calls); the planner prices the combine pass of such a plan at ~1 us instead of a launch of its own (3.5 us).
// Round 5 set 48 MiB from two points (a 29 MB slab: 10.1 -> 8.6 us; ogbl-ppa, 280 MB: a wash; pokec's SWEEP image, 247 MB: 73.5 -> 76.0).  Round 6 measured the
// middle (profiles/r06_carry_mid_size.txt, alternating runs): gplus (87 MB) 19.9 -> 19.3 us, one rank's slab of mouse_gene split 2 ways (89 MB) 20.2 -> 19.4,
// of hollywood split 8 ways (SWEEP, 113 MB) 24.6 -> 24.1, of ogbl-ppa split 2 ways (155 MB) 32.2 -> 31.4: carried up to 160 MiB.
constexpr uint64_t kCarryMaxImageBytes = 160ull << 20;
constexpr uint64_t kSlicedDeltaMaxImageBytes = 48ull << 20;   // the sliced DELTA plan of fixed-point dense layers (stream_tiles.cpp): measured up to 8.5 M non-zeros, no further
constexpr uint64_t kResidentMaxImageBytes = 256ull << 20;   // SWEEP images up to the size of the Infinity Cache are streamed without the non-temporal hint (hs_api.cpp: stream_resident)
// Row-block (PAIRS / DELTA) images, round 6 (profiles/r06_rowblock_stream_policy*.txt): without `nt` where the image fits the Infinity Cache AND its blocks walk
// several units.  Measured to gain a little even above the cache (ogbl-ppa, 267 MiB: -1.5 % warm) -- but an image that does not fit is evicted between
// steps anyway, and once a caller alternates between matrices the cacheable policy COSTS: the headline matrix round-robin over three images ran 65.4 us per SpMV
// with `sc1` streams against 56 us with `nt` (bench.py's MALL-cold leg), hollywood (872 MB) 155 against 137 us.  So: the cache's own size, no more.
// One-unit-per-block plans (a pure stream per block, no drained pipeline to refill) keep `nt` unless the whole image is a few tens of MB, where the
// step is launch- and latency-bound (the sliced DELTA plans of the pruned-NN layers: -2.5 %; mouse_gene's 50 / 100 MB slabs: +4 %).
constexpr uint64_t kRowblockResidentMaxImageBytes = 256ull << 20;
constexpr uint64_t kRowblockResidentSmallImageBytes = 32ull << 20;
constexpr uint32_t kMaxLdsBytes = 160 * 1024;
// PAIRS format
constexpr uint32_t kChunkBytes = kWaveLanes * 8;              // one wavefront step: 64 elements
constexpr uint32_t kWaveStrideBytes = kChunkBytes * kConsumerWaves;   // chunks of the 14 wavefronts are interleaved in memory
// DELTA format
// One DELTA record = TWO consecutive slots of every lane: 64 x {u32 value A, u32 value B} (one dwordx2 per lane), then 64 x {u16 gap A,
// u16 gap B} (one dword per lane): two loads per two slots -- the 384-byte single-slot records of round 1 took four (dword + ushort
// per slot), and the consumer ring is bound by requests, not bytes (profiles/r02_record_stream_bench.txt: 6.76 against 6.39 TB/s in
// isolation).  A (unit, wavefront) run = its head (slot A of the first record: absolute start positions) + run_len slots, padded to
// an even slot count with one dead slot (value 0; gap 0 in fixed point, a bridge in the float modes).
constexpr uint32_t kRecordBytes = kWaveLanes * 2 * (4 + 2);
constexpr uint32_t kMaxGap = 0xfffeu;                         // largest position gap an element slot can carry
constexpr uint32_t kBridgeGap = 0xffffu;                      // gap code of a slot without element ...
constexpr uint32_t kBridgeAdvance = 0xffffu;                  // ... which advances the position by this much
constexpr double kDeltaMinMeanGap = 8.0;                      // (denser matrices are BITMAP candidates)
constexpr double kDeltaMaxMeanGap = 20000.0;                  // sparser matrices: > 4 % of the gaps need bridges, PAIRS wins
constexpr uint64_t kDeltaMinSavedBytes = 23u << 20;           // DELTA must save this much stream against PAIRS (3.5 us at 6.5 TB/s) ...
constexpr uint64_t kDeltaMinSavedBytesFloat = 40u << 20;      // ... 6 us in the float modes (stream_tiles.cpp: the choice after the sort)
constexpr double kDenseMeanGap = 320.0;                       // DELTA blocks denser than this (>= 24 elements per row and sub-tile) sum per lane in registers (kBlockDenseRows);
                                                              // sparser ones lose with it (400000 x 100000, gap 512: 89.8 vs 83.2 us), denser ones win big (40000^2, gap 64: 34.5 vs 53.0)
enum StreamFormat : uint32_t { kFormatPairs = 0, kFormatDelta = 1, kFormatBitmap = 2, kFormatOwner = 3, kFormatPairs24 = 4, kFormatOwner24 = 5, kFormatSweep = 6 };
// PAIRS24: PAIRS with a 24-bit position word -- 7 instead of 8 bytes per element.  A wavefront step is 448 bytes: 64 value dwords, then
// 64 x 3 bytes (local_row << 13 | local_col, little endian), which the kernel reads as unaligned dwords at byte 256 + 3 * lane.
// 11 bits of row: whenever no block has more than 2046 rows (nrows itself -- the spare accumulator -- must fit).  Opt-in
// (HISPARSE_AUX_BITS=24): two loads per step, one of them unaligned, measured slower than the 8-byte form.
constexpr uint32_t kChunkBytes24 = kWaveLanes * 7;             // 448
constexpr uint32_t kWaveStrideBytes24 = kChunkBytes24 * kConsumerWaves;
constexpr uint32_t kAux24MaxRows = 2046;
// OWNER24 (round 3): OWNER with 24-bit position words in RECORDS of four steps -- 7 bytes per slot with two ALIGNED loads per four
// steps (round 2's unaligned 448-byte steps took two loads per step and lost).  Record r of a wavefront's stream holds its steps
// 4r .. 4r+3 (steps are numbered through the block, across units):
//     bytes    0 .. 1023   64 lanes x { value word of step 4r, 4r+1, 4r+2, 4r+3 }          one global_load_dwordx4 per lane
//     bytes 1024 .. 1791   64 lanes x 96 bits = the four 24-bit position words, little end first    one global_load_dwordx3 per lane
// position word = (row - row_base) << 13 | local_col: the row RELATIVE to the first row of the (unit, wavefront) share in 11 bits,
// 2047 = the wavefront's spare accumulator (padding).  Shares are still cut per unit (balanced_owner_shares), now also so that no
// share spans more than kOwnerShareRows rows; the share's row_base travels in the high half of Unit::end_step[w] (the step counts of
// a wavefront stay below 2^16 or the builder keeps the 8-byte OWNER form).  The tail of a wavefront's stream is padded to a whole
// record with steps nobody consumes.  ogbn-products: 7.06 instead of 8.07 bytes per non-zero.
constexpr uint32_t kOwnerRecordSteps = 4;
constexpr uint32_t kOwnerRecordValueBytes = kWaveLanes * 4 * kOwnerRecordSteps;            // 1024
constexpr uint32_t kOwnerRecordBytes = kOwnerRecordValueBytes + kWaveLanes * 3 * kOwnerRecordSteps;   // 1792
constexpr uint32_t kOwnerSpareField = 2047;
constexpr uint32_t kOwnerShareRows = 2047;                     // relative rows 0 .. 2046
constexpr uint32_t kOwnerStepMask = 0xffffu;                   // Unit::end_step[w] = steps | row_base << 16 in an OWNER24 image
// OWNER format (float modes, hyper-sparse matrices: ogbn-products, 2.4 M columns, 50 non-zeros per row): the cost there is not the
// element stream but x -- every row block pulls the WHOLE vector through its CU, sub-tile by sub-tile, so the staged x volume is
// (rows / rows per block) x 4 cols bytes (3.1 GB per SpMV with 8191-row blocks against 1 GB of matrix).  Rows per block are
// bounded by the LDS accumulators, and LDS float atomics force those to be doubles (ds_add_f32 runs at 0.33 lanes/clk,
// tools/lds_accum_bench.hip).  OWNER gets 4-byte accumulators WITHOUT atomics: inside a unit every consumer wavefront owns a
// contiguous stretch of the rows -- the unit's elements, sorted by (row, column), are cut on row boundaries into 14 shares of
// equal work (tiles_common.h: balanced_owner_shares; the unit barrier orders the accumulator writes, so rows may change hands
// from unit to unit; the 24-bit form keeps one ownership per block) -- and a share is dealt to the lanes in consecutive runs, so that
//   * no other wavefront touches a row's accumulator while the unit lasts (LDS executes one wavefront's instructions in order:
//     ds_read / v_add_f32 / ds_write needs no atomic),
//   * the lanes of one instruction hold non-decreasing rows, a lane sums its run in a register while the row stays the same,
//     and two lanes can only meet on a row at the end of a unit, where one segmented wavefront reduction sorts it out.
// Element = { u32 value word, u32 (local_row << 13 | local_col) }, 512 bytes per wavefront step, every wavefront's steps
// contiguous; padding slots aim value 0 at the wavefront's own spare accumulator ys[nrows + wave].
// Measured (tools/lds_accum_bench.hip): gather + read-modify-write 3.6 lanes/clk/CU against 2.3 for gather + ds_add_f64, and
// twice the rows per block.
constexpr uint32_t kOwnerAccumulatorBytes = 4;
constexpr uint32_t kOwnerColBits = 13;                        // local_col < kSubTileCols = 2^13; local_row in the 19 bits above
constexpr uint32_t owner_max_block_rows(uint32_t ring) { return (kMaxLdsBytes - ring * kSubTileCols * 4u) / kOwnerAccumulatorBytes - kConsumerWaves - 1u; }
constexpr double kOwnerMinMeanGap = 20000.0;                  // == kDeltaMaxMeanGap: sparser than anything DELTA takes
// BITMAP format (dense-row matrices, e.g. the pruned-NN layers of sw/bm.sh:21-27: 512 rows x 33 K columns, half of them set):
// a row is cut into GROUPS of 64 consecutive columns; per (row, group) the image holds one 64-bit occupancy mask and the
// values of the set columns, compacted, in column order.  4 bytes + 1 bit per column position instead of 8 bytes per non-zero
// (4.25 B per non-zero at 50 % density), and x is read LINEARLY (x[64 g + lane], coalesced, straight from L2): no x sub-tiles,
// no LDS gather, no per-sub-tile barriers.  A workgroup owns whole rows (or, for matrices with fewer rows than CUs, a column
// slice of them); each of its 16 wavefronts streams a contiguous run of groups, sums per lane in registers and adds one
// wavefront-wide sum per row to the row's LDS accumulator.  kernel: spmv_bitmap.hip; builder: bitmap_tiles.cpp.
constexpr double kBitmapMinDensity = 0.125;                   // below: < 8 of 64 lanes busy per step, PAIRS / DELTA win
constexpr uint32_t kBitmapMinCols = 2048;                     // shorter rows: a wavefront's run per row is too short to pipeline
constexpr uint32_t kBitmapGroupCols = 64;                     // one wavefront step
#ifndef HS_BITMAP_WAVES
#define HS_BITMAP_WAVES 16                                    // (12 and 8 were measured slower, round 4: transformer-50 10.6 / 11.3 / 12.0 us, profiles/r04_bitmap_waves.txt)
#endif
constexpr uint32_t kBitmapWaves = HS_BITMAP_WAVES;            // all wavefronts of the workgroup stream (no loader wavefronts)
constexpr uint32_t kBitmapMaxBlockRows = 8191;                // 64 KiB of 8-byte row accumulators
constexpr uint32_t kBitmapMaxXLdsGroups = 576;                // a block's stretch of x is kept in LDS when it has at most this many groups (144 KiB) and its accumulators fit beside it
constexpr uint32_t kBitmapSkew[4] = {170, 140, 65, 25};       // share of a wavefront by its place on its SIMD (wavefronts 0-3, 4-7, 8-11, 12-15): bitmap_tiles.cpp
constexpr uint32_t kBitmapMaskBatch = 32;                     // masks fetched per vector load (one dword per lane)
constexpr uint32_t kBitmapRunSlots = 5;                       // Unit-sized (64-byte) slots per wavefront run: the WaveSeg + a copy of its first 32 masks
// LIGHT plan (round 4): matrices of at most kLightMaxNnz non-zeros over at most kLightMaxUnits x sub-tiles are launch-bound in the row-block
// kernel (1024-thread workgroups, x staged through LDS, a combine launch for sliced plans).  They get a PAIRS image cut into up to
// kLightWorkgroupsPerCu x CUs row ranges of one column slice, every block in the strided layout (lane l of chunk c = sorted element l x chunks + c), and
// the kernel spmv_light_kernel (spmv_kernels.hip): 256-thread workgroups, x gathered straight from L2, y written by the one launch.
constexpr uint64_t kLightMaxNnz = 2u << 20;                   // 17 MB of PAIRS stream.  Above that the gathers decide: one rank's slab of mouse_gene split 8 ways
                                                              // (3.6 M non-zeros, random columns: a 64-lane gather is 64 L1 look-ups, ~6 us per CU) runs 17-18 us in this
                                                              // kernel whatever the accumulation scheme against 10 us for the sliced row-block plan + combine pass (x in LDS)
constexpr uint32_t kLightMaxUnits = 16;                       // sub-tiles of a block (one slice): lanes 0 .. 15 hold the unit ends
constexpr uint32_t kLightWorkgroupsPerCu = 4;                 // measured flat between 2 and 6 per CU (the kernel is compiled for 6: 85 registers); HISPARSE_LIGHT_WGS=1..6 for experiments
constexpr uint32_t kLightMaxBlockRows = 3071;                 // 24 KiB of accumulators per workgroup: up to six of them per CU
constexpr uint32_t kLightMinBlockNnz = 1024;                  // no block smaller than 16 chunks (unless the matrix is)
// SWEEP format (round 4; hyper-sparse matrices, x NOT staged in LDS): a block = (row range x CONTIGUOUS column slice), its elements sorted by
// (column, row) and stored in that order as 512-byte chunks of 64 x { u32 value word, u32 (local_row << 16 | column - chunk base) }; chunk k
// of a block belongs to step k / 8 of wavefront k % 8 (kSweepWaves), so the wavefronts of the workgroup sweep the slice's columns together, once,
// left to right.  The chunk bases (absolute column of the chunk's first element) sit in a table per block, [wavefront][step].  x[column] is
// a per-lane global load (x is L2 / Infinity-Cache resident; column order makes the 64 lanes of one gather touch a handful of 128-byte lines),
// products go to LDS accumulators with atomics (doubles; fixed point: 32-bit sums + a carry bit): no units, no x refills, no barriers between a block's prologue and epilogue.
// Padding slots (the tail of a block's last step, and a chunk cut short because the next column lies more than 65535 beyond its base): value 0,
// local_row = nrows (the spare accumulator), offset 0.  kernel: spmv_sweep.hip; builder: sweep_tiles.cpp; tools/gather_bench.hip is the
// block-level measurement it was designed from.
#ifndef HS_SWEEP_WAVES
#define HS_SWEEP_WAVES 8                                      // (4 and 16 were measured too: -DHS_SWEEP_WAVES=..., tools/history/r04/sweep_waves.sh, profiles/r04_sweep_waves.txt)
#endif
constexpr uint32_t kSweepWaves = HS_SWEEP_WAVES;              // all wavefronts stream (no loaders)
// the LDS holds nothing but accumulators: doubles in the float modes; fixed point: a wrapping 32-bit sum + a carry bit per row (spmv_sweep.hip)
constexpr uint32_t kSweepMaxBlockRowsFloat = kMaxLdsBytes / kAccumulatorBytes - 1;                // 20479
constexpr uint32_t kSweepMaxBlockRowsFixed = (kMaxLdsBytes / 4 - 2) * 32 / 33 - 1;                // 39716: (rows + 1) x 4 bytes + (rows + 32) / 32 x 4 bytes <= 160 KiB
constexpr uint32_t kSweepColAlign = 32;                       // slices start on a 128-byte line of x
// Chosen (unforced) where OWNER24 would be (mean position gap rows x cols / nnz above kOwnerMinMeanGap, more than kSweepMinNnz non-zeros) and its
// plan is modelled faster (stream_tiles.cpp: "SWEEP"): OWNER24 pays per (row range x sub-tile) unit whatever the unit holds, SWEEP per element
// and per line of x.  For square power-law matrices that comes out as a mean gap of ~60 K in fixed point, ~70 K in the float modes.  Measured on power-law squares of 1.0 / 1.6 / 2.4 M rows (tools/probe_sweep.py,
// profiles/r04_sweep_vs_owner_synthetic.txt), whole step, SWEEP against OWNER24: gap 50 K +3 ... -3 % (fixed) / +5 ... +11 % (float), 70 K -6 ...
// -13 % / +2 ... -7 %, 100 K -12 ... -24 % / -3 ... -20 %, 200 K -20 ... -38 % in both; pokec (gap 87 K) 95.5 -> 78.0 us fixed, 122.6 -> 88.3 us
// float_pob; ogbn-products (48 K) stays OWNER24 (204 against 216 us).
constexpr double kSweepSlabMinMeanGap = 8000.0;                // short, wide fixed-point slabs that fit the Infinity Cache take SWEEP from this mean gap on (stream_tiles.cpp)
constexpr uint64_t kSweepMinNnz = (2u << 20) + 1;             // smaller matrices: the LIGHT plan's (when x is short) or the row-block kernel's
constexpr uint64_t kFloatOneSliceOwnerMinNnz = 8u << 20;         // float modes: a one-slice PAIRS-family plan of at least this many non-zeros is planned again as OWNER24 (stream_tiles.cpp, round 6)
constexpr uint64_t kSweepMinNnzWide = 256u << 10;             // ... unless x is wider than the LIGHT plan takes (kLightMaxUnits sub-tiles): SWEEP from here on (round 6)
constexpr uint32_t kDenseBlockRows = 32;                      // blocks with at most this many rows use the dense-row layout
constexpr uint32_t kBlockDenseRows = 1u;                      // Block::flags bit
constexpr uint32_t kBlockLastOfPartition = 2u;                // Block::flags bit: the workgroup's next block (if any) begins in a LATER row partition than this one ends in
constexpr uint32_t kNoBlock = 0xffffffffu;

// Mirrored in the kernel source (read through scalar loads).
struct Block {
    uint32_t row0;          // first row (absolute, padded numbering)
    uint32_t nrows;         // <= max_block_rows(); local row nrows is the spare accumulator padding elements hit (OWNER: nrows + wave)
    uint32_t row_part;      // row partition of the block's first row (hs_run_partition runs the blocks with row_part <= filter <= last_part)
    uint32_t unit_begin;    // units [unit_begin, unit_end), consumed in this order
    uint32_t unit_end;
    uint32_t flags;         // kBlockDenseRows: long rows.  PAIRS: chunks are dealt linearly and mostly hold ONE row (wavefront-wide
                            // register sums); DELTA: every lane sums its own run in a register until its row changes
    uint32_t out_offset;    // word offset of the block's first row in the output: y (one slice) or the per-slice partials
    uint32_t next;          // index of the next block of the same workgroup, 0 = none (workgroup g starts at blocks[g])
    uint64_t wave_offset[kConsumerWaves];   // byte offset of each consumer wavefront's first chunk (PAIRS: its next is kWaveStrideBytes on)
                                            // or of its contiguous record stream (DELTA)
    // copies of what the kernel would otherwise fetch through two more dependent loads before its first stream load
    uint32_t total_steps[kConsumerWaves];   // == units[unit_end - 1].end_step
    uint32_t first_end[kConsumerWaves];     // == units[unit_begin].end_step
    uint32_t first_col0, first_ncols;       // == units[unit_begin].col0 / .ncols (0 / 0 for a block without units)
    uint32_t last_part;     // row partition of the block's LAST row (>= row_part: since round 5 a row range may cross partition borders)
    uint32_t next_part;     // row_part of the workgroup's next block, 0xffffffff: none -- a run of partition p (hs_run_partition) goes on while next_part <= p
    uint32_t pad[12];
};
struct Unit {
    uint32_t col0;          // first absolute column of the x sub-tile
    uint32_t ncols;         // multiple of 8, <= kSubTileCols
    uint32_t end_step[kConsumerWaves];      // per wavefront: its stream position (in chunks / records, heads included) after this unit
};
// SWEEP images: a Block describes (row range x column slice) -- row0, nrows, row_part, flags, out_offset, next as above, wave_offset[0] = byte
// offset of the block's first chunk, wave_offset[1] = byte offset of its chunk-base table (u32 [kSweepWaves][steps]), total_steps[0] = steps, first_col0 /
// first_ncols = the slice's first column / column count; no units.
// BITMAP images re-use the two tables: a Block describes (row range x column slice) -- row0, nrows, row_part, flags, out_offset, next as
// above, first_col0 = first column of the slice, first_ncols = groups per row in the slice -- and its units [unit_begin, unit_end)
// are kBitmapWaves RUN HEADERS of kBitmapRunSlots x 64 bytes, one per wavefront: a WaveSeg (same 64 bytes as a Unit) followed by a copy
// of the run's first 32 masks (256 bytes, zero padded) -- the kernel fetches descriptor and first masks in ONE round trip:
struct WaveSeg {
    uint32_t row_begin, row_end;   // local rows [row_begin, row_end) of the block; row_end - row_begin == 1: the groups [g_begin, g_end)
    uint32_t g_begin, g_end;       //   of that row (relative to the slice), otherwise WHOLE rows (g_begin = 0, g_end = groups per row)
    uint32_t value_lo, value_hi;   // word offset (from the image start) of the first compacted value of the wavefront's run
    uint32_t mask_lo, mask_hi;     // 8-byte offset (from the image start) of the first mask of the run; masks and values of a
    uint32_t pad[8];               //   block are stored in (row, group) order, so a run is contiguous in both
};
static_assert(sizeof(WaveSeg) == 64, "WaveSeg overlays Unit");
static_assert(sizeof(Block) == 320, "Block layout is shared with the device code");
static_assert(sizeof(Unit) == 8 + 4 * kConsumerWaves, "Unit layout is shared with the device code");

class GpuTiler;   // gpu_tiles.h

// resize() leaves new bytes uninitialised: the builders zero-fill a few hundred MB from many threads (detail::resize_zeroed), which
// also spreads the first-touch page faults a single-threaded assign() would take one after the other
template <typename T>
struct DefaultInitAllocator : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInitAllocator<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U* p) { ::new (static_cast<void*>(p)) U; }
    template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
using ImageBytes = std::vector<uint8_t, DefaultInitAllocator<uint8_t>>;

// The second image of a float BITMAP matrix, for the SpMM on the matrix engine (spmm_mfma.hip): rows in TILES of 16, every
// (tile, 64-column group) holds the 16 rows' occupancy masks side by side; the values are compacted in the order the kernel's lanes take
// them -- tile, group, MFMA step s (columns 4 s .. 4 s + 3), lane 16 k + i (row i, column 4 s + k) -- so that the values of one step are
// one coalesced load at (running base + set bits below the lane in the step's ballot); the value index at the start of every wavefront's
// chunk of groups sits in a table.  One allocation of 32-bit words:
//   [masks: tiles x groups x 16 x u64][unit bases: tiles x chunks x u32][values]
struct MfmaImage {
    ImageBytes words;                    // as laid out above (empty: no such image)
    uint32_t tiles = 0;                  // ceil(num_rows / 16)
    uint32_t groups = 0;                 // 64-column groups per row
    uint32_t chunk = 0;                  // groups per wavefront unit
    uint32_t chunks = 0;                 // ceil(groups / chunk), rounded up to a multiple of 4 (a workgroup of four units stays inside one row tile)
    uint64_t offsets_word = 0, values_word = 0;   // word offsets of the two tables behind the masks
    uint64_t words_bytes = 0;            // size of the image (= words.size() when it was built on the host)
    uint8_t* d_words = nullptr;          // built on the device (gpu_tiles.hip): the new owner frees it; `words` is empty then
    bool present() const { return words_bytes != 0 && (d_words != nullptr || !words.empty()); }
};
constexpr uint32_t kMfmaTileRows = 16;

struct StreamTiles {
    ImageBytes image;                    // element streams, uploaded verbatim (host builder)
    MfmaImage mfma;                      // float BITMAP images only (bitmap_tiles.cpp)
    uint32_t bitmap_x_groups = 0;        // BITMAP: groups of x a block reads (its column slice), when the kernel is to keep that stretch in LDS; else 0
    uint8_t* d_image = nullptr;          // GPU builder: the image, already in device memory (image_bytes + slack); the caller owns it
    uint64_t image_bytes = 0;
    std::vector<Block> blocks;
    std::vector<Unit> units;
    std::vector<uint32_t> wg_first;      // workgroup g owns block_order[wg_first[g] .. wg_first[g+1]), in this order: a host-side
    std::vector<uint32_t> block_order;   // description (tests, statistics); the kernel follows blocks[g] -> Block::next
    std::vector<uint32_t> part_heads;    // [row partition][workgroup]: first block of that workgroup in that partition or kNoBlock
                                         // (hs_run_partition starts there and stops at kBlockLastOfPartition)
    uint32_t num_workgroups = 0;
    uint32_t max_block_rows = 0;
    uint64_t sweep_table_bytes = 0;      // SWEEP: bytes of the chunk-base tables at the end of the image (statistics)
    uint32_t spmm_vectors = 1;           // 4: a SWEEP image planned for spmm_sweep.hip (rows per block / 4; HISPARSE_SPMM_VECTORS=4)
    bool light = false;                  // the LIGHT plan (below): PAIRS image, one slice, up to kLightWorkgroupsPerCu x CUs small blocks, spmv_light_kernel
    uint32_t col_slices = 1;             // > 1: blocks write per-slice partial results, a combine pass adds them
    uint32_t ring_buffers = kMaxXBuffers;
    StreamFormat format = kFormatPairs;
    uint64_t nnz = 0;
    uint64_t elements = 0;               // element slots including bridges and chunk padding (DELTA head records not counted)
};

// hs_load_matrix_csr: the matrix BEFORE csr2cpsr -- CSRMatrix<float> arrays (sw/data_loader.h:19-31), dimensions not yet rounded up.
// The builder pads like util_round_csr_matrix_dim (empty rows / counted columns) and converts the values like
// csr_matrix_convert_from_float, so the image is byte for byte the one the CPSR path gives for the same matrix.
struct CsrView {
    uint32_t num_rows = 0, num_cols = 0;     // as loaded (<= the padded dimensions handed to build_stream_tiles)
    const uint32_t* indptr = nullptr;        // num_rows + 1
    const uint32_t* indices = nullptr;
    const float* values = nullptr;
};

// Decode + validate + re-tile.  `max_workgroups` = workgroups the device keeps resident (one per CU).
// Returns false and sets `error` when the buffers are not a valid CPSR image for the geometry.
// `gpu_stream` != nullptr: the per-non-zero passes run on the device of the current HIP context (gpu_tiles.h), the image stays there
// (out.d_image); formats it does not cover (BITMAP) and matrices with duplicate entries are built on the host as before.
// image_slack: bytes the device allocation of the image must extend past its end (the kernels' clamped prefetches).
// `csr` != nullptr: the source is a CSR matrix instead of the CPSR image (channel / n_packets are ignored; needs use_gpu -- BITMAP
// images are still built by host threads, straight from the CSR rows).
bool build_stream_tiles(const void* const channel[NUM_HBM_CHANNELS], const uint64_t n_packets[NUM_HBM_CHANNELS],
                        const Geometry& geom, uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions,
                        uint32_t num_col_partitions, uint32_t max_workgroups, StreamTiles& out, std::string& error,
                        void* gpu_stream = nullptr, bool use_gpu = false, uint64_t image_slack = 0, const CsrView* csr = nullptr);

}  // namespace dev
}  // namespace hisparse

#endif  // HISPARSE_STREAM_TILES_H_
