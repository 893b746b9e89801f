/*
 * hisparse_hip.h — the drop-in boundary: C-ABI of libhisparse_hip.so (MI355X / gfx950).
 *
 * This library takes the place of the FPGA launch path of the reference.  What a reference driver
 * hands to OpenCL/XRT — 16 matrix channel buffers, the packed vector, the packed result and five
 * scalars — is what these entry points take, byte for byte:
 *
 *   reference (sw/benchmark.cpp, sw/host.cpp, spmv_csim/csim.cpp)             this library
 *   ------------------------------------------------------------------------  -----------------------
 *   cl::Context / cl::Program(xclbin) / 5 x cl::Kernel / queue  (:372-404)    hs_create
 *   16 x CL_BUFFER_RDONLY + enqueueMigrateMemObjects            (:231-252,266-269)  hs_load_matrix
 *   vector_buf + migrate                                        (:255-260,270-272)  hs_load_vector
 *   per row partition: setArg(row_part_id, part_len) x4, 5 x enqueueTask, finish() (:318-338)
 *                                                                             hs_run_partition
 *   the whole `for row_part_id` loop of one SpMV                (:318-339)    hs_run (one launch)
 *   queue.finish()                                              (:337)        hs_sync
 *   enqueueMigrateMemObjects(result, D2H)                       (sw/host.cpp:370)   hs_read_result
 *   csim: top_wrapper(m0..m15, x, y, row_part_id, part_len, num_col_partitions,
 *                     num_partitions, num_cols)                 (csim.cpp:22-46)    hs_run_partition
 *   OCL_CHECK / CHECK_ERR -> print + exit(EXIT_FAILURE)         (xcl2.hpp:40-46, benchmark.cpp:56-61)
 *                                                                             negative return codes
 *
 * Ownership: the caller owns every host pointer it passes; hs_load_* copy what they need before
 * returning.  The library owns all device memory until hs_destroy.  At load time the library builds
 * private device-side structures from the CPSR buffers (see DESIGN.md "stream tiles"); the inputs
 * themselves are never modified.
 *
 * Threading: one context per GPU; calls on one context must not overlap.  hs_run* are asynchronous
 * on the context's HIP stream; hs_sync / hs_read_result wait for them.
 *
 * There is NO CPU fallback: every entry point that needs the GPU fails with HS_ERR_NO_DEVICE or
 * HS_ERR_HIP when none is usable.  (libhisparse_cpu.so -- hisparse_amd/csrc/cpu_backend.cpp -- exports the core entry points of this
 * header on host threads for machines without a GPU; it is a separate library a driver links INSTEAD, as the reference's driver is
 * built against csim or against the xclbin, and this library never loads it.)
 */
#ifndef HISPARSE_HIP_H_
#define HISPARSE_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HS_NUM_CHANNELS 16 /* NUM_HBM_CHANNELS, spmv/libfpga/common.h:173-176 */

enum {
    HS_OK = 0,
    HS_ERR_BAD_ARG = -1,      /* null pointer, unknown impl, dimension not padded ... */
    HS_ERR_NO_DEVICE = -2,    /* no usable gfx950 device */
    HS_ERR_HIP = -3,          /* a HIP runtime call failed; see hs_last_error */
    HS_ERR_BAD_MATRIX = -4,   /* channel buffers are not a valid CPSR image for the given geometry */
    HS_ERR_NOT_LOADED = -5,   /* run before hs_load_matrix / hs_load_vector */
    HS_ERR_UNSUPPORTED = -6,  /* configuration the device cannot hold */
    HS_ERR_NO_MEMORY = -7,
};

/* numeric mode == the reference's IMPL make variable (sw/Makefile:2-12) */
enum { HS_IMPL_FIXED = 0, HS_IMPL_FLOAT_POB = 1, HS_IMPL_FLOAT_STALL = 2 };

/* device-private stream formats (hisparse_amd/csrc/stream_tiles.h) */
enum { HS_STREAM_PAIRS = 0, HS_STREAM_DELTA = 1, HS_STREAM_BITMAP = 2, HS_STREAM_OWNER = 3, HS_STREAM_PAIRS24 = 4, HS_STREAM_OWNER24 = 5, HS_STREAM_SWEEP = 6 };

typedef struct hs_context hs_context;

typedef struct {
    uint64_t nnz;               /* true non-zeros found in the CPSR image */
    uint64_t cpsr_bytes;        /* bytes of the 16 channel buffers as handed in */
    uint64_t stream_bytes;      /* bytes of the device-private element streams the kernel reads per SpMV */
    uint64_t stream_elements;   /* element slots in those streams (non-zeros + bridges + chunk padding) */
    uint32_t num_blocks;        /* row blocks (each owned by one workgroup at a time) */
    uint32_t num_units;         /* (row block, x sub-tile) units */
    uint32_t num_workgroups;    /* grid size of the SpMV kernel */
    uint32_t lds_bytes;         /* dynamic LDS per workgroup (two x sub-tile buffers + row accumulators) */
    uint32_t num_compute_units; /* of the device */
    uint32_t col_slices;        /* column slices (1 = none; > 1 adds the small combine pass) */
    uint32_t ring_buffers;      /* x sub-tile buffers in the LDS ring */
    uint32_t stream_format;     /* HS_STREAM_PAIRS (8 B per element), HS_STREAM_DELTA (6 B per slot) or HS_STREAM_BITMAP (4 B + 1 bit per column) HS_STREAM_OWNER (8 B per element, float accumulators) the 7-byte forms of PAIRS / OWNER, or HS_STREAM_SWEEP (8 B per element in column order, x gathered from L2: very sparse matrices), chosen per matrix */
    double load_seconds;        /* wall time of the last hs_load_matrix (decode + re-tile + H2D) */
    uint32_t retiled_on_gpu;    /* 1: the per-non-zero passes of the re-tiling ran on the device (gpu_tiles.h); 0: on the host */
    uint32_t light_kernel;      /* 1: the LIGHT plan -- a small matrix run by the 256-thread single-launch kernel (spmv_light_kernel) over a PAIRS image */
    uint32_t stream_resident;   /* 1: the plan found the image resident in the 256 MiB Infinity Cache across consecutive SpMVs and streams it WITHOUT the non-temporal hint (SWEEP images up to the cache size; PAIRS / DELTA images by the rule of stream_tiles.h: kRowblockResidentMaxImageBytes); option "stream_resident" = 0 | 1 decides otherwise */
    uint32_t reserved0;
} hs_stats;

const char* hs_strerror(int code);
/* message of the last failure on this context (or of the last failed hs_create when ctx is NULL) */
const char* hs_last_error(const hs_context* ctx);

/* Open device `device_id` (HIP ordinal) for one numeric mode and one bank geometry.
 * ob_bank / vb_bank are OB_BANK_SIZE / VB_BANK_SIZE in words (0 = the shipped bitstream's value:
 * 8192 (1024 for float_pob) and 4096).  They must equal the values the matrix was formatted with —
 * the `<v> <o>` arguments of sw/benchmark.cpp:364-365 times 1024. */
int hs_create(hs_context** ctx, int device_id, int impl, uint32_t ob_bank, uint32_t vb_bank);
int hs_destroy(hs_context* ctx);

/* channel[c] points to n_packets[c] 64-byte packets {u32 idx[8]; u32 val[8];} laid out as in
 * SURVEY.md Appendix A.2 (headers + interleaved payload).  num_rows / num_cols are the PADDED
 * dimensions.  Replaces any previously loaded matrix. */
int hs_load_matrix(hs_context* ctx, const void* const channel[HS_NUM_CHANNELS], const uint64_t n_packets[HS_NUM_CHANNELS],
                   uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions, uint32_t num_col_partitions);

/* packed_x: num_cols value words (PACKED_VAL_T[num_cols/8]) in natural order. */
int hs_load_vector(hs_context* ctx, const void* packed_x, uint32_t num_cols);

/* One full SpMV (every row partition) in one launch sequence; asynchronous.
 * Column-sliced plans (hs_stats.col_slices > 1: the SpMV kernel leaves per-slice partial rows, a small combine pass adds them up):
 * when hs_run follows hs_run on the library's own stream, the combine pass of the earlier
 * call is done by the LATER call's kernel as its first act (a second set of partial vectors takes the later call's own), and the
 * stand-alone combine is launched only when something else follows -- every other entry point (hs_sync, hs_read_result, hs_feedback,
 * hs_run_partition, hs_bind_device_result with another target, hs_set_stream, ...) launches it first, so the order of effects on the
 * stream is exactly that of "kernel, combine" per call.  With a caller-owned stream (hs_set_stream), or once hs_get_stream has handed the
 * stream out, every call completes in itself.  Chosen per matrix at load time (hs_api.cpp: images below 160 MiB, where the second launch is a
 * large part of the step, and OWNER images); hs_set_option "carry_combine" = 0 | 1 decides otherwise.
 * CONSEQUENCE for callers that look at y through a device pointer (hs_device_result, hs_bind_device_result) on the library's own stream:
 * y of the last hs_run is complete only after an entry point of THIS library has settled it -- hs_sync, hs_read_result, hs_push_result,
 * hs_get_stream (from then on every call completes in itself) -- not after hipDeviceSynchronize / an event alone.  Transient targets are
 * never left owed: hs_spmm_device, hs_spmspv's dense dispatch and its one-time timing settle before they return.  x and y that share
 * memory (in-place y = A*y) never take the carried path. */
int hs_run(hs_context* ctx);
/* EXTENSION: `steps` x hs_run from ONE call -- the reference's NUM_RUNS loop (sw/benchmark.cpp:315-343) as a unit.  A step of a small
 * matrix is two launches of a few microseconds each, and how fast the HOST enqueues them then decides the step time (a Python loop over
 * hs_run: one ctypes call + two launches per step).  Default: the launches are enqueued from a C loop.  hs_set_option "batch_graph" = 1:
 * the step sequence is captured once into a hipGraph (per step count, vector, result target and stream; re-captured when one of them
 * changes, dropped by hs_load_matrix) and replayed with one hipGraphLaunch.  Same kernels, same results; asynchronous like hs_run.
 * A batch is ONE unit in stream order: inside it the steps of a column-sliced plan carry each other's combine pass (see hs_run) also on a
 * caller-owned stream, and the last step's combine is enqueued before the call returns. */
int hs_run_batch(hs_context* ctx, uint32_t steps);
/* One row partition, with the reference's scalar arguments; part_len = rows per cluster
 * (checked against the geometry).  Rows of other partitions keep their previous contents. */
int hs_run_partition(hs_context* ctx, uint32_t row_part_id, uint32_t part_len);
int hs_sync(hs_context* ctx);
/* Waits, then copies num_rows value words (PACKED_VAL_T[num_rows/8]) to the host. */
int hs_read_result(hs_context* ctx, void* packed_y, uint32_t num_rows);

/* ---- zero-copy hooks for callers that already live on the GPU (PyTorch tensors, RCCL) ---------- */
/* Run on a caller-owned hipStream_t instead of the context's private stream (NULL restores it). */
int hs_set_stream(hs_context* ctx, void* hip_stream);
/* The hipStream_t the context currently launches on (its private one unless hs_set_stream changed it): lets several contexts of
 * one device share a stream, i.e. run strictly one after the other (bench.py's round-robin over several matrices). */
int hs_get_stream(hs_context* ctx, void** hip_stream);
/* Device addresses of the library's packed x (num_cols words) and packed y (num_rows words). */
int hs_device_vector(hs_context* ctx, void** x_dev);
int hs_device_result(hs_context* ctx, void** y_dev);
/* Make the kernels read x from / write y to caller-owned device memory, 16-byte aligned (NULL restores the library's). */
int hs_bind_device_vector(hs_context* ctx, const void* x_dev);
int hs_bind_device_result(hs_context* ctx, void* y_dev);
/* EXTENSION (multi-GPU): stream-ordered copy of the first num_words words of the result (the context's own y or
 * the bound one) into n_dst <= 8 other device buffers with plain stores -- on a node with peer access enabled
 * (hipDeviceEnablePeerAccess) those are the other GPUs' gather buffers, written over xGMI: a gather of the row slabs with no collective
 * on the critical path (hisparse_amd/csrc/benchmark.cpp --gpus N --peer-gather).  num_words: multiple of 4; dst: 16-byte aligned. */
int hs_push_result(hs_context* ctx, void* const* dst, uint32_t n_dst, uint32_t num_words);

/* ---- tuning options (EXTENSION) ----------------------------------------------------------------------
 * The library's configuration surface.  key: the name of a tuning switch, case-insensitive, with or without the "HISPARSE_" prefix of its
 * environment spelling -- plan-time keys (take effect at the NEXT hs_load_matrix / hs_load_matrix_csr of this context): stream_format
 * (pairs|delta|owner|owner24|sweep|bitmap), col_slices, max_rows, cross_partitions (0: row blocks end at the reference's row-partition borders), spmm_vectors (4: plan the image for the four-column SpMM kernel, see hs_spmm), row_runs, delta_deal (wave: the dealing of DELTA runs of rounds 1-4), pow2_slices (1: column-slice counts 1, 2, 4, 8 only for matrices of more than sixteen sub-tiles, the rule of rounds 1-4), aux_bits, xcd_affinity, retile (host), bitmap_skew, bitmap_x_lds,
 * bitmap_build, walk_lanes, no_mfma_image, mfma_chunk, light (0|1: the small-matrix kernel), sweep (0|1: the
 * column-ordered format of very sparse matrices), plan_debug; call-time keys: spmm_fused, spmm_mfma,
 * spmspv (sparse|auto|dense), spmspv_crossover, iterate_graph, batch_graph (hs_run_batch); carry_combine (0|1, plan-time: see hs_run), stream_resident (0|1, plan-time: SWEEP and PAIRS / DELTA images streamed without the non-temporal hint; default: SWEEP images up to 256 MiB, the Infinity Cache; PAIRS / DELTA images up to 256 MiB whose blocks walk several units, or up to 32 MiB: hs_stats.stream_resident says what the plan took).  autotune (0|1, plan-time, round 6: the load builds the planner's own image AND every other element format the matrix can take, times a few SpMVs of each on a zero vector and keeps the fastest -- a handful of extra loads of tens of milliseconds each, for callers that run one matrix thousands of times; a forced stream_format switches it off; hs_get_stats says what was kept), plan_census (0: plan from the rows' non-zero counts alone, as rounds 1-5 did).  value NULL or "" clears the option.  An unknown key is HS_ERR_BAD_ARG.
 * Options set here win over the environment variable of the same name, which stays as the fallback for tools and tests.  None of them
 * changes WHAT is computed.  The switches that do (HISPARSE_ABLATE, HISPARSE_DEPTH: profiling builds with parts of the work removed) are
 * not options: they exist only in libhisparse_hip_prof.so, and this library refuses to run (HS_ERR_BAD_ARG from hs_run, hs_run_partition,
 * hs_iterate, hs_time_runs) while one of them is set in the environment. */
int hs_set_option(hs_context* ctx, const char* key, const char* value);

/* ---- iterative callers (EXTENSION: no reference counterpart) --------------
 * The reference's drivers run one SpMV and read y back; an iterative caller (PageRank: sw/data_formatter.h:33-47
 * normalises the matrix for it) would feed y back into x over PCIe.  On the GPU the feedback stays in HBM:
 *   hs_feedback: x[i] = scale (*) y[i] (+) shift for i < min(num_rows, num_cols), in the context's arithmetic
 *     (Q8.24 multiply with AP_RND/AP_SAT then saturating add; or one fp32 multiply, then one fp32 add);
 *     the remaining words of x keep their value.  scale / shift are value words (hsf_pack_vector of one float).
 *     Writes the context's vector -- also when it was bound with hs_bind_device_vector.
 *   hs_iterate: `iterations` x { hs_run; hs_feedback } enqueued from one call (HISPARSE_ITERATE_GRAPH=1: replayed from
 *     captured hipGraphs of 32 iterations -- measured no faster on ROCm 7.2, see hs_api.cpp).  Asynchronous; y holds the
 *     last SpMV's result. */
int hs_feedback(hs_context* ctx, uint32_t scale_word, uint32_t shift_word);
int hs_iterate(hs_context* ctx, uint32_t iterations, uint32_t scale_word, uint32_t shift_word);

/* ---- SpMSpV (EXTENSION: the reference stubs the types -- SPMSPV_MAT_PKT_T, IDX_VAL_T, spmv/libfpga/common.h:52-54 -- and the CSC
 * conversion csr2csc, sw/data_loader.h:109-144 -- but has no kernel) --------------------------------------
 * y = A x for a SPARSE x: only the columns named by x's entries are read.
 *   hs_load_matrix_csc: CSCMatrix arrays (indptr[num_cols + 1], row index and value word per non-zero, value words in the context's
 *     numeric mode: hsf_csr_to_csc), independent of the matrix hs_load_matrix holds (see DENSE DISPATCH below); validated, copied to the device.
 *   hs_spmspv: x as `count` IDX_VAL_T pairs in HOST memory.  Asynchronous: the pairs are checked and copied into a pinned, device-mapped
 *     staging buffer of the context (two halves used in turn; the call returns once they are in it) and two kernels follow on the context's
 *     stream -- EXPAND (a workgroup per 64 entries computes their columns' products and places each in the BIN of the row block it falls
 *     in: one atomic per workgroup and row block claims the room) and ACCUMULATE (one workgroup per row block of 8192 rows adds ITS bin's
 *     products in LDS and writes its rows) -- no scan, no sort, no copy command, no host synchronisation.  The result is a dense packed y of
 *     num_rows words: hs_read_spmspv_result.  An entry may name a column more than once (the products simply add up): a bin holds as many
 *     products as the matrix has non-zeros in its row block, so such a call is cut into passes of unique columns.
 *     CROSSOVER: the sparse path costs ~10-14 us + products / 45 G/s (measured, profiles/r04_spmspv_binned.txt: ogbl-ppa 0.05 % / 0.1 % / 1 % /
 *     5 % of the columns 14 / 14 / 24 / 61 us with host entries -- 10 / 10 / 20 / 56 with device entries --, against 62-68 us for the dense SpMV,
 *     which wins from ~6 %).
 *     DENSE DISPATCH (opt-in).  The CSC matrix is independent of the matrix hs_load_matrix / hs_load_matrix_csr holds: a context may keep A for
 *     SpMV and A^T, or any other matrix of the same shape, as CSC, and by default hs_spmspv ALWAYS multiplies by the CSC matrix (the sparse
 *     path, whatever the size of x).  A caller who has loaded the SAME matrix both ways may say so -- hs_set_option(ctx, "spmspv", "auto") --
 *     and hs_spmspv then answers calls beyond the crossover with the dense SpMV of the loaded matrix instead (x scattered into a zero vector,
 *     one hs_run), provided the shapes match after padding and x names no column twice.  The rule: the host knows the call's product count
 *     exactly, and the dense SpMV of the loaded matrix is timed once (the first call that could use it: three launches and one
 *     synchronisation).  `spmspv_crossover` (a fraction of the columns; 0 = never) replaces the rule and is the same declaration;
 *     `spmspv` = sparse | dense forces a path.  That the two ARE the same matrix is the caller's contract under every one of these options.
 *   hs_spmspv_device: the same with the pairs already in DEVICE memory (8-byte aligned): nothing but the two launches (3-4 us less).  No
 *     index check (out-of-range columns are ignored), no dense dispatch, and no column may be named twice (a bin that overflows contributes
 *     nothing -- its rows come out zero -- and hs_read_spmspv_result reports HS_ERR_BAD_ARG; hs_spmspv_status tells without a read-back).
 * Arithmetic as in hs_run: fixed = saturating sum of individually rounded / saturated products (bit-exact, order free);
 * float = fp32 products summed in double per row block, rounded once (tolerance). */
typedef struct { uint32_t index; uint32_t val; } hs_idx_val;    /* IDX_VAL_T, spmv/libfpga/common.h:54 */
int hs_load_matrix_csc(hs_context* ctx, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, uint32_t num_rows,
                       uint32_t num_cols);
int hs_spmspv(hs_context* ctx, const hs_idx_val* x_entries, uint32_t count);
int hs_spmspv_device(hs_context* ctx, const hs_idx_val* x_entries_dev, uint32_t count);
int hs_read_spmspv_result(hs_context* ctx, void* packed_y, uint32_t num_rows);
/* For callers that consume the SpMSpV result ON THE DEVICE (hs_device_... of d_csc_y is not exported: they bind their own y) and never call
 * hs_read_spmspv_result: *overflowed (may be NULL; waits for the stream) = 1 when a hs_spmspv_device call since the last
 * hs_read_spmspv_result asked a bin for more products than it holds (a column named twice) -- the rows of such a row block are then ZERO,
 * never an earlier call's products; *overflow_word_dev (may be NULL) = the device address of that word, for a kernel of the caller's to test. */
int hs_spmspv_status(hs_context* ctx, uint32_t* overflowed, void** overflow_word_dev);

/* ---- load straight from CSR (EXTENSION, SURVEY.md section 8(f)-1: the pre-processing on the GPU) ------------------------------------
 * The reference's driver turns CSRMatrix<float> (sw/data_loader.h:19-31) into CPSR on the host (csr2cpsr + packet assembly,
 * sw/data_formatter.h:468-544, sw/benchmark.cpp:105-195: 0.2 - 10.6 s single-threaded, paper Table 8) only for the device to decode it
 * again.  hs_load_matrix_csr skips that round trip: the three CSR arrays are copied to the device, padded like
 * util_round_csr_matrix_dim (rows to a multiple of 128 * interleave, columns to a multiple of 8), their values converted like
 * csr_matrix_convert_from_float (sw/data_loader.h:76-84), and the device image is built by the same planner and kernels as in
 * hs_load_matrix -- it is byte for byte the image hs_load_matrix builds from csr2cpsr's output of the same matrix
 * (tests/test_gpu_retile.py), so every parity statement carries over.  Column indices inside a row may be in any order.  A matrix that
 * holds a (row, column) twice (legal for the reference's formatter: both products are added) is formatted on the host instead --
 * csr2cpsr + packet assembly + the host builder, i.e. what hs_load_matrix does with such a matrix.  padded_rows / padded_cols (may be
 * NULL) receive the dimensions hs_load_vector / hs_read_result then expect. */
int hs_load_matrix_csr(hs_context* ctx, uint32_t num_rows, uint32_t num_cols, const uint32_t* indptr, const uint32_t* indices, const float* values,
                       uint32_t* padded_rows, uint32_t* padded_cols);

/* ---- SpMM (EXTENSION; the reference has no SpMM) ----------------------------------------------------------
 * Y = A X for k dense vectors, column j of X / Y being a packed vector of num_cols / num_rows words (the layouts of hs_load_vector
 * and hs_read_result).  Column j of Y is what hs_run gives for column j of X: bit for bit in fixed point; in the float modes within the
 * float tolerance of the parity contract (the fused kernels add a row's partial sums in another order than the SpMV kernel).
 *   BITMAP images (dense rows -- the pruned-NN layers, which meet batches of activations in practice), one column slice: FUSED, 4
 *     then 2 columns at a time (spmm_bitmap.hip): masks and values are streamed once per group, x interleaved [column][vector];
 *     transformer-50: 6.3 us per column against 13.5 us for an SpMV (profiles/r02_spmm_bitmap.txt).  HISPARSE_SPMM_FUSED=0 turns it off.
 *   SWEEP images PLANNED for it -- hs_set_option "spmm_vectors" = 4 before the load (any element-stream matrix then takes the SWEEP format
 *     with a quarter of the rows per block; hs_run works on it as on any image, a little slower than on the planner's own choice) --: FOUR
 *     columns per pass through the matrix (spmm_sweep.hip: X interleaved [column][4], one 16-byte gather per element, four sets of row
 *     accumulators); ogbl-ppa, 16 columns: 598 us against 895 as 16 SpMVs, mouse_gene 338 against 571 (profiles/r05_spmm_sweep.txt).
 *   every other image, and a last odd column: one SpMV launch per column over the resident image (the matrix is streamed k times).
 * The context's own vector / result and its bindings are left as they were.
 *   hs_spmm_device: X and Y in device memory, column j at x_dev + j * ldx words / y_dev + j * ldy words; 16-byte aligned, ldx and ldy
 *     multiples of 4 words with ldx >= num_cols, ldy >= num_rows.  Asynchronous on the context's stream.
 *   hs_spmm: host pointers, columns back to back (ldx = num_cols, ldy = num_rows); copies in, runs, copies out, synchronous. */
int hs_spmm_device(hs_context* ctx, const void* x_dev, uint64_t ldx, void* y_dev, uint64_t ldy, uint32_t k);
int hs_spmm(hs_context* ctx, const void* packed_x, uint32_t num_cols, uint32_t k, void* packed_y, uint32_t num_rows);

/* ---- measurement ------------------------------------------------------------------------------- */
int hs_get_stats(const hs_context* ctx, hs_stats* stats);
/* `runs` back-to-back hs_run calls after `warmup` untimed ones, bracketed by HIP events on the
 * stream the kernels are launched on.  total_ms: wall time of the `runs` SpMVs (events around the
 * whole loop).  kernel_ms: sum over the runs of the duration of the SpMV kernel
 * (spmv_rowblock_kernel) alone, from per-launch event pairs.  Either output may be NULL. */
int hs_time_runs(hs_context* ctx, int warmup, int runs, float* total_ms, float* kernel_ms);
/* The dominant kernel ALONE: `runs` back-to-back launches of the SpMV kernel of hs_run (without the slice-combine pass of a
 * column-sliced plan) after `warmup` untimed ones, ONE HIP event pair around the whole loop on the stream they are launched on;
 * *kernel_ms = the elapsed time of the `runs` launches (divide by runs: the average launch duration rocprofv3 --stats reports,
 * plus the sub-microsecond dispatch gap).  No per-launch events, which add ~3 us to each launch.  Ends with one whole hs_run so that
 * y holds the product again. */
int hs_time_kernel(hs_context* ctx, int warmup, int runs, float* kernel_ms);

/* What hs_load_matrix left on the device: the image (stats.stream_bytes), Block[] (num_blocks x 320 B) and Unit[] (num_units x 64 B).
 * Tests compare this with hs_tiles_build (the host builder) byte for byte.  Any pointer may be NULL. */
int hs_debug_read_tiles(hs_context* ctx, void* image, uint64_t image_capacity, void* blocks, void* units);
/* The second image of a float BITMAP matrix (the SpMM on the matrix engine reads it): *bytes = its size (0: none); copied out when
 * `words` is not NULL.  Tests compare the image built by the device kernels with the host builder's (HISPARSE_BITMAP_BUILD=host). */
int hs_debug_read_mfma_image(hs_context* ctx, void* words, uint64_t capacity, uint64_t* bytes);

/* ---- introspection of the load-time re-tiling (host only, no GPU needed; used by the tests) -------- */
typedef struct hs_tiles hs_tiles;
/* Builds exactly what hs_load_matrix would upload for a device wanting `max_workgroups` workgroups. */
int hs_tiles_build(const void* const channel[HS_NUM_CHANNELS], const uint64_t n_packets[HS_NUM_CHANNELS], int impl, uint32_t ob_bank,
                   uint32_t vb_bank, uint32_t num_rows, uint32_t num_cols, uint32_t num_row_partitions,
                   uint32_t num_col_partitions, uint32_t max_workgroups, hs_tiles** out);
int hs_tiles_info(const hs_tiles* t, uint64_t* image_bytes, uint32_t* num_blocks, uint32_t* num_units, uint32_t* num_workgroups,
                  uint32_t* max_block_rows, uint64_t* nnz, uint64_t* elements, uint32_t* col_slices, uint32_t* ring_buffers,
                  uint32_t* stream_format);
/* image: image_bytes; blocks: num_blocks x 320 B; units: num_units x 64 B (layouts: hisparse_amd/csrc/stream_tiles.h);
 * wg_first: num_workgroups + 1; block_order: num_blocks */
int hs_tiles_copy(const hs_tiles* t, void* image, void* blocks, void* units, uint32_t* wg_first, uint32_t* block_order);
void hs_tiles_free(hs_tiles* t);
const char* hs_tiles_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* HISPARSE_HIP_H_ */
