/*
 * hisparse_host.h — C-ABI of the HOST side of the SpMV path (libhisparse_host.so, plain g++, no GPU).
 *
 * These entry points expose the C++ host library under include/hisparse/ (the re-implementation of
 * the reference's sw/data_loader.h + sw/data_formatter.h + the channel-assembly block of
 * sw/benchmark.cpp:127-195) to non-C++ callers: the Python binding in hisparse_amd/, the tests and
 * bench.py.  C++ callers include the headers directly, as the reference's drivers do.
 *
 * Conventions: every function returns 0 on success or a negative HSF_* code; hsf_last_error()
 * returns a thread-local message for the last failure.  Handles are opaque; the caller owns every
 * array it passes in and every handle it receives (free with the matching *_free).
 */
#ifndef HISPARSE_HOST_H_
#define HISPARSE_HOST_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { HSF_OK = 0, HSF_BAD_ARG = -1, HSF_IO_ERROR = -2, HSF_FORMAT_ERROR = -3, HSF_NO_MEMORY = -4 };

/* numeric mode == the reference's IMPL make variable (sw/Makefile:2-12) */
enum { HSF_IMPL_FIXED = 0, HSF_IMPL_FLOAT_POB = 1, HSF_IMPL_FLOAT_STALL = 2 };

typedef struct hsf_csr hsf_csr;       /* spmv::io::CSRMatrix<float>                 (sw/data_loader.h:19-30) */
typedef struct hsf_matrix hsf_matrix; /* the 16 channel packet buffers + geometry   (sw/benchmark.cpp:139)   */

typedef struct {
    int32_t impl;
    uint32_t interleave;          /* INTERLEAVE_FACTOR */
    uint32_t ob_bank, vb_bank;    /* words per output / vector bank */
    uint32_t num_rows, num_cols;  /* padded */
    uint32_t num_row_partitions, num_col_partitions;
    uint64_t nnz;                 /* true non-zeros: the GBPS/GOPS numerator (sw/benchmark.cpp:312) */
    uint64_t streamed_bytes;      /* sum of the 16 channel buffer sizes */
    int32_t skip_empty_rows;
} hsf_matrix_info;

const char* hsf_last_error(void);

/* ---- CSR ingest (sw/data_loader.h) -------------------------------------------------------------- */
/* load_csr_matrix_from_float_npz (:51-70) */
int hsf_csr_load_npz(const char* path, hsf_csr** out);
/* create_csr_matrix (:35-47); arrays are copied */
int hsf_csr_from_arrays(uint32_t num_rows, uint32_t num_cols, uint64_t nnz, const uint32_t* indptr,
                        const uint32_t* indices, const float* data, hsf_csr** out);
int hsf_csr_dims(const hsf_csr* m, uint32_t* num_rows, uint32_t* num_cols, uint64_t* nnz);
/* copy the three CSR arrays out (any pointer may be NULL to skip it) */
int hsf_csr_copy(const hsf_csr* m, uint32_t* indptr, uint32_t* indices, float* data);
/* overwrite all values with one constant (the drivers' `x = 1 / num_cols`, sw/benchmark.cpp:411) */
int hsf_csr_fill(hsf_csr* m, float value);
/* util_normalize_csr_matrix_by_outdegree (sw/data_formatter.h:33-47): value := 1 / (non-zeros in the value's column) */
int hsf_csr_normalize_by_outdegree(hsf_csr* m);
void hsf_csr_free(hsf_csr* m);

/* ---- synthetic stand-ins for the absent datasets -------------------------------- */
/* Seeded generators, deterministic for a given argument list regardless of thread count.
 * kind: "uniform"  — csim's create_uniform_sparse_CSR (spmv_csim/csim.cpp:411-435): a = nnz_per_row, all ones
 *       "dense"    — csim's create_dense_CSR (:387-409), all ones
 *       "powerlaw" — Chung-Lu style graph: a = target nnz, b = skew exponent beta; values uniform(0, c)
 *       "bernoulli"— pruned-NN layer: every entry present with probability b; values N(0, c)
 *       "rmat"     — symmetric R-MAT graph, quadrant probabilities .57 / .19 / .19 / .05: a = entries DRAWN (both directions; duplicates are dropped),
 *                    b != 0 scrambles the vertex ids; values uniform(0, c); square matrices only
 */
int hsf_csr_generate(const char* kind, uint32_t num_rows, uint32_t num_cols, double a, double b, double c,
                     uint64_t seed, hsf_csr** out);

/* ---- CSR -> CPSR -> 16 channel buffers (sw/data_formatter.h, sw/benchmark.cpp:110-195) ----------- */
/* Pads the CSR handle's dimensions in place (util_round_csr_matrix_dim mutates its argument too). */
int hsf_format(hsf_csr* m, int impl, uint32_t ob_bank, uint32_t vb_bank, int skip_empty_rows, hsf_matrix** out);
int hsf_matrix_get_info(const hsf_matrix* m, hsf_matrix_info* info);
/* channel c in [0,16): pointer to its 64-byte packets and their count */
int hsf_matrix_channel(const hsf_matrix* m, uint32_t c, const void** packets, uint64_t* num_packets);
/* `part_len` kernel argument of row partition j (sw/benchmark.cpp:301-322) */
int hsf_matrix_part_len(const hsf_matrix* m, uint32_t row_partition, uint32_t* part_len);
void hsf_matrix_free(hsf_matrix* m);

/* ---- x / y word conversion (sw/benchmark.cpp:207-212, spmv_csim/csim.cpp:172) -------------------- */
int hsf_pack_vector(int impl, const float* x, uint64_t n, uint32_t* words);
int hsf_unpack_result(int impl, const uint32_t* words, uint64_t n, float* y);

/* ---- CSC for SpMSpV (sw/data_loader.h:109-157: csr2csc + csc_matrix_convert_from_float) ------------------------------ */
/* indptr: num_cols + 1; row_indices / value_words: nnz each (value words in the numeric mode's representation, like hsf_pack_vector). */
int hsf_csr_to_csc(const hsf_csr* m, int impl, uint32_t* indptr, uint32_t* row_indices, uint32_t* value_words);

/* ---- multi-GPU row slabs (hisparse/row_sharding.h; the C++ benchmark's --gpus N uses the same routine) ----------- */
/* bounds[0..parts]: row boundaries balancing non-zeros, interior ones multiples of `granule` (128 * interleave). */
int hsf_split_rows_by_nnz(const uint32_t* indptr, uint32_t num_rows, uint32_t parts, uint32_t granule, uint32_t* bounds);

#ifdef __cplusplus
}
#endif

#endif /* HISPARSE_HOST_H_ */
