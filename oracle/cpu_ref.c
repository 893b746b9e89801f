/*
 * oracle/cpu_ref.c — CPU restatement of the HiSparse SpMV device path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this; the product
 * (libhisparse_hip.so, hisparse_amd/) never links, imports or executes anything under oracle/.
 *
 * What it restates: the functional behaviour of spmv_csim's `top_wrapper`
 * (/root/reference/spmv_csim/csim.cpp:22-136) — the chain
 *   spmv_vector_loader -> spmv_sk0/1/2 (16 x spmv_cluster) -> spmv_result_drain
 * — consuming exactly the boundary inputs (16 channel packet buffers, packed x, five scalars) and
 * producing packed y, for the three numeric modes.  Each step cites the reference lines it follows.
 * The FIFO/arbiter micro-architecture (hls::stream, shuffler_core) is NOT emulated: its functional
 * contract is "route each payload to lane addr%8, order within a lane unspecified"
 * (spmv/libfpga/shuffle.h:49-50,127-128; unit_tests/test_shuffle.cpp:129,215-241), which for the
 * fixed-point mode cannot change the result (saturating sums of non-negative terms are order free)
 * and for the float modes only changes the association of the fp32 sum (tolerance parity).
 *
 * PARITY PINS.  The reference cannot be built in this image (it needs the Vitis HLS 2020.2 headers
 * ap_fixed.h / ap_int.h / hls_stream.h / ap_axi_sdata.h and the cnpy library, none of which are in
 * /root/reference or on this machine), so this oracle is pinned by the reference's own known
 * answers instead: the csim synthetic cases (csim.cpp:443-479, expected y = integer row sums from
 * compute_ref :143-158) and the formatter goldens of unit_tests/test_io.cpp (see
 * tests/test_oracle_pins.py).  Those never exercise AP_RND rounding or AP_SAT saturation
 * (all-ones matrices, x in {0,1}): for those two rules the arithmetic below follows the documented
 * semantics of ap_ufixed<32,8,AP_RND,AP_SAT> and is "parity unpinned" against the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P 8u           /* PACK_SIZE, spmv/libfpga/common.h:30 */
#define C 16u          /* NUM_HBM_CHANNELS = 4 + 6 + 6, common.h:173-176 */
#define MARKER 0xffffffffu /* IDX_MARKER, common.h:8 */
#define POB_DEPTH 7u   /* DEP_DISTANCE = 1 + FPADD_LATENCY + 1 + 1, spmv-fp/libfpga/common.h:175,179 */

enum { IMPL_FIXED = 0, IMPL_FLOAT_POB = 1, IMPL_FLOAT_STALL = 2 };

enum {
    ORACLE_OK = 0,
    ORACLE_BAD_ARG = -1,
    ORACLE_ROW_OUT_OF_RANGE = -2, /* a decoded row index falls outside the partition's output bank */
    ORACLE_NO_MEMORY = -3,
    ORACLE_COL_OUT_OF_RANGE = -4, /* a column index falls outside the vector bank */
};

typedef struct {
    uint32_t idx[P];
    uint32_t val[P];
} mat_pkt_t; /* SPMV_MAT_PKT_T, common.h:44-50: indices first, then values */

/* ---- ap_ufixed<32,8,AP_RND,AP_SAT> on raw words (common.h:35-38) ------------------------------ */

/* float -> VAL_T (sw/data_loader.h:80, sw/benchmark.cpp:210): round half up, saturate, negatives -> 0 */
uint32_t oracle_q_from_float(float f) {
    double d = (double)f;
    if (!(d > 0.0)) return 0u;
    double s = floor(d * 16777216.0 + 0.5);
    return s >= 4294967296.0 ? 0xffffffffu : (uint32_t)s;
}
/* VAL_T -> float (spmv_csim/csim.cpp:172) */
float oracle_q_to_float(uint32_t raw) { return (float)((double)raw / 16777216.0); }
/* mat_val * vec_val narrowed to VAL_T (spmv/libfpga/pe.h:64): exact Q16.48, + half LSB, >> 24, clamp */
static uint32_t q_mul(uint32_t a, uint32_t b) {
    uint64_t w = (uint64_t)a * (uint64_t)b;
    uint64_t r = (w >> 24) + ((w >> 23) & 1u);
    return r > 0xffffffffull ? 0xffffffffu : (uint32_t)r;
}
/* q + incr narrowed to VAL_T (pe.h:72): exact Q9.24, clamp */
static uint32_t q_add(uint32_t a, uint32_t b) {
    uint64_t s = (uint64_t)a + (uint64_t)b;
    return s > 0xffffffffull ? 0xffffffffu : (uint32_t)s;
}
static float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exported for property tests of the arithmetic */
uint32_t oracle_q_mul(uint32_t a, uint32_t b) { return q_mul(a, b); }
uint32_t oracle_q_add(uint32_t a, uint32_t b) { return q_add(a, b); }

static unsigned interleave_of(int impl) { return impl == IMPL_FLOAT_STALL ? 8u : 1u; } /* spmv-fp common.h:181,187 */

/*
 * One launch of the five kernels for one row partition == csim's top_wrapper (csim.cpp:22-46).
 *   ch[16]      channel packet buffers            (matrix_hbm_0..15)
 *   x           packed dense vector, num_cols words
 *   y           packed dense result (only this row partition's packets are written)
 *   part_len    rows per cluster in this row partition (`rows_per_c_in_partition`)
 *   ob_bank/vb_bank  OB_BANK_SIZE / VB_BANK_SIZE the "bitstream" was built with (runtime here)
 */
static int top_wrapper_threads(int impl, const void* const ch[C], const uint32_t* x, uint32_t* y, uint32_t row_part_id,
                               uint32_t part_len, uint32_t num_col_partitions, uint32_t num_partitions, uint32_t num_cols,
                               uint32_t ob_bank, uint32_t vb_bank, int threads) {
    if (impl < IMPL_FIXED || impl > IMPL_FLOAT_STALL || !ch || !x || !y || ob_bank == 0 || vb_bank == 0) return ORACLE_BAD_ARG;
    if (part_len % P != 0) return ORACLE_BAD_ARG;
    const unsigned F = interleave_of(impl);
    const uint32_t used = part_len / P;            /* used_buf_len, spmv_cluster.h:330 */
    if (used > ob_bank) return ORACLE_BAD_ARG;
    const uint64_t logical_vb = (uint64_t)vb_bank * P; /* LOGICAL_VB_SIZE, common.h:179 */
    const uint64_t logical_ob = (uint64_t)ob_bank * P * C;
    /* spmv_vector_loader.cpp:13-19: the loader derives the column partitioning from num_cols */
    const uint32_t vl_parts = (uint32_t)((num_cols + logical_vb - 1) / logical_vb);
    const unsigned nbuf = impl == IMPL_FLOAT_POB ? POB_DEPTH : 1u;

    int status = ORACLE_OK;
    /* The 16 clusters share nothing but x (read) and write disjoint y packets (spmv_sk0/1/2 run them concurrently on
       the FPGA): `threads` > 1 gives every cluster its own banks and its own host thread -- CPU baseline (B) of SURVEY
       section 8(d); threads == 1 is the csim order. */
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 1 ? threads : 1)
    for (int pci = 0; pci < (int)C; ++pci) { /* 16 clusters, spmv_sk0.cpp:43-114 */
        const unsigned pc = (unsigned)pci;
        int rc = ORACLE_OK;
        uint32_t* bank = (uint32_t*)malloc((size_t)P * vb_bank * sizeof(uint32_t));          /* 8 vector banks */
        uint32_t* ob = (uint32_t*)malloc((size_t)nbuf * P * (used ? used : 1) * sizeof(uint32_t)); /* 8 output banks (x7 for pob) */
        if (!bank || !ob) {
            free(bank); free(ob);
#pragma omp critical
            status = ORACLE_NO_MEMORY;
            continue;
        }
        memset(bank, 0, (size_t)P * vb_bank * sizeof(uint32_t));
        const mat_pkt_t* m = (const mat_pkt_t*)ch[pc];
        /* pe.h:131-135 (pe-pob.h:124-130): zero the used part of the output banks */
        memset(ob, 0, (size_t)nbuf * P * (used ? used : 1) * sizeof(uint32_t));
        const uint64_t payload_base = (uint64_t)(1 + F) * num_partitions; /* spmv_cluster.h:41, fp :46 */

        for (uint32_t cp = 0; cp < num_col_partitions && rc == ORACLE_OK; ++cp) {
            /* -- vector path: loader -> unpacker -> vecbuf_writer ------------------------------- */
            if (cp < vl_parts) {
                uint64_t cols_here = logical_vb;
                if (cp == vl_parts - 1 && num_cols % logical_vb != 0) cols_here = num_cols % logical_vb; /* :14-19,34-37 */
                for (uint64_t i = 0; i < cols_here / P; ++i) {
                    uint64_t dv_idx = i + (uint64_t)cp * vb_bank; /* spmv_vector_loader.cpp:44 */
                    for (unsigned k = 0; k < P; ++k)              /* unpacker idx = pkt*8+k (spmv_cluster.h:121); */
                        bank[(size_t)k * vb_bank + dv_idx % vb_bank] = x[dv_idx * P + k]; /* writer: bank k, addr (idx/8)%size (vecbuf_access_unit.h:71) */
                }
            }
            /* -- matrix path: CPSR_matrix_loader (spmv_cluster.h:34-107, fp :39-129) ---------- */
            const uint32_t pid = row_part_id * num_col_partitions + cp;
            const uint64_t info = (uint64_t)(1 + F) * pid;
            const uint32_t start = m[info].idx[0];
            uint32_t len[8][P];
            uint32_t row_idx[8][P];
            uint32_t longest = 0;
            for (unsigned f = 0; f < F; ++f)
                for (unsigned k = 0; k < P; ++k) {
                    len[f][k] = m[info + 1 + f].idx[k];
                    if (len[f][k] > longest) longest = len[f][k];
                    row_idx[f][k] = f * P + k; /* fp :84; fixed :62 with f = 0 */
                }
            uint32_t arrivals[P]; /* per-PE arrival counter, reset per column partition (pe-pob.h:38) */
            memset(arrivals, 0, sizeof(arrivals));
            const uint64_t reads = (uint64_t)longest * F;
            for (uint64_t i = 0; i < reads && rc == ORACLE_OK; ++i) {
                const mat_pkt_t* pkt = &m[payload_base + start + i];
                const unsigned f = (unsigned)(i % F);
                const uint64_t pos = i / F;
                for (unsigned k = 0; k < P; ++k) {
                    if (pos >= len[f][k]) continue;       /* lane exhausted: padding, ignored */
                    const uint32_t col = pkt->idx[k], val = pkt->val[k];
                    if (col == MARKER) {
                        /* fixed: integer part of the Q8.24 word (spmv_cluster.h:82); float: raw bits (fp :104) */
                        const uint32_t n = impl == IMPL_FIXED ? (val >> 24) : val;
                        row_idx[f][k] += P * n * F;
                        continue;
                    }
                    /* shuffle 1 routes by col % 8 to VAU lane; reader attaches bank[(col/8) % size]
                       (vecbuf_access_unit.h:126-128).  col is partition-local. */
                    if (col >= logical_vb) { rc = ORACLE_COL_OUT_OF_RANGE; break; }
                    const uint32_t xv = bank[(size_t)(col % P) * vb_bank + (col / P) % vb_bank];
                    /* shuffle 2 routes by row % 8 to the PE; PE address = row / 8 (pe.h:63) */
                    const uint32_t row = row_idx[f][k];
                    const unsigned pe = row % P;
                    const uint32_t addr = row / P;
                    if (addr >= used) { rc = ORACLE_ROW_OUT_OF_RANGE; break; }
                    if (impl == IMPL_FIXED) {
                        uint32_t* q = &ob[(size_t)pe * used + addr];
                        *q = q_add(*q, q_mul(val, xv)); /* pe.h:64,72 */
                    } else {
                        /* float_pob: which of the DEP_DISTANCE = 7 partial buffers a product lands in.  NOT the reference's rule word for
                           word: pe-pob.h:71 advances pb_idx once per pipeline CYCLE, outside `if (valid)` (:61-69), i.e. also in
                           cycles where the PE's input stream is empty -- which buffer an arrival meets depends on the shuffle
                           arbiter's cycle-level back-pressure, which neither csim (C simulation: a dataflow process runs after
                           its producers, so read_nb finds every payload waiting until EOD) nor this restatement models.  Here the
                           index advances once per ARRIVAL at the PE -- what the reference's loop does under C simulation.  The
                           difference can only change the ASSOCIATION of one fp32 row sum (same products, same 7-way split of a
                           sum that the dump adds up again, pe-pob.h:91-95), never the set of terms: it is inside the float
                           contract's tolerance by construction, and irrelevant in fixed point. */
                        const unsigned d = impl == IMPL_FLOAT_POB ? (arrivals[pe]++ % POB_DEPTH) : 0u;
                        uint32_t* q = &ob[((size_t)d * P + pe) * used + addr];
                        volatile float incr = bits2f(val) * bits2f(xv); /* separate multiply, then add (pe-pob.h:63-65, pe-stall.h:52,138) */
                        *q = f2bits(bits2f(*q) + incr);
                    }
                }
            }
        }
        if (rc != ORACLE_OK) {
            free(bank); free(ob);
#pragma omp critical
            status = rc;
            continue;
        }
        /* -- PE dump -> result packer -> axis_merge -> result drain ------------------------------
           packet n of cluster pc lands at y packet row_part*LOGICAL_OB/8 + n*16 + pc
           (spmv_result_drain.cpp:36,43-113 with the 4/6/6 round robin; stream_utils.h:46-62). */
        for (uint32_t n = 0; n < used; ++n) {
            uint32_t* dst = y + ((uint64_t)row_part_id * logical_ob / P + (uint64_t)n * C + pc) * P;
            for (unsigned k = 0; k < P; ++k) {
                if (impl == IMPL_FLOAT_POB) {
                    float q = 0.0f; /* pe-pob.h:91-95 */
                    for (unsigned d = 0; d < POB_DEPTH; ++d) q += bits2f(ob[((size_t)d * P + k) * used + n]);
                    dst[k] = f2bits(q);
                } else {
                    dst[k] = ob[(size_t)k * used + n];
                }
            }
        }
        free(bank);
        free(ob);
    }
    return status;
}

int oracle_top_wrapper(int impl, const void* const ch[C], const uint32_t* x, uint32_t* y, uint32_t row_part_id,
                       uint32_t part_len, uint32_t num_col_partitions, uint32_t num_partitions, uint32_t num_cols,
                       uint32_t ob_bank, uint32_t vb_bank) {
    return top_wrapper_threads(impl, ch, x, y, row_part_id, part_len, num_col_partitions, num_partitions, num_cols, ob_bank, vb_bank, 1);
}

/* All row partitions, the way every driver loops them (sw/benchmark.cpp:301-338, csim.cpp:329-365). */
static int spmv_threads(int impl, const void* const ch[C], const uint32_t* x, uint32_t* y, uint32_t num_rows, uint32_t num_cols,
                        uint32_t num_row_partitions, uint32_t num_col_partitions, uint32_t ob_bank, uint32_t vb_bank, int threads) {
    const uint64_t logical_ob = (uint64_t)ob_bank * P * C;
    uint32_t last = (uint32_t)(num_rows % logical_ob == 0 ? logical_ob / C : (num_rows % logical_ob) / C);
    for (uint32_t rp = 0; rp < num_row_partitions; ++rp) {
        uint32_t part_len = rp == num_row_partitions - 1 ? last : (uint32_t)(logical_ob / C);
        int rc = top_wrapper_threads(impl, ch, x, y, rp, part_len, num_col_partitions, num_row_partitions * num_col_partitions,
                                     num_cols, ob_bank, vb_bank, threads);
        if (rc != ORACLE_OK) return rc;
    }
    return ORACLE_OK;
}

int oracle_spmv(int impl, const void* const ch[C], const uint32_t* x, uint32_t* y, uint32_t num_rows, uint32_t num_cols,
                uint32_t num_row_partitions, uint32_t num_col_partitions, uint32_t ob_bank, uint32_t vb_bank) {
    return spmv_threads(impl, ch, x, y, num_rows, num_cols, num_row_partitions, num_col_partitions, ob_bank, vb_bank, 1);
}

/* Same result, one host thread per cluster (at most 16 are useful).  Only bench.py's cpu_baseline leg calls it. */
int oracle_spmv_per_channel_threads(int impl, const void* const ch[C], const uint32_t* x, uint32_t* y, uint32_t num_rows,
                                    uint32_t num_cols, uint32_t num_row_partitions, uint32_t num_col_partitions, uint32_t ob_bank,
                                    uint32_t vb_bank, int threads) {
    return spmv_threads(impl, ch, x, y, num_rows, num_cols, num_row_partitions, num_col_partitions, ob_bank, vb_bank, threads);
}

/* compute_ref (csim.cpp:143-158): float32 CSR loop, accumulation in float, in CSR order. */
void oracle_compute_ref(uint32_t num_rows, const uint32_t* indptr, const uint32_t* indices, const float* data,
                        const float* x, float* y) {
    for (uint32_t r = 0; r < num_rows; ++r) {
        volatile float acc = 0.0f;
        for (uint32_t e = indptr[r]; e < indptr[r + 1]; ++e) {
            volatile float prod = data[e] * x[indices[e]];
            acc = acc + prod;
        }
        y[r] = acc;
    }
}

/* verify (csim.cpp:160-184): |kernel - reference| < 1e-4 ABSOLUTE, first failing index or -1. */
int64_t oracle_verify(const float* reference, const float* kernel, uint64_t n) {
    const float epsilon = 0.0001f;
    for (uint64_t i = 0; i < n; ++i)
        if (!(fabsf(kernel[i] - reference[i]) < epsilon)) return (int64_t)i;
    return -1;
}

/* The same float32 CSR loop spread over all host cores (rows are independent): the "honest CPU SpMV" context number of
 * BASELINE.md §3 (baseline C).  Only bench.py's cpu_baseline leg calls it. */
void oracle_compute_ref_parallel(uint32_t num_rows, const uint32_t* indptr, const uint32_t* indices, const float* data,
                                 const float* x, float* y, int threads) {
#pragma omp parallel for schedule(dynamic, 1024) num_threads(threads > 0 ? threads : 1)
    for (int64_t r = 0; r < (int64_t)num_rows; ++r) {
        float acc = 0.0f;
        for (uint32_t e = indptr[r]; e < indptr[r + 1]; ++e) acc += data[e] * x[indices[e]];
        y[r] = acc;
    }
}


/*
 * SpMSpV (EXTENSION; the reference only stubs it: SPMSPV_MAT_PKT_T / IDX_VAL_T in spmv/libfpga/common.h:52-54, csr2csc in
 * sw/data_loader.h:109-144, paper section 7): y = A x for a SPARSE x given as (index, value) pairs over a CSC matrix -- for every
 * stored x entry, its column's non-zeros are multiplied and accumulated with the PE arithmetic of the numeric mode
 * (pe.h:64,72 / pe-stall.h:52,138).  The row sums equal those of the dense SpMV with x scattered into a zero vector: bit for bit
 * in fixed point (saturating sums of non-negative terms are order free), up to association in the float modes.
 */
int oracle_spmspv(int impl, const uint32_t* indptr, const uint32_t* row_indices, const uint32_t* value_words, uint32_t num_rows,
                  uint32_t num_cols, const uint32_t* x_index, const uint32_t* x_words, uint32_t x_count, uint32_t* y) {
    if (impl < IMPL_FIXED || impl > IMPL_FLOAT_STALL || !indptr || !y || (x_count && (!x_index || !x_words))) return ORACLE_BAD_ARG;
    memset(y, 0, (size_t)num_rows * 4);
    for (uint32_t k = 0; k < x_count; ++k) {
        const uint32_t c = x_index[k];
        if (c >= num_cols) return ORACLE_COL_OUT_OF_RANGE;
        for (uint32_t e = indptr[c]; e < indptr[c + 1]; ++e) {
            const uint32_t r = row_indices[e];
            if (r >= num_rows) return ORACLE_ROW_OUT_OF_RANGE;
            if (impl == IMPL_FIXED) y[r] = q_add(y[r], q_mul(value_words[e], x_words[k]));
            else y[r] = f2bits(bits2f(y[r]) + bits2f(value_words[e]) * bits2f(x_words[k]));
        }
    }
    return ORACLE_OK;
}
