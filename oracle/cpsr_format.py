"""oracle/cpsr_format.py — plain-Python restatement of the reference's host pre-processing.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and by tests/golden/make_golden.py); the product's
formatter is the C++ in include/hisparse/data_formatter.h + channel_packets.h.

Each function follows one routine of /root/reference/sw/data_formatter.h or one block of
/root/reference/sw/benchmark.cpp and says which.  Values travel as raw 32-bit words (Q8.24 words in
fixed mode, IEEE-754 bit patterns in the float modes) so outputs compare byte for byte with the
product's channel buffers.  Pure loops: use on small matrices only.

Pinned against the reference's own goldens (unit_tests/test_io.cpp:143-390) in
tests/test_oracle_pins.py.
"""
import math
import struct

import numpy as np

PACK_SIZE = 8            # spmv/libfpga/common.h:30
NUM_HBM_CHANNELS = 16    # common.h:173-176
IDX_MARKER = 0xFFFFFFFF  # common.h:8
IMPL_FIXED, IMPL_FLOAT_POB, IMPL_FLOAT_STALL = 0, 1, 2


def interleave_factor(impl):
    """INTERLEAVE_FACTOR: spmv/libfpga/common.h:169, spmv-fp/libfpga/common.h:181,187."""
    return 8 if impl == IMPL_FLOAT_STALL else 1


# ---- value words -----------------------------------------------------------------------------

def q_from_float(f):
    """float -> ap_ufixed<32,8,AP_RND,AP_SAT> raw word (sw/data_loader.h:80)."""
    d = float(np.float32(f))
    if not d > 0.0:
        return 0
    s = math.floor(d * 16777216.0 + 0.5)
    return 0xFFFFFFFF if s >= 4294967296 else int(s)


def q_from_uint(n):
    """integer n -> raw word of n.0, saturating (sw/data_formatter.h:73,158)."""
    return 0xFFFFFFFF if n >= 256 else n << 24


def f32_bits(f):
    return struct.unpack("<I", struct.pack("<f", float(np.float32(f))))[0]


def value_word(impl, f):
    return q_from_float(f) if impl == IMPL_FIXED else f32_bits(f)


def marker_word(impl, n):
    """fixed: n.0 as Q8.24 (:73,158); float: the integer's bit pattern (:69-71,154-156)."""
    return q_from_uint(n) if impl == IMPL_FIXED else n & 0xFFFFFFFF


# ---- sw/data_formatter.h ---------------------------------------------------------------------

def round_csr_matrix_dim(num_rows, num_cols, indptr, row_divisor, col_divisor):
    """util_round_csr_matrix_dim (:15-29): returns (rows, cols, indptr) padded."""
    indptr = list(indptr)
    if num_rows % row_divisor:
        pad = row_divisor - num_rows % row_divisor
        indptr += [indptr[num_rows]] * pad
        num_rows += pad
    if num_cols % col_divisor:
        num_cols += col_divisor - num_cols % col_divisor
    return num_rows, num_cols, indptr


def convert_csr_to_dds(num_rows, num_cols, data, indices, indptr, cols_per_partition):
    """util_convert_csr_to_dds (:256-313): list of (data, indices, indptr) per column partition."""
    parts = (num_cols + cols_per_partition - 1) // cols_per_partition
    out = [([], [], [0]) for _ in range(parts)]
    for r in range(num_rows):
        for e in range(indptr[r], indptr[r + 1]):
            p = indices[e] // cols_per_partition
            out[p][0].append(data[e])
            out[p][1].append(indices[e] - p * cols_per_partition)  # partition-local id (:308-309)
        for p in range(parts):
            out[p][2].append(len(out[p][0]))
    return out


def pad_marker_end_of_row(impl, data, indices, indptr, stride, skip_empty_rows):
    """util_pad_marker_end_of_row (:175-187) and its two variants (:51-83, :87-171)."""
    rows = len(indptr) - 1
    nd, ni, nptr = [], [], [0]
    if not skip_empty_rows:
        for r in range(rows):
            nd += data[indptr[r]:indptr[r + 1]]
            ni += indices[indptr[r]:indptr[r + 1]]
            nd.append(marker_word(impl, 1))
            ni.append(IDX_MARKER)
            nptr.append(len(nd))
        return nd, ni, nptr
    assert rows % stride == 0
    # :96-110 — the first `stride` rows always count as non-empty
    empty = [(r >= stride) and indptr[r + 1] == indptr[r] for r in range(rows)]
    # :120-139 — marker value = 1 + directly following empty rows in the same residue class
    val = [0 if empty[r] else 1 for r in range(rows)]
    for k in range(stride):
        r = k
        while r < rows:
            nxt = r + stride
            if not empty[r]:
                while nxt < rows and empty[nxt]:
                    val[r] += 1
                    nxt += stride
            r = nxt
    for r in range(rows):
        if not empty[r]:
            nd += data[indptr[r]:indptr[r + 1]]
            ni += indices[indptr[r]:indptr[r + 1]]
            nd.append(marker_word(impl, val[r]))
            ni.append(IDX_MARKER)
        nptr.append(len(nd))
    return nd, ni, nptr


def pack_rows(data, indices, indptr, channels, pack_size):
    """util_pack_rows (:384-446): per channel (data[n][pack], indices[n][pack], indptr[rounds+1][pack])."""
    rows = len(indptr) - 1
    rounds = (rows + channels * pack_size - 1) // (channels * pack_size)
    out = []
    for c in range(channels):
        running = [0] * pack_size
        ptr = [list(running)]
        for i in range(rounds):
            for j in range(pack_size):
                r = i * channels * pack_size + c * pack_size + j  # :410
                if r < rows:
                    running[j] += indptr[r + 1] - indptr[r]
            ptr.append(list(running))
        longest = max(running)
        d = [[0] * pack_size for _ in range(longest)]   # zero filled to the longest lane (:421-428)
        x = [[0] * pack_size for _ in range(longest)]
        for j in range(pack_size):
            at = 0
            for i in range(rounds):
                r = i * channels * pack_size + c * pack_size + j  # :432
                if r >= rows:
                    continue
                for e in range(indptr[r], indptr[r + 1]):
                    d[at][j] = data[e]
                    x[at][j] = indices[e]
                    at += 1
        out.append((d, x, ptr))
    return out


def csr2cpsr(impl, num_rows, num_cols, data_words, indices, indptr, out_buf_len, vec_buf_len, channels,
             skip_empty_rows, pack_size=PACK_SIZE):
    """csr2cpsr (:468-544).  Returns dict[(row_part, col_part, channel)] -> (data, indices, indptr)."""
    assert num_rows % (pack_size * channels) == 0 and num_cols % pack_size == 0          # :475-488
    assert out_buf_len % (pack_size * channels) == 0 and vec_buf_len % pack_size == 0    # :489-490
    row_parts = (num_rows + out_buf_len - 1) // out_buf_len
    col_parts = (num_cols + vec_buf_len - 1) // vec_buf_len
    cpsr = {}
    for j in range(row_parts):
        rows_here = out_buf_len if j < row_parts - 1 else num_rows - (row_parts - 1) * out_buf_len  # :504-507
        base = indptr[j * out_buf_len]
        local = [indptr[j * out_buf_len + r] - base for r in range(rows_here + 1)]                   # :508-511
        dds = convert_csr_to_dds(rows_here, num_cols, data_words[base:], indices[base:], local, vec_buf_len)
        for i in range(col_parts):
            d, x, p = pad_marker_end_of_row(impl, dds[i][0], dds[i][1], dds[i][2], channels * pack_size, skip_empty_rows)
            for c, triple in enumerate(pack_rows(d, x, p, channels, pack_size)):
                cpsr[(j, i, c)] = triple
    return cpsr, row_parts, col_parts


# ---- sw/benchmark.cpp:127-195 (same block: csim.cpp:229-297, host.cpp:163-231) ---------------

def assemble_channel_packets(impl, cpsr, row_parts, col_parts):
    """16 numpy arrays of shape (n_packets, 16) uint32: words 0-7 indices, 8-15 values."""
    F = interleave_factor(impl)
    parts = row_parts * col_parts
    out = []
    for pc in range(NUM_HBM_CHANNELS):
        starts, lens, payload = [], [], [[] for _ in range(F)]
        start = 0
        for j in range(row_parts):
            for i in range(col_parts):
                per_vc = []
                for f in range(F):
                    vc = pc + f * NUM_HBM_CHANNELS                       # :146
                    per_vc.append(cpsr[(j, i, vc)][2][-1])                # indptr.back() = lane lengths
                longest = max(max(l) for l in per_vc)                    # :148-153
                for f in range(F):
                    vc = pc + f * NUM_HBM_CHANNELS
                    d, x, _ = cpsr[(j, i, vc)]
                    pk = [x[p] + d[p] for p in range(len(x))]
                    pk += [[0] * 16] * (longest - len(pk))               # resize(start + max_num_packets) (:160-163)
                    payload[f] += pk
                starts.append(start)
                lens.append(per_vc)
                start += longest                                         # :167-170
        n_payload = len(payload[0])
        buf = np.zeros((parts * (1 + F) + n_payload * F, 16), dtype=np.uint32)   # :175
        for ij in range(parts):
            buf[ij * (1 + F), 0] = starts[ij] * F                        # :178-179
            for f in range(F):
                buf[ij * (1 + F) + 1 + f, 0:8] = lens[ij][f]             # :182
        off = parts * (1 + F)                                            # :186
        for p in range(n_payload):
            for f in range(F):
                buf[off + p * F + f, :] = payload[f][p]                  # :190-192
        out.append(buf)
    return out


def format_matrix(impl, num_rows, num_cols, data_f32, indices, indptr, ob_bank, vb_bank, skip_empty_rows):
    """Round dims, convert values, csr2cpsr, assemble (sw/benchmark.cpp:110-195).
    Returns (channels, padded_rows, padded_cols, row_parts, col_parts)."""
    F = interleave_factor(impl)
    rows, cols, ptr = round_csr_matrix_dim(num_rows, num_cols, [int(v) for v in indptr],
                                           PACK_SIZE * NUM_HBM_CHANNELS * F, PACK_SIZE)
    words = [value_word(impl, v) for v in data_f32]
    cpsr, rp, cp = csr2cpsr(impl, rows, cols, words, [int(v) for v in indices], ptr,
                            ob_bank * PACK_SIZE * NUM_HBM_CHANNELS, vb_bank * PACK_SIZE,
                            NUM_HBM_CHANNELS * F, skip_empty_rows)
    return assemble_channel_packets(impl, cpsr, rp, cp), rows, cols, rp, cp
