"""The reference's own formatter known answers (unit_tests/test_io.cpp) against the product formatter and the oracle."""
import os
import subprocess

import numpy as np

from oracle import cpsr_format as of

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
X = 0xFFFFFFFF


def test_product_formatter_passes_reference_goldens(tmp_path):
    exe = tmp_path / "formatter_goldens"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", f"-I{ROOT}/include", f"{ROOT}/tests/cpp/test_formatter_goldens.cpp",
                           "-o", str(exe), "-lz"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL FORMATTER GOLDENS PASSED" in out.stdout


# csr_matrix_1 / _3 of test_io.cpp:31-44,356-367
M1 = (4, 4, [1, 2, 3, 4, 5, 6, 7, 8], [0, 1, 2, 3, 0, 2, 1, 3], [0, 4, 6, 7, 8])
M3 = (8, 4, [1, 2, 3, 4, 5], [0, 2, 0, 1, 0], [0, 0, 2, 3, 3, 3, 4, 5, 5])


def test_oracle_dds_golden():
    # ConvertCsr2DDS, test_io.cpp:143-174
    rows, cols, data, idx, ptr = M1
    parts = of.convert_csr_to_dds(rows, cols, data, idx, ptr, 3)
    assert parts[0] == ([1, 2, 3, 5, 6, 7], [0, 1, 2, 0, 2, 1], [0, 3, 5, 6, 6])
    assert parts[1] == ([4, 8], [0, 0], [0, 1, 1, 1, 2])


def test_oracle_pack_rows_golden():
    # PackRows, test_io.cpp:206-245
    rows, cols, data, idx, ptr = M1
    ch = of.pack_rows(data, idx, ptr, 2, 2)
    assert ch[0] == ([[1, 5], [2, 6], [3, 0], [4, 0]], [[0, 0], [1, 2], [2, 0], [3, 0]], [[0, 0], [4, 2]])
    assert ch[1] == ([[7, 8]], [[1, 3]], [[0, 0], [1, 1]])


def test_oracle_csr2cpsr_row_partitioning_golden():
    # Csr2CpsrRowPartitioning, test_io.cpp:309-354 (marker value = integer bit pattern, sw/data_formatter.h:69-74)
    rows, cols, data, idx, ptr = M1
    cpsr, rp, cp = of.csr2cpsr(of.IMPL_FLOAT_POB, rows, cols, data, idx, ptr, 2, 4, 1, False, pack_size=2)
    assert (rp, cp) == (2, 1)
    assert cpsr[(0, 0, 0)] == ([[1, 5], [2, 6], [3, 1], [4, 0], [1, 0]], [[0, 0], [1, 2], [2, X], [3, 0], [X, 0]], [[0, 0], [5, 3]])
    assert cpsr[(1, 0, 0)] == ([[7, 8], [1, 1]], [[1, 3], [X, X]], [[0, 0], [2, 2]])


def test_oracle_csr2cpsr_skip_empty_rows_golden():
    # Csr2CpsrRowPartitioningSkipEmptyRows, test_io.cpp:370-394
    rows, cols, data, idx, ptr = M3
    cpsr, _, _ = of.csr2cpsr(of.IMPL_FLOAT_POB, rows, cols, data, idx, ptr, 8, 4, 1, True, pack_size=2)
    d, x, p = cpsr[(0, 0, 0)]
    assert d == [[1, 1], [3, 2], [2, 2], [5, 4], [1, 2]]
    assert x == [[X, 0], [0, 2], [X, X], [0, 1], [X, X]]
    assert p == [[0, 0], [1, 3], [3, 3], [3, 5], [5, 5]]


def test_oracle_round_dims_golden():
    # RoundCSRMatrixDim, test_io.cpp:121-130
    rows, cols, ptr = of.round_csr_matrix_dim(4, 4, [0, 4, 6, 7, 8], 3, 5)
    assert (rows, cols) == (6, 5) and ptr == [0, 4, 6, 7, 8, 8, 8]


def test_fixed_marker_saturates_like_ap_ufixed():
    # marker count n is stored as Q8.24 n.0; n >= 256 saturates (the decoder then reads 255: SURVEY.md Appendix B.7)
    assert of.marker_word(of.IMPL_FIXED, 1) == 1 << 24
    assert of.marker_word(of.IMPL_FIXED, 255) == 255 << 24
    assert of.marker_word(of.IMPL_FIXED, 256) == 0xFFFFFFFF
    assert of.marker_word(of.IMPL_FLOAT_STALL, 300) == 300
