"""Pins of the oracle against the reference's own known answers for the device path.

spmv_csim's synthetic cases (csim.cpp:443-479) use all-ones matrices and x = rand() % 2 from glibc's unseeded
generator, so the expected result is an exact integer row sum computed by compute_ref (:143-158) and checked by verify
(:160-184, absolute 1e-4).  The oracle must reproduce those exactly, in all three numeric modes, through the real
boundary (channel buffers built by the product formatter).  Rounding (AP_RND) and saturation (AP_SAT) are NOT exercised
by these cases — see tests/test_q8_24.py for the documented-semantics checks ("parity unpinned" by the reference).
"""
import ctypes

import numpy as np
import pytest

from hisparse_amd import host
from oracle import oracle as orc


def glibc_rand_mod2(count, skip=0):
    """x = rand() % 2 with glibc's default seed, the stream csim's main() draws from (csim.cpp:304,597-601)."""
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)   # the state an unseeded program starts with
    for _ in range(skip):
        libc.rand()
    return np.array([libc.rand() % 2 for _ in range(count)], dtype=np.float32)


def test_glibc_stream_prefix():
    # first draws quoted in SURVEY.md §8c
    assert glibc_rand_mod2(16).astype(int).tolist() == [1, 0, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1, 0, 1, 1, 0]


CASES = [  # (generator kind, rows, cols, nnz_per_row, draws consumed by earlier cases in csim's main order)
    ("dense", 128, 128, 0, 0),            # test_basic          csim.cpp:443-454
    ("uniform", 1000, 1024, 10, 128),     # test_basic_sparse   :456-466
    ("uniform", 20000, 20000, 10, 0),     # test_large_sparse   :468-479, shrunk from 100 000^2 to keep the CPU suite fast
]


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("kind,rows,cols,per_row,skip", CASES)
def test_csim_synthetic_cases(impl, kind, rows, cols, per_row, skip):
    csr = host.CSRMatrix.generate(kind, rows, cols, a=per_row)
    cp = host.format_matrix(csr, impl, skip_empty_rows=False)         # csim passes skip_empty_rows = false here
    x = glibc_rand_mod2(cp.num_cols, skip)
    xw = host.pack_vector(impl, x)
    y = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                 cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    ip, ix, dv = csr.arrays()                                         # padded matrix, as compute_ref sees it (csim.cpp:213,372)
    ref = orc.compute_ref(cp.num_rows, ip, ix, dv, x)
    got = orc.unpack_result(impl, y)
    assert orc.verify(ref, got) == -1                                 # csim's own acceptance test
    assert np.array_equal(got, ref)                                   # and in fact exact: integer sums <= 128 / 10
    assert np.array_equal(ref[:rows], np.round(ref[:rows])) and (got[rows:] == 0).all()


def test_per_cluster_threads_equal_sequential():
    # the cpu_baseline variant with one host thread per cluster must be the same function
    import cases
    from hisparse_amd import host
    for impl in (0, 1, 2):
        m = cases.random_csr(1500, 400, 0.04, 31, impl)
        _, cp = cases.formatted(m, impl, 4, 8 if impl == 2 else 1, True)
        xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 31, impl))
        chans = [cp.channel(c) for c in range(16)]
        args = (cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
        assert np.array_equal(orc.spmv(impl, chans, xw, *args), orc.spmv_per_channel_threads(impl, chans, xw, *args, threads=4))
