"""The product's host pipeline (C++: round dims -> convert -> csr2cpsr -> channel assembly) byte for byte against the
oracle's plain-Python restatement of sw/data_formatter.h + sw/benchmark.cpp:127-195, on seeded inputs."""
import numpy as np
import pytest
import scipy.sparse as sp

from hisparse_amd import host
from oracle import cpsr_format as of
from oracle import oracle as orc

import cases

GEOMS = [  # (impl, vb_bank, ob_bank): small banks force many partitions; float_stall needs ob % 8 == 0
    (0, 2, 1), (0, 16, 8), (1, 2, 1), (1, 4, 2), (2, 2, 8), (2, 8, 16),
]


@pytest.mark.parametrize("impl,vb,ob", GEOMS)
@pytest.mark.parametrize("skip", [False, True])
def test_channel_buffers_match_oracle(impl, vb, ob, skip):
    m = cases.random_csr(700, 90, 0.04, 100 + impl, impl)
    csr, cp = cases.formatted(m, impl, vb, ob, skip)
    ref, rows, cols, rp, cpn = of.format_matrix(impl, 700, 90, m.data, m.indices, m.indptr, ob, vb, skip)
    assert (cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions) == (rows, cols, rp, cpn)
    assert cp.nnz == m.nnz and cp.interleave == of.interleave_factor(impl)
    for c in range(16):
        got = cp.channel(c)
        assert got.shape == ref[c].shape, f"channel {c}"
        assert np.array_equal(got, ref[c]), f"channel {c}"
    # the handle's dimensions were padded in place, like util_round_csr_matrix_dim does (sw/benchmark.cpp:110)
    assert csr.num_rows == rows and csr.num_cols == cols


def test_part_len_follows_reference_rule():
    # sw/benchmark.cpp:301-322: LOGICAL_OB / 16 for every partition but the last, (rows % LOGICAL_OB) / 16 there
    m = cases.random_csr(700, 40, 0.05, 3, 0)
    _, cp = cases.formatted(m, 0, 2, 2, True)   # LOGICAL_OB = 2*8*16 = 256 rows; 700 -> 768 rows -> 3 partitions
    assert cp.num_rows == 768 and cp.num_row_partitions == 3
    assert [cp.part_len(j) for j in range(3)] == [16, 16, 16]
    _, cp = cases.formatted(cases.random_csr(600, 40, 0.05, 3, 0), 0, 2, 2, True)   # 640 rows: last partition 128 rows
    assert [cp.part_len(j) for j in range(cp.num_row_partitions)] == [16, 16, 8]


def test_skip_counts_above_one_and_empty_matrix_rows():
    dense = np.zeros((1024, 96), dtype=np.float32)
    dense[5, 3] = 1.5
    dense[5 + 128 * 4, 40] = 0.25
    dense[1000, 95] = 2.0
    m = sp.csr_matrix(dense)
    for impl in (0, 1):
        for skip in (False, True):
            _, cp = cases.formatted(m, impl, 4, 8, skip)
            ref = of.format_matrix(impl, 1024, 96, m.data, m.indices, m.indptr, 8, 4, skip)[0]
            for c in range(16):
                assert np.array_equal(cp.channel(c), ref[c])


def test_value_conversion_matches_oracle():
    rng = np.random.default_rng(5)
    v = np.concatenate([rng.uniform(-1, 300, 4000), [0.0, -0.0, 255.99999, 256.0, 1e-9, 2.0 ** -25, 3 * 2.0 ** -25, np.inf]]).astype(np.float32)
    assert np.array_equal(host.pack_vector(0, v), orc.pack_vector(0, v))
    assert np.array_equal(host.pack_vector(0, v), orc.q_from_float(v))
    assert np.array_equal(host.pack_vector(1, v), v.view(np.uint32))
    w = rng.integers(0, 2 ** 32, 5000, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(host.unpack_result(0, w), orc.unpack_result(0, w))


def test_formatter_rejects_bad_arguments():
    m = cases.random_csr(300, 40, 0.05, 1, 0)
    csr = host.CSRMatrix.from_scipy(m)
    with pytest.raises(host.HostError):
        host.format_matrix(csr, 2, vb_bank=4, ob_bank=3)      # float_stall: 128*ob must be a multiple of 1024
    with pytest.raises(host.HostError):
        host.format_matrix(csr, 7)                            # unknown impl
    with pytest.raises(host.HostError):
        host.CSRMatrix.from_arrays(2, 2, [0, 1, 2], [0, 5], [1.0, 1.0])   # column out of range


def test_generators_are_deterministic_and_sorted():
    a = host.CSRMatrix.generate("powerlaw", 3000, 5000, a=40000, b=0.4, c=1.0, seed=9).arrays()
    b = host.CSRMatrix.generate("powerlaw", 3000, 5000, a=40000, b=0.4, c=1.0, seed=9).arrays()
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    ip, ix, _ = a
    assert abs(int(ip[-1]) - 40000) < 2000
    for r in range(0, 3000, 97):   # strictly increasing columns inside a row
        row = ix[ip[r]:ip[r + 1]].astype(np.int64)
        assert (np.diff(row) > 0).all()
    # csim's generators (spmv_csim/csim.cpp:387-435)
    ip, ix, dv = host.CSRMatrix.generate("uniform", 1000, 1024, a=10).arrays()
    assert ip[-1] == 10000 and (dv == 1.0).all()
    assert np.array_equal(ix[:10], (102 * np.arange(10) + 0) % 1024) and np.array_equal(ix[10:20], (102 * np.arange(10) + 1) % 1024)


def test_normalize_by_outdegree():
    # util_normalize_csr_matrix_by_outdegree (sw/data_formatter.h:33-47): value = 1 / (non-zeros in the value's column)
    csr = host.CSRMatrix.generate("powerlaw", 500, 400, a=6000, b=0.3, c=1.0, seed=5)
    ip, ix, _ = csr.arrays()
    csr.normalize_by_outdegree()
    _, _, dv = csr.arrays()
    counts = np.bincount(ix, minlength=400)
    assert np.array_equal(dv, (1.0 / counts[ix]).astype(np.float32))


def test_rmat_generator_is_symmetric_and_deterministic():
    """SURVEY.md section 8d's stand-in recipe: symmetric R-MAT (a=.57 b=.19 c=.19); the same matrix for every thread count."""
    import os
    m = host.CSRMatrix.generate("rmat", 5000, 5000, a=60000, b=1.0, c=1.0, seed=42)
    ip, ix, dv = m.arrays()
    a = sp.csr_matrix((np.ones(len(ix), dtype=np.int8), ix.astype(np.int64), ip.astype(np.int64)), shape=(5000, 5000))
    assert (a != a.T).nnz == 0 and a.diagonal().sum() == 0
    assert 30000 < m.dims[2] <= 60000
    for r in range(5000):
        assert (np.diff(ix[ip[r]:ip[r + 1]].astype(np.int64)) > 0).all()
    deg = np.diff(ip.astype(np.int64))
    assert deg.max() > 20 * deg.mean()                      # heavy hubs: what makes it a harder stand-in than the Chung-Lu one
    old = os.environ.get("HISPARSE_FORMAT_THREADS")
    again = host.CSRMatrix.generate("rmat", 5000, 5000, a=60000, b=1.0, c=1.0, seed=42).arrays()
    assert np.array_equal(again[1], ix) and np.array_equal(again[2], dv)
    other = host.CSRMatrix.generate("rmat", 5000, 5000, a=60000, b=1.0, c=1.0, seed=43).arrays()
    assert not np.array_equal(other[0], ip)
    with pytest.raises(host.HostError):
        host.CSRMatrix.generate("rmat", 5000, 4000, a=60000, b=1.0, c=1.0, seed=42)
