"""SpMSpV extension: y = A x for a sparse x over a CSC matrix.  The reference only stubs the operator
(IDX_VAL_T / SPMSPV_MAT_PKT_T, spmv/libfpga/common.h:52-54; csr2csc, sw/data_loader.h:109-144), so the contract is derived:
the result equals the SpMV of the same matrix with x scattered into a zero vector -- bit for bit in fixed point (order free),
within the float tolerance otherwise.  CPU part: the CSC conversion and the oracle restatement against the SpMV oracle;
GPU part: hs_load_matrix_csc / hs_spmspv against both."""
import numpy as np
import pytest
import scipy.sparse as sp

from hisparse_amd import device, host
from oracle import oracle as orc

import cases


def _case(impl, rows, cols, density, seed, x_nnz):
    if rows * cols > 2e8:      # scipy.sparse.random is far too slow at this size: the host library's seeded generator instead
        g = host.CSRMatrix.generate("powerlaw", rows, cols, a=density * rows * cols, b=0.3, c=1.0 if impl == 0 else 2.0, seed=seed)
        ip, ix, dv = g.arrays()
        m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(rows, cols))
    else:
        m = cases.random_csr(rows, cols, density, seed, impl)
    csr = host.CSRMatrix.from_scipy(m)
    indptr, ridx, words = host.csr_to_csc(csr, impl)
    rng = np.random.default_rng(seed)
    xi = np.sort(rng.choice(cols, size=min(x_nnz, cols), replace=False)).astype(np.uint32)
    xv = cases.random_x(len(xi), seed, impl)
    xw = host.pack_vector(impl, xv)
    return m, csr, (indptr, ridx, words), xi, xv, xw


def _dense_spmv_oracle(m, impl, xi, xv):
    """The SpMV oracle (csim restatement) on the same matrix with x densified."""
    csr = host.CSRMatrix.from_scipy(m)
    v, o = host.default_banks(impl)
    cp = host.format_matrix(csr, impl, vb_bank=v, ob_bank=o, skip_empty_rows=True)
    x = np.zeros(cp.num_cols, dtype=np.float32)
    x[xi] = xv
    y = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], host.pack_vector(impl, x), cp.num_rows, cp.num_cols,
                 cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    return y[:m.shape[0]]


@pytest.mark.parametrize("impl", [0, 1])
def test_csc_conversion_and_oracle_against_the_spmv_oracle(impl):
    m, csr, (indptr, ridx, words), xi, xv, xw = _case(impl, 700, 500, 0.03, 5, 60)
    ref = m.tocsc()
    ref.sort_indices()
    assert np.array_equal(indptr, ref.indptr) and np.array_equal(ridx, ref.indices)       # rows ascending inside a column
    assert np.array_equal(words, host.pack_vector(impl, ref.data))
    y = orc.spmspv(impl, indptr, ridx, words, 700, 500, xi, xw)
    want = _dense_spmv_oracle(m, impl, xi, xv)
    if impl == 0:
        assert np.array_equal(y, want)
    else:
        assert cases.float_close(y, want)
    assert not orc.spmspv(impl, indptr, ridx, words, 700, 500, xi[:0], xw[:0]).any()      # empty x -> zero y
    with pytest.raises(orc.OracleError):
        orc.spmspv(impl, indptr, ridx, words, 700, 500, np.array([500], dtype=np.uint32), xw[:1])


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["sparse", "dense", "auto"])
@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("rows,cols,density,x_nnz", [(700, 500, 0.03, 60), (40000, 30000, 0.002, 3000), (3000, 9000, 0.05, 1), (2000, 2000, 0.2, 2000),
                                                     (400000, 60000, 3.3e-4, 30000)])
def test_device_spmspv_matches_oracle_and_dense_spmv(impl, rows, cols, density, x_nnz, path, monkeypatch):
    # "sparse": the two-launch path (expand into the product list, one workgroup per row block accumulates its own) whatever the size;
    # "dense": the dispatch to the dense SpMV of the same matrix loaded on the same context; "auto": the library's choice by the crossover
    monkeypatch.setenv("HISPARSE_SPMSPV", path)      # (the dense dispatch is opt-in: "auto" declares that both loads hold the same matrix)
    m, csr, (indptr, ridx, words), xi, xv, xw = _case(impl, rows, cols, density, 9, x_nnz)
    want = orc.spmspv(impl, indptr, ridx, words, rows, cols, xi, xw)
    with device.SpmvEngine(impl) as eng:
        if path != "sparse":
            eng.load_matrix_csr(csr)                   # the same matrix for the dense dispatch
        eng.load_matrix_csc(indptr, ridx, words, rows)
        got = eng.spmspv(xi, xw)
        again = eng.spmspv(xi, xw)                 # accumulators are re-armed every call
        empty = eng.spmspv(xi[:0], xw[:0])
        with pytest.raises(device.DeviceError):
            eng.spmspv(np.array([cols], dtype=np.uint32), xw[:1])
    assert not empty.any()
    if impl == 0:
        assert np.array_equal(got, want) and np.array_equal(again, want)
        if rows * cols <= 2e8:
            assert np.array_equal(got, _dense_spmv_oracle(m, impl, xi, xv))
    else:
        assert cases.float_close(got, want) and cases.float_close(again, want)


@pytest.mark.gpu
@pytest.mark.parametrize("impl", [0, 1])
def test_csc_matrix_is_independent_of_the_dense_matrix_by_default(impl, monkeypatch):
    # A for SpMV and a DIFFERENT matrix of the same shape as CSC on one context -- a large x (every column: far beyond
    # any crossover) must still be multiplied by the CSC matrix; only spmspv = auto | dense (the caller's declaration that the two are the
    # same matrix) may answer with the dense SpMV
    monkeypatch.delenv("HISPARSE_SPMSPV", raising=False)
    monkeypatch.delenv("HISPARSE_SPMSPV_CROSSOVER", raising=False)
    rows, cols = 40000, 30000
    m_a, csr_a, _, _, _, _ = _case(impl, rows, cols, 0.004, 5, 10)
    m_b, csr_b, (indptr, ridx, words), xi, xv, xw = _case(impl, rows, cols, 0.004, 6, cols)       # x names every column: 4.8 M products
    want_b = orc.spmspv(impl, indptr, ridx, words, rows, cols, xi, xw)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix_csr(csr_a)                     # same shape, other matrix
        eng.load_matrix_csc(indptr, ridx, words, rows)
        got = eng.spmspv(xi, xw)
        assert np.array_equal(got, want_b) if impl == 0 else cases.float_close(got, want_b)
        eng.set_option("spmspv", "dense")              # the declaration (here: a false one) switches to the dense matrix
        other = eng.spmspv(xi, xw)
        assert not np.array_equal(other, got)


@pytest.mark.gpu
@pytest.mark.parametrize("impl", [0, 1])
def test_repeated_entries_add_up_and_oversized_calls_are_split(impl):
    # a column named several times is legal (its products simply add up) and makes the product count exceed the matrix's
    # non-zeros -- the list's capacity.  hs_spmspv cuts such a call into passes (y =, then y +=); every column four times here.
    rows, cols = 30000, 2000
    m, csr, (indptr, ridx, words), xi, xv, xw = _case(impl, rows, cols, 0.01, 4, cols)
    xi4 = np.concatenate([xi, xi[::-1], xi, xi]).astype(np.uint32)
    rng = np.random.default_rng(1)
    xw4 = host.pack_vector(impl, cases.random_x(len(xi4), 6, impl) * (0.25 if impl == 0 else 1.0))
    want = orc.spmspv(impl, indptr, ridx, words, rows, cols, xi4, xw4)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix_csr(csr)                       # a dense matrix is there, but repeats rule the dense dispatch out
        eng.load_matrix_csc(indptr, ridx, words, rows)
        got = eng.spmspv(xi4, xw4)
        few = eng.spmspv(xi4[:3], xw4[:3])             # and a small call afterwards starts from a clean list
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
    want_few = orc.spmspv(impl, indptr, ridx, words, rows, cols, xi4[:3], xw4[:3])
    assert np.array_equal(few, want_few) if impl == 0 else cases.float_close(few, want_few)


@pytest.mark.gpu
@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("rows", [20_000_000, 40_000_000, 120_000_000])
def test_matrices_taller_than_2048_row_blocks(impl, rows):
    # more than 2048 row blocks of 8192 rows used to be refused.  Above 16.7 M rows a block is 16 384 rows
    # (20 M: 1221 bins), above 33.5 M the expand kernel counts more than 2048 bins (40 M: 2442), above 100 M in more than 48 KiB of LDS
    # (120 M: 7325 bins).  A tall thin CSC matrix written out by hand: rows spread over the whole height, first and last row included.
    cols, per_col = 300, 400
    rng = np.random.default_rng(rows % 1000 + impl)
    stride = rows // per_col                              # one row per stratum: ascending and distinct inside a column
    ridx = np.arange(per_col, dtype=np.int64)[None, :] * stride + rng.integers(0, stride, size=(cols, per_col), dtype=np.int64)
    ridx[0, 0], ridx[-1, -1] = 0, rows - 1
    indptr = (np.arange(cols + 1) * per_col).astype(np.uint32)
    vals = cases.random_x(cols * per_col, 9, impl)
    words = host.pack_vector(impl, vals)
    ridx = ridx.reshape(-1).astype(np.uint32)
    xi = np.arange(0, cols, 2, dtype=np.uint32)
    xw = host.pack_vector(impl, cases.random_x(len(xi), 3, impl))
    want = orc.spmspv(impl, indptr, ridx, words, rows, cols, xi, xw)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix_csc(indptr, ridx, words, rows)
        got = eng.spmspv(xi, xw)
        again = eng.spmspv(xi, xw)                       # cursors re-armed
    assert got.shape == (rows,) and np.count_nonzero(want) > 0
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
    assert np.array_equal(again, got)


@pytest.mark.gpu
def test_device_resident_entries_and_the_overflow_report():
    import ctypes as C
    impl, rows, cols = 0, 50000, 40000
    m, csr, (indptr, ridx, words), xi, xv, xw = _case(impl, rows, cols, 0.001, 12, 500)
    pairs = np.empty((len(xi), 2), dtype=np.uint32)
    pairs[:, 0], pairs[:, 1] = xi, xw
    want = orc.spmspv(impl, indptr, ridx, words, rows, cols, xi, xw)
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix_csc(indptr, ridx, words, rows)
        every = np.empty((cols * 3, 2), dtype=np.uint32)          # every column three times: 3 x nnz products, more than the list holds
        every[:, 0] = np.tile(np.arange(cols, dtype=np.uint32), 3)
        every[:, 1] = host.pack_vector(impl, np.full(cols * 3, 0.001, dtype=np.float32))
        d = C.c_void_p()
        assert rt.hipMalloc(C.byref(d), every.nbytes) == 0
        try:
            assert rt.hipMemcpy(d, pairs.ctypes.data, pairs.nbytes, 1) == 0
            for _ in range(3):                                     # nothing but the two launches per call; the counters re-arm themselves
                eng.spmspv_device(d.value, len(xi))
            assert np.array_equal(eng.read_spmspv_result(), want)
            assert rt.hipMemcpy(d, every.ctypes.data, every.nbytes, 1) == 0
            eng.spmspv_device(d.value, cols * 3)
            # a consumer on the device never reads y back through the library: hs_spmspv_status tells
            flag, word = C.c_uint32(7), C.c_void_p()
            assert device.lib().hs_spmspv_status(eng._h, C.byref(flag), C.byref(word)) == 0 and flag.value == 1 and word.value
            seen = C.c_uint32(0)
            assert rt.hipMemcpy(C.byref(seen), word, 4, 2) == 0 and seen.value == 1
            with pytest.raises(device.DeviceError) as e:
                eng.read_spmspv_result()
            assert "more products" in str(e.value)
            assert device.lib().hs_spmspv_status(eng._h, C.byref(flag), None) == 0 and flag.value == 0
            assert rt.hipMemcpy(d, pairs.ctypes.data, pairs.nbytes, 1) == 0
            eng.spmspv_device(d.value, len(xi))                    # the flag is cleared, the next call is fine
            assert np.array_equal(eng.read_spmspv_result(), want)
        finally:
            rt.hipFree(d)


@pytest.mark.gpu
def test_saturation_and_errors_on_device():
    # one row collecting 600 products of 1.0 * 1.0 saturates at 256 - 2^-24 (AP_SAT), exactly like the dense path
    rows, cols = 128, 640
    m = sp.csr_matrix((np.ones(600, dtype=np.float32), (np.full(600, 7), np.arange(600))), shape=(rows, cols))
    csr = host.CSRMatrix.from_scipy(m)
    indptr, ridx, words = host.csr_to_csc(csr, 0)
    xi = np.arange(600, dtype=np.uint32)
    xw = host.pack_vector(0, np.ones(600, dtype=np.float32))
    with device.SpmvEngine(0) as eng:
        with pytest.raises(device.DeviceError):
            eng.csc_rows = rows
            eng.spmspv(xi, xw)                     # no CSC matrix loaded yet
        eng.load_matrix_csc(indptr, ridx, words, rows)
        y = eng.spmspv(xi, xw)
        bad = indptr.copy()
        bad[3] = 5000
        with pytest.raises(device.DeviceError):
            eng.load_matrix_csc(bad, ridx, words, rows)
    assert y[7] == 0xFFFFFFFF and not np.delete(y, 7).any()
    assert np.array_equal(y, orc.spmspv(0, indptr, ridx, words, rows, cols, xi, xw))
