"""The slice join (round 5; spmv_device.h: join_slices): in a column-sliced plan the last block of a row range to finish adds the partial
vectors up and writes y INSIDE the SpMV kernel -- one launch per SpMV instead of SpMV + combine_slices_kernel.  The sum is taken in slice
order exactly as the combine launch takes it, so both paths must give the same words bit for bit, in every numeric mode, in every kernel
that can run a sliced plan (row-block PAIRS / DELTA / OWNER24, SWEEP, BITMAP), on every launch (the arrival counters re-arm themselves),
through hs_run and through the reference's per-partition launch loop."""
import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases

pytestmark = pytest.mark.gpu


def _matrix(impl, rows=30000, cols=70000, nnz=900000, seed=17):
    csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=nnz, b=0.4, c=1.0 if impl == 0 else 2.0, seed=seed)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)
    import scipy.sparse as sp
    return sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(rows, cols))


def _engine(impl, cp, xw, fmt, slices, join):
    eng = device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank)
    eng.set_option("stream_format", fmt)
    eng.set_option("col_slices", str(slices))
    eng.set_option("light", "0")
    eng.set_option("slice_join", "1" if join else "0")
    eng.load_matrix(cp)
    eng.load_vector(xw)
    return eng


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("fmt", ["pairs", "delta", "owner24", "sweep", "bitmap"])
@pytest.mark.parametrize("slices", [2, 3, 5, 8])
def test_join_equals_combine_launch_bit_for_bit(impl, fmt, slices):
    if fmt == "bitmap":
        m = cases.random_csr(700, 9000, 0.2, 5, impl)
        v, o = host.default_banks(impl)
    else:
        m = _matrix(impl)
        v, o = host.default_banks(impl)
    csr, cp = cases.formatted(m, impl, v, o, True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl))
    want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    ys = {}
    for join in (False, True):
        with _engine(impl, cp, xw, fmt, slices, join) as eng:
            st = eng.stats()
            if st["col_slices"] == 1:
                pytest.skip(f"the {fmt} builder does not slice this matrix")
            assert st["slice_join"] == (1 if join else 0), st
            eng.run()
            ys[join] = eng.read_result()
            for _ in range(5):                       # the counters re-arm: every later launch gives the same words
                eng.run()
            assert np.array_equal(eng.read_result(), ys[join])
    assert np.array_equal(ys[True], ys[False]), np.nonzero(ys[True] != ys[False])[0][:8]
    if impl == 0:
        assert np.array_equal(ys[True], want)
    else:
        assert cases.float_close(ys[True], want)


@pytest.mark.parametrize("impl", [0, 2])
def test_join_through_the_partition_loop(impl):
    # several row partitions (small output banks): hs_run_partition runs the blocks of one partition only; rows of the others keep their words
    ob = 8 if impl == 2 else 1
    m = cases.random_csr(4000, 40000, 0.004, 9, impl)
    csr, cp = cases.formatted(m, impl, 4, ob, True)
    assert cp.num_row_partitions > 1
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 4, impl))
    with _engine(impl, cp, xw, "pairs", 4, True) as eng:
        assert eng.stats()["slice_join"] == 1
        eng.run()
        whole = eng.read_result()
        eng.load_vector(host.pack_vector(impl, np.zeros(cp.num_cols, dtype=np.float32)))
        eng.run()                                    # y = 0 everywhere
        assert not eng.read_result().any()
        eng.load_vector(xw)
        for j in range(cp.num_row_partitions):
            eng.run_partition(j, cp.part_len(j))
            eng.sync()
        assert np.array_equal(eng.read_result(), whole)


def test_join_many_launches_back_to_back():
    # 300 launches without a host synchronisation in between: a ticket of launch k must never be seen by launch k + 1
    impl = 0
    m = _matrix(impl, rows=60000, cols=90000, nnz=2500000, seed=23)
    csr, cp = cases.formatted(m, impl, *host.default_banks(impl), True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 8, impl))
    want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    for fmt in ("delta", "sweep"):
        with _engine(impl, cp, xw, fmt, 4, True) as eng:
            for _ in range(300):
                eng.run()
            assert np.array_equal(eng.read_result(), want)
            # alternate two vectors: a stale partial (the other vector's) anywhere in a sum would show
            xw2 = host.pack_vector(impl, cases.random_x(cp.num_cols, 9, impl))
            want2 = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw2, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
            for k in range(6):
                eng.load_vector(xw2 if k % 2 == 0 else xw)
                eng.run()
                assert np.array_equal(eng.read_result(), want2 if k % 2 == 0 else want)
