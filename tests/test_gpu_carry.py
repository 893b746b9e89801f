"""Column-sliced plans with carry_combine = 1 (round 5; hs_api.cpp: enqueue / flush_combine, spmv_device.h: CarriedCombine): when hs_run
follows hs_run the combine pass of the earlier step is done by the later step's kernel as its first act, and the stand-alone combine is
launched only when something else follows.  Nothing observable may change: each scenario runs with carry_combine = 0 and = 1 and must give
the same words, bit for bit (float modes too: the sums are taken in the same order)."""
import ctypes as C

import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases

pytestmark = pytest.mark.gpu


def _setup(impl, rows=30000, cols=70000, nnz=900000, vb=None, ob=None, seed=17):
    csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=nnz, b=0.4, c=1.0 if impl == 0 else 2.0, seed=seed)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(rows, cols))
    v, o = host.default_banks(impl)
    _, cp = cases.formatted(m, impl, vb or v, ob or o, True)
    return m, cp


def _engine(impl, cp, fmt, slices, carry):
    eng = device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank)
    eng.set_option("stream_format", fmt)
    eng.set_option("col_slices", str(slices))
    eng.set_option("light", "0")
    eng.set_option("carry_combine", "1" if carry else "0")
    eng.load_matrix(cp)
    return eng


def _oracle(impl, cp, xw):
    return orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("fmt,slices", [("pairs", 4), ("delta", 3), ("owner24", 5), ("sweep", 4), ("sweep", 7), ("pairs", 8)])
def test_consecutive_runs_with_changing_vectors(impl, fmt, slices):
    m, cp = _setup(impl)
    xs = [host.pack_vector(impl, cases.random_x(cp.num_cols, s, impl)) for s in (3, 4, 5)]
    wants = [_oracle(impl, cp, xw) for xw in xs]
    got = {}
    for carry in (False, True):
        with _engine(impl, cp, fmt, slices, carry) as eng:
            assert eng.stats()["col_slices"] == slices
            out = []
            for rounds in range(3):
                for k, xw in enumerate(xs):
                    eng.load_vector(xw)
                    for _ in range(1 + 7 * (rounds == 1)):
                        eng.run()                     # back to back: the kernel of step k + 1 writes y of step k
                    out.append(eng.read_result())     # the last step's sum is owed until here
            got[carry] = out
    for k, (a, b) in enumerate(zip(got[False], got[True])):
        assert np.array_equal(a, b), (k, np.nonzero(a != b)[0][:8])
        want = wants[k % 3]
        assert np.array_equal(b, want) if impl == 0 else cases.float_close(b, want)


def test_bitmap_slices_carry():
    impl = 1
    m = cases.random_csr(200, 9000, 0.2, 5, impl)       # fewer rows than CUs: the BITMAP builder slices the columns
    _, cp = cases.formatted(m, impl, *host.default_banks(impl), True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl))
    ys = {}
    for carry in (False, True):
        with _engine(impl, cp, "bitmap", 4, carry) as eng:
            if eng.stats()["col_slices"] == 1:
                pytest.skip("the BITMAP builder does not slice this matrix")
            eng.load_vector(xw)
            for _ in range(6):
                eng.run()
            ys[carry] = eng.read_result()
    assert np.array_equal(ys[True], ys[False]) and cases.float_close(ys[True], _oracle(impl, cp, xw))


def test_every_entry_point_settles_the_owed_sum():
    impl = 0
    m, cp = _setup(impl, ob=1, vb=4, rows=300000, cols=4000, nnz=1500000)      # three row partitions
    assert cp.num_row_partitions > 1
    x1 = host.pack_vector(impl, cases.random_x(cp.num_cols, 1, impl))
    x0 = host.pack_vector(impl, np.zeros(cp.num_cols, dtype=np.float32))
    want = _oracle(impl, cp, x1)
    rt = C.CDLL("libamdhip64.so")
    with _engine(impl, cp, "pairs", 3, True) as eng:
        eng.load_vector(x1)
        for _ in range(10):
            eng.run()
        eng.sync()
        assert np.array_equal(eng.read_result(), want)
        eng.run()                                                # ONE run, then a read: the combine is launched by the read
        assert np.array_equal(eng.read_result(), want)
        # hs_run_partition right behind carried hs_run calls
        eng.load_vector(x0)
        eng.run(); eng.run(); eng.run()
        eng.load_vector(x1)
        for j in range(cp.num_row_partitions):
            eng.run_partition(j, cp.part_len(j))
        assert np.array_equal(eng.read_result(), want)
        # the event-timed loops and the batch call (plain and graph replay)
        total, kern = eng.time_runs(3, 20)
        assert total > 0 and kern > 0
        assert eng.time_kernel(2, 10) > 0
        assert np.array_equal(eng.read_result(), want)
        for graph in ("0", "1"):
            eng.set_option("batch_graph", graph)
            eng.load_vector(x0); eng.run_batch(4)
            assert not eng.read_result().any()
            eng.load_vector(x1); eng.run_batch(4); eng.run_batch(4)
            assert np.array_equal(eng.read_result(), want)
        # SpMM as k SpMVs: every column through the carried path, each sum owed to ITS column of Y
        X = np.stack([x1, x0, x1])
        Y = eng.spmm(X)
        assert np.array_equal(Y[0], want) and not Y[1].any() and np.array_equal(Y[2], want)
        # iterate (feedback folded into the combine launch: never carried) after carried runs
        eng.run(); eng.run()
        one = host.pack_vector(impl, np.ones(1, dtype=np.float32))[0]
        eng.iterate(1, int(one), 0)
        assert np.array_equal(eng.read_result(), want)
        # a caller-owned stream: every step completes in itself, in stream order
        st = C.c_void_p()
        assert rt.hipStreamCreate(C.byref(st)) == 0
        eng.load_vector(x1)
        eng.run(); eng.run()
        eng.set_stream(st.value)
        for _ in range(5):
            eng.run()
        assert rt.hipStreamSynchronize(st) == 0
        y_dev = C.c_void_p()
        assert device.lib().hs_device_result(eng._h, C.byref(y_dev)) == 0
        y = np.empty(cp.num_rows, dtype=np.uint32)
        rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        assert rt.hipMemcpy(y.ctypes.data, y_dev, y.nbytes, 2) == 0      # no library call in between: y must be complete already
        assert np.array_equal(y, want)
        eng.set_stream(None)
        rt.hipStreamDestroy(st)
        # hs_get_stream hands the stream out: no more carrying, results unchanged
        eng.get_stream()
        for _ in range(5):
            eng.run()
        assert np.array_equal(eng.read_result(), want)


def test_bound_result_buffers_receive_their_own_step():
    impl = 0
    m, cp = _setup(impl)
    xa = host.pack_vector(impl, cases.random_x(cp.num_cols, 2, impl))
    xb = host.pack_vector(impl, cases.random_x(cp.num_cols, 7, impl))
    wa, wb = _oracle(impl, cp, xa), _oracle(impl, cp, xb)
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    bufs = [C.c_void_p(), C.c_void_p()]
    for b in bufs:
        assert rt.hipMalloc(C.byref(b), cp.num_rows * 4) == 0
    try:
        with _engine(impl, cp, "delta", 4, True) as eng:
            for k in range(6):                        # alternate vector and target: step k's sum belongs to ITS target
                eng.load_vector(xa if k % 2 == 0 else xb)
                eng.bind_device_result(bufs[k & 1].value)
                eng.run()
            eng.sync()
            for b, want in zip(bufs, (wa, wb)):
                y = np.empty(cp.num_rows, dtype=np.uint32)
                assert rt.hipMemcpy(y.ctypes.data, b, y.nbytes, 2) == 0
                assert np.array_equal(y, want)
            eng.bind_device_result(None)
    finally:
        for b in bufs:
            rt.hipFree(b)


def test_spmspv_auto_timing_block_leaves_nothing_owed():
    # spmspv = auto times the dense SpMV once, on a zero vector, with y swapped to the SpMSpV result buffer.  On a
    # carried column-sliced plan that left a sum owed to that buffer; when the rule then took the SPARSE path, the owed combine ran after
    # spmspv_pass and overwrote y with A*0.  The FIRST auto call must give the oracle's words, whichever path the rule takes.
    impl = 0
    rows, cols = 200000, 200000
    g = host.CSRMatrix.generate("powerlaw", rows, cols, a=20e6, b=0.3, c=1.0, seed=23)       # ~100 non-zeros per column: 8000 entries of x
    indptr, ridx, words = host.csr_to_csc(g, impl)                                              # -> ~800 K products (beyond the 30 us gate)
    rng = np.random.default_rng(4)
    xi = np.sort(rng.choice(cols, size=8000, replace=False)).astype(np.uint32)
    xw = host.pack_vector(impl, cases.random_x(len(xi), 4, impl))
    want = orc.spmspv(impl, indptr, ridx, words, rows, cols, xi, xw)
    assert want.any()
    for path in ("auto", "dense"):
        with device.SpmvEngine(impl) as eng:
            eng.set_option("stream_format", "pairs")
            eng.set_option("col_slices", "4")
            eng.set_option("carry_combine", "1")
            eng.set_option("spmspv", path)
            eng.load_matrix_csr(g)
            assert eng.stats()["col_slices"] == 4
            eng.load_matrix_csc(indptr, ridx, words, rows)
            first = eng.spmspv(xi, xw)
            second = eng.spmspv(xi, xw)
        assert np.array_equal(first, want), (path, int((first != want).sum()))
        assert np.array_equal(second, want), path


def test_spmm_device_and_in_place_vectors_on_a_carried_plan():
    # ADVICE round 5 (medium / low): hs_spmm_device's k-SpMV path must not leave the last column's sum owed to the CALLER's memory (a
    # device-wide synchronisation completes Y; the caller may free it), and x bound to the same memory as y never takes the carried path
    impl = 0
    m, cp = _setup(impl, rows=70016, cols=70016)       # 547 x 128: the padded row and column counts coincide
    xs = [host.pack_vector(impl, cases.random_x(cp.num_cols, s, impl)) for s in (11, 12, 13)]
    wants = [_oracle(impl, cp, xw) for xw in xs]
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    k, n = len(xs), cp.num_rows
    assert cp.num_cols == n and n % 4 == 0
    xd, yd = C.c_void_p(), C.c_void_p()
    assert rt.hipMalloc(C.byref(xd), k * n * 4 + 64) == 0 and rt.hipMalloc(C.byref(yd), k * n * 4) == 0
    try:
        with _engine(impl, cp, "pairs", 4, True) as eng:
            X = np.ascontiguousarray(np.stack(xs))
            assert rt.hipMemcpy(xd, X.ctypes.data, X.nbytes, 1) == 0
            lib = device.lib()
            assert lib.hs_spmm_device(eng._h, xd, C.c_uint64(n), yd, C.c_uint64(n), C.c_uint32(k)) == 0
            assert rt.hipDeviceSynchronize() == 0                    # NOT hs_sync: nothing may still be owed to yd
            Y = np.empty((k, n), dtype=np.uint32)
            assert rt.hipMemcpy(Y.ctypes.data, yd, Y.nbytes, 2) == 0
            for j in range(k):
                assert np.array_equal(Y[j], wants[j]), j
            # in place: y = A*x with x and y the same buffer, three times; the oracle iterates the same way
            eng.bind_device_vector(xd.value)
            eng.bind_device_result(xd.value)
            want = xs[0]
            for _ in range(3):
                eng.run()
                want = _oracle(impl, cp, want)
            eng.sync()
            y = np.empty(n, dtype=np.uint32)
            assert rt.hipMemcpy(y.ctypes.data, xd, y.nbytes, 2) == 0
            assert np.array_equal(y, want)
            eng.bind_device_vector(None)
            eng.bind_device_result(None)
    finally:
        rt.hipFree(xd)
        rt.hipFree(yd)


def test_batch_graph_is_dropped_when_bindings_or_options_change():
    # the cached hs_run_batch graph bakes in x, y and every option enqueue() reads
    impl = 0
    m, cp = _setup(impl)
    xa = host.pack_vector(impl, cases.random_x(cp.num_cols, 2, impl))
    want = _oracle(impl, cp, xa)
    with _engine(impl, cp, "delta", 4, True) as eng:
        eng.set_option("batch_graph", "1")
        eng.load_vector(xa)
        eng.run_batch(3)
        assert np.array_equal(eng.read_result(), want)
        eng.set_option("batch_graph", "1")          # any option change: re-captured, same result
        eng.run_batch(3)
        assert np.array_equal(eng.read_result(), want)
        eng.set_option("batch_graph", "0")
        eng.run_batch(3)
        assert np.array_equal(eng.read_result(), want)
