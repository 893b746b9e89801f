"""SpMM extension (hs_spmm / hs_spmm_device): every column of Y = A X against the oracle's SpMV of that column
(bit-exact fixed point, 1e-4 float), host-pointer and device-pointer entry points, and the context's own vector / result untouched."""
import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases

pytestmark = pytest.mark.gpu


def _oracle(cp, impl, xw):
    return orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("k", [1, 3, 8])
def test_spmm_columns_match_spmv(impl, k):
    m = cases.random_csr(3000, 2500, 0.01, 17, impl)
    _, cp = cases.formatted(m, impl, 64, 8 if impl == 2 else 2, True)
    X = np.stack([host.pack_vector(impl, cases.random_x(cp.num_cols, 100 + j, impl)) for j in range(k)])
    own = host.pack_vector(impl, cases.random_x(cp.num_cols, 99, impl))
    with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
        eng.load_matrix(cp)
        eng.load_vector(own)
        eng.run()
        before = eng.read_result()
        Y = eng.spmm(X)
        assert np.array_equal(eng.read_result(), before)          # the context's own result buffer was not written
        eng.run()
        assert np.array_equal(eng.read_result(), before)          # nor its vector
    assert Y.shape == (k, cp.num_rows)
    for j in range(k):
        want = _oracle(cp, impl, X[j])
        assert np.array_equal(Y[j], want) if impl == 0 else cases.float_close(Y[j], want)


class _Hip:
    """hipMalloc / hipMemcpy through the HIP runtime libhisparse_hip.so is linked against (ctypes; a second runtime in the process --
    e.g. the one bundled with the torch wheel, when torch is imported AFTER the library -- would not see the GPU)."""

    def __init__(self):
        import ctypes as C
        device.lib()
        self.C, self.rt = C, C.CDLL("libamdhip64.so")
        self.rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.rt.hipFree.argtypes = [C.c_void_p]

    def upload(self, a):
        p = self.C.c_void_p()
        assert self.rt.hipMalloc(self.C.byref(p), a.nbytes) == 0
        assert self.rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p.value

    def download(self, ptr, shape):
        a = np.empty(shape, dtype=np.uint32)
        assert self.rt.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2) == 0
        return a

    def free(self, ptr):
        self.rt.hipFree(ptr)


def test_spmm_device_pointers_and_strides():
    hip = _Hip()
    impl, k = 0, 4
    m = cases.random_csr(5000, 4000, 0.004, 5, impl)
    _, cp = cases.formatted(m, impl, 4096, 8192, True)
    ldx, ldy = cp.num_cols + 12, cp.num_rows + 4                  # strides wider than the vectors
    X = np.zeros((k, ldx), dtype=np.uint32)
    for j in range(k):
        X[j, :cp.num_cols] = host.pack_vector(impl, cases.random_x(cp.num_cols, 7 + j, impl))
    xd = hip.upload(X)
    yd = hip.upload(np.full((k, ldy), 0xffffffff, dtype=np.uint32))
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        eng.spmm_device(xd, ldx, yd, ldy, k)
        eng.sync()
        Y = hip.download(yd, (k, ldy))
        for j in range(k):
            assert np.array_equal(Y[j, :cp.num_rows], _oracle(cp, impl, X[j, :cp.num_cols]))
            assert (Y[j, cp.num_rows:] == 0xffffffff).all()       # the padding between the columns is not written
        with pytest.raises(device.DeviceError):
            eng.spmm_device(xd, cp.num_cols - 4, yd, ldy, k)     # stride shorter than a vector
        with pytest.raises(device.DeviceError):
            eng.spmm_device(xd + 4, ldx, yd, ldy, k)             # misaligned
    hip.free(xd)
    hip.free(yd)
    with device.SpmvEngine(impl) as eng:
        with pytest.raises(device.DeviceError):
            eng.num_rows = 8
            eng.spmm(np.zeros((1, 8), dtype=np.uint32))           # no matrix loaded


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("shape", [(600, 4096, 0.4), (300, 9000, 0.2), (2, 2500, 0.9)])
def test_fused_spmm_over_a_bitmap_image(monkeypatch, impl, shape):
    """Dense rows -> BITMAP image -> hs_spmm takes 4 and 2 columns at a time through spmm_bitmap.hip (k = 7: 4 + 2 + one SpMV).  Every
    column must equal the SpMV kernel's answer for it BIT FOR BIT (same products, same summation order) and the oracle within the
    mode's contract."""
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    rows, cols, density = shape
    csr = host.CSRMatrix.generate("bernoulli", rows, cols, b=density, c=0.05 if impl else 1.0, seed=rows)
    if impl == 0:
        ip, ix, dv = csr.arrays()
        csr = host.CSRMatrix.from_arrays(rows, cols, ip, ix, np.abs(dv) * 0.01)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    k = 7
    X = np.stack([host.pack_vector(impl, cases.random_x(cp.num_cols, 200 + j, impl)) for j in range(k)])
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        Y = eng.spmm(X)
        singles = []
        for j in range(k):
            eng.load_vector(X[j])
            eng.run()
            singles.append(eng.read_result())
    if rows >= 600:                                   # (fewer rows than CUs -> column slices, float_stall's 1024-row padding on a short
        # matrix -> PAIRS: the fused kernel steps aside there, the results must still agree)
        assert device.STREAM_FORMATS[st["stream_format"]] == "bitmap" and st["col_slices"] == 1
    for j in range(k):
        # fixed point: bit for bit the SpMV kernel's answer (exact integer sums); float: the fused kernel adds a row's partial sums in
        # another order than the SpMV kernel may -- within the float tolerance (hisparse_hip.h)
        if impl == 0:
            assert np.array_equal(Y[j], singles[j]), f"column {j} differs from the SpMV kernel's result"
        else:
            assert cases.float_close(Y[j], singles[j], rtol=1e-5, atol=1e-5), f"column {j} differs from the SpMV kernel's result"
        want = _oracle(cp, impl, X[j])
        assert np.array_equal(Y[j], want) if impl == 0 else cases.float_close(Y[j], want)


def test_fused_spmm_transformer_50_full_size():
    """BASELINE config 3's matrix with a batch of 6 activations: 4 + 2 columns through the fused kernel, the SpMV kernel's answers
    (float: to 1e-5), within 1e-4 * max(1, |y|) of the oracle."""
    from hisparse_amd import datasets
    cfg, csr = datasets.load("transformer_50")
    impl = host.impl_id(cfg.impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
    rng = np.random.default_rng(50)
    X = np.stack([host.pack_vector(impl, rng.normal(size=cp.num_cols).astype(np.float32)) for _ in range(6)])
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        assert device.STREAM_FORMATS[eng.stats()["stream_format"]] == "bitmap"
        Y = eng.spmm(X)
        for j in range(6):
            eng.load_vector(X[j])
            eng.run()
            assert cases.float_close(Y[j], eng.read_result(), rtol=1e-5, atol=1e-5)
    for j in (0, 5):
        want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], X[j], cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                        cp.ob_bank, cp.vb_bank).view(np.float32).astype(np.float64)
        got = Y[j].view(np.float32).astype(np.float64)
        assert (np.abs(got - want) <= 1e-4 * np.maximum(1.0, np.abs(want))).all()


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("rows,cols,density,k", [(600, 4096, 0.4, 16), (300, 9000, 0.2, 35), (1030, 2500, 0.6, 16), (17, 3000, 0.5, 32), (600, 4096, 0.4, 8),
                                                 (300, 9000, 0.2, 21), (100, 5000, 0.3, 5)])
def test_spmm_on_the_matrix_engine(monkeypatch, impl, rows, cols, density, k):
    """Float BITMAP matrices, k >= 5: up to 16 columns at a time through spmm_mfma.hip (a pass of fewer vectors is filled up with zero vectors) (v_mfma_f32_16x16x4_f32 over the second image: rows in
    tiles of 16, x shared by the 16 rows of a tile, the matrix streamed once per 16 columns), the rest through the fused 4-column kernel
    and the SpMV kernel.  Every column against the oracle's SpMV of it (1e-4) and against the same call with the matrix engine switched
    off."""
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    csr = host.CSRMatrix.generate("bernoulli", rows, cols, b=density, c=0.05, seed=rows + k)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    X = np.stack([host.pack_vector(impl, cases.random_x(cp.num_cols, 300 + j, impl)) for j in range(k)])
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        Y = eng.spmm(X)
        monkeypatch.setenv("HISPARSE_SPMM_MFMA", "0")
        Y0 = eng.spmm(X)
        monkeypatch.delenv("HISPARSE_SPMM_MFMA")
        # a non-finite activation: inf / NaN must reach exactly the rows that hold the column, every other row stays what it was
        Xbad = X.copy()
        col = int(cp.num_cols // 3)
        Xbad[3, col] = np.array([np.inf], dtype=np.float32).view(np.uint32)[0]
        Ybad = eng.spmm(Xbad)
    if device.STREAM_FORMATS[st["stream_format"]] != "bitmap":
        pytest.skip("not a BITMAP image at this shape (float_stall's row padding): no second image")
    for j in range(k):
        want = _oracle(cp, impl, X[j])
        assert cases.float_close(Y[j], want), f"column {j}"
        assert cases.float_close(Y[j], Y0[j], rtol=1e-5, atol=1e-5), f"column {j} against the vector-ALU path"
    ip, ix, _ = csr.arrays()
    touching = np.zeros(cp.num_rows, dtype=bool)
    for r in range(rows):
        touching[r] = col in ix[ip[r]:ip[r + 1]]
    y3 = Ybad[3].view(np.float32)
    assert touching.any() and not np.isfinite(y3[touching]).any()
    assert np.isfinite(y3[~touching]).all() and cases.float_close(Ybad[3][~touching], Y[3][~touching], rtol=1e-5, atol=1e-5)
    for j in (0, k - 1):
        assert cases.float_close(Ybad[j], Y[j], rtol=1e-5, atol=1e-5)


# ---- round 5: four columns per pass over an element-stream (graph) image planned for it (spmm_sweep.hip; option spmm_vectors = 4) -------
@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("k", [2, 4, 7, 16])
@pytest.mark.parametrize("slices", ["", "3"])
def test_spmm_four_vectors_per_pass_over_a_sweep_image(impl, k, slices):
    """Every column of Y against the oracle's SpMV of that column (bit-exact in fixed point, tolerance in float); the same image still
    answers hs_run; strides wider than the vectors; the context's own vector / result untouched."""
    csr = host.CSRMatrix.generate("powerlaw", 60000, 90000, a=1500000, b=0.4, c=1.0 if impl == 0 else 2.0, seed=29)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(60000, 90000))
    v, o = host.default_banks(impl)
    _, cp = cases.formatted(m, impl, v, o, True)
    X = np.stack([host.pack_vector(impl, cases.random_x(cp.num_cols, 300 + j, impl)) for j in range(k)])
    own = host.pack_vector(impl, cases.random_x(cp.num_cols, 299, impl))
    with device.SpmvEngine(impl) as eng:
        eng.set_option("spmm_vectors", "4")
        if slices:
            eng.set_option("col_slices", slices)
        eng.load_matrix(cp)
        st = eng.stats()
        assert device.STREAM_FORMATS[st["stream_format"]] == "sweep"
        if slices:
            assert st["col_slices"] == int(slices)
        eng.load_vector(own)
        eng.run()
        before = eng.read_result()
        assert np.array_equal(before, _oracle(cp, impl, own)) if impl == 0 else cases.float_close(before, _oracle(cp, impl, own))
        Y = eng.spmm(X)
        assert np.array_equal(eng.read_result(), before)
        again = eng.spmm(X)
        eng.set_option("spmm_fused", "0")                          # the same columns as k SpMVs over the same image
        loop = eng.spmm(X)
    for j in range(k):
        want = _oracle(cp, impl, X[j])
        if impl == 0:
            assert np.array_equal(Y[j], want), (j, np.nonzero(Y[j] != want)[0][:8])
            assert np.array_equal(again[j], want) and np.array_equal(loop[j], want)
        else:
            assert cases.float_close(Y[j], want) and cases.float_close(again[j], want) and cases.float_close(loop[j], want)


def test_spmm_sweep_saturation_and_strided_device_buffers():
    """fixed point: a row whose sum passes 2^32 - 1 saturates per vector (the carry bit of ITS accumulator set); device pointers with
    leading dimensions wider than the vectors"""
    hip = _Hip()
    impl, k = 0, 5
    rows, cols = 20000, 30000
    rng = np.random.default_rng(3)
    import scipy.sparse as sp
    m = sp.random(rows, cols, density=0.002, random_state=np.random.RandomState(3), format="csr", dtype=np.float32)
    m.data = rng.uniform(0.0, 2.0, m.nnz).astype(np.float32)
    m = m.tolil()
    m[7, :3000] = 1.9                                            # 3000 x 1.9 x (x up to 2) >> 256: saturates for the columns with large x
    m = m.tocsr()
    m.sort_indices()
    _, cp = cases.formatted(m, impl, *host.default_banks(impl), True)
    ldx, ldy = cp.num_cols + 12, cp.num_rows + 4
    X = np.zeros((k, ldx), dtype=np.uint32)
    for j in range(k):
        xf = cases.random_x(cp.num_cols, 40 + j, impl)
        if j % 2:
            xf[:] = 0.0                                          # every other column: y = 0, no saturation anywhere
        X[j, :cp.num_cols] = host.pack_vector(impl, xf)
    xd = hip.upload(X)
    yd = hip.upload(np.full((k, ldy), 0xdeadbeef, dtype=np.uint32))
    try:
        with device.SpmvEngine(impl) as eng:
            eng.set_option("spmm_vectors", "4")
            eng.load_matrix(cp)
            eng.spmm_device(xd, ldx, yd, ldy, k)
            eng.sync()
            Y = hip.download(yd, (k, ldy))
    finally:
        hip.free(xd)
        hip.free(yd)
    assert (Y[:, cp.num_rows:] == 0xdeadbeef).all()               # nothing written beyond a column's rows
    for j in range(k):
        want = _oracle(cp, impl, np.ascontiguousarray(X[j, :cp.num_cols]))
        assert np.array_equal(Y[j, :cp.num_rows], want)
        assert (want[7] == 0xFFFFFFFF) == (j % 2 == 0)
