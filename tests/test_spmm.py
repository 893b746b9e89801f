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
