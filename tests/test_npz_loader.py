"""load_csr_matrix_from_float_npz: scipy.sparse.save_npz archives (the reference's dataset format, sw/data_loader.h:51-70)
through the from-scratch zip/npy reader in include/hisparse/npz.h."""
import numpy as np
import pytest
import scipy.sparse as sp

from hisparse_amd import host


@pytest.mark.parametrize("compressed", [True, False])
@pytest.mark.parametrize("index_dtype", [np.int32, np.int64])
def test_scipy_npz_roundtrip(tmp_path, compressed, index_dtype):
    m = sp.random(500, 321, density=0.03, random_state=np.random.RandomState(1), format="csr", dtype=np.float32)
    m.indices = m.indices.astype(index_dtype)
    m.indptr = m.indptr.astype(index_dtype)
    path = tmp_path / "m.npz"
    sp.save_npz(path, m, compressed=compressed)
    csr = host.load_csr_matrix_from_float_npz(str(path))
    assert csr.dims == (500, 321, m.nnz)
    ip, ix, dv = csr.arrays()
    assert np.array_equal(ip, m.indptr) and np.array_equal(ix, m.indices) and np.array_equal(dv, m.data)


def test_missing_and_corrupt_files(tmp_path):
    with pytest.raises(host.HostError):
        host.load_csr_matrix_from_float_npz(str(tmp_path / "nope.npz"))
    bad = tmp_path / "bad.npz"
    bad.write_bytes(b"PK\x03\x04 this is not a zip archive")
    with pytest.raises(host.HostError):
        host.load_csr_matrix_from_float_npz(str(bad))
    np.savez(tmp_path / "partial.npz", data=np.zeros(3, dtype=np.float32))
    with pytest.raises(host.HostError):
        host.load_csr_matrix_from_float_npz(str(tmp_path / "partial.npz"))


def test_float64_data_is_narrowed(tmp_path):
    m = sp.random(64, 64, density=0.1, random_state=np.random.RandomState(2), format="csr", dtype=np.float64)
    sp.save_npz(tmp_path / "d.npz", m)
    _, _, dv = host.load_csr_matrix_from_float_npz(str(tmp_path / "d.npz")).arrays()
    assert np.array_equal(dv, m.data.astype(np.float32))
