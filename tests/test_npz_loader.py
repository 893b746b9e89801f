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


def _save_raw(path, shape, data, indices, indptr):
    np.savez(path, shape=np.asarray(shape, dtype=np.int64), data=np.asarray(data, dtype=np.float32),
             indices=np.asarray(indices, dtype=np.int32), indptr=np.asarray(indptr, dtype=np.int32), format=np.array(b"csr"))


@pytest.mark.parametrize("what,shape,data,indices,indptr", [
    ("column index beyond num_cols", (2, 4), [1, 2, 3], [0, 9, 1], [0, 2, 3]),
    ("negative column index", (2, 4), [1, 2, 3], [0, -1, 1], [0, 2, 3]),
    ("indptr does not start at 0", (2, 4), [1, 2, 3], [0, 1, 1], [1, 2, 3]),
    ("indptr decreases", (3, 4), [1, 2, 3], [0, 1, 1], [0, 3, 2, 3]),
    ("indptr beyond nnz", (2, 4), [1, 2, 3], [0, 1, 1], [0, 7, 3]),
    ("indptr does not end at nnz", (2, 4), [1, 2, 3], [0, 1, 1], [0, 1, 2]),
    ("negative shape", (-2, 4), [1, 2, 3], [0, 1, 1], [0, 2, 3]),
])
def test_inconsistent_csr_arrays_are_refused(tmp_path, what, shape, data, indices, indptr):
    """A corrupt archive must fail in the loader, not corrupt the heap in csr2cpsr (which indexes tables by column/indptr)."""
    path = tmp_path / "bad.npz"
    _save_raw(path, shape, data, indices, indptr)
    with pytest.raises(host.HostError):
        host.load_csr_matrix_from_float_npz(str(path))


def test_truncated_and_bit_flipped_archives_never_crash(tmp_path):
    """Every prefix and a few hundred single-byte corruptions of a valid archive either load or raise HostError."""
    m = sp.random(40, 33, density=0.1, random_state=np.random.RandomState(3), format="csr", dtype=np.float32)
    good = tmp_path / "good.npz"
    sp.save_npz(good, m, compressed=False)
    blob = good.read_bytes()
    rng = np.random.default_rng(0)
    victim = tmp_path / "v.npz"
    cuts = sorted(set(rng.integers(0, len(blob), 120).tolist() + list(range(len(blob) - 80, len(blob)))))
    for cut in cuts:
        victim.write_bytes(blob[:cut])
        try:
            host.load_csr_matrix_from_float_npz(str(victim))
        except host.HostError:
            pass
    for _ in range(300):
        b = bytearray(blob)
        at = int(rng.integers(0, len(b)))
        b[at] = int(rng.integers(0, 256))
        victim.write_bytes(bytes(b))
        try:
            csr = host.load_csr_matrix_from_float_npz(str(victim))
            rows, cols, nnz = csr.dims
            ip, ix, _ = csr.arrays()
            assert ip[0] == 0 and ip[-1] == nnz and (nnz == 0 or ix.max() < cols)
        except host.HostError:
            pass
