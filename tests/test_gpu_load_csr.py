"""hs_load_matrix_csr (load straight from CSR, the pre-processing on the GPU) against the drop-in path
CSR -> csr2cpsr -> channel packets -> hs_load_matrix: the SAME device image byte for byte, hence the same y."""
import numpy as np
import pytest

from hisparse_amd import datasets, device, host
from oracle import oracle as orc

import cases

pytestmark = pytest.mark.gpu


def _both(csr, impl, vb=0, ob=0, skip=True):
    kw = {}
    if vb:
        kw = dict(vb_bank=vb, ob_bank=ob)
    cp = host.format_matrix(csr, impl, skip_empty_rows=skip, **kw)
    with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as a, device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as b:
        a.load_matrix(cp)
        b.load_matrix_csr(csr)
        sa, sb = a.stats(), b.stats()
        assert (b.num_rows, b.num_cols, b.row_parts, b.col_parts) == (cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions)
        for key in ("nnz", "stream_bytes", "stream_elements", "num_blocks", "num_units", "num_workgroups", "lds_bytes", "col_slices", "ring_buffers", "stream_format"):
            assert sa[key] == sb[key], key
        ta, tb = a.read_tiles(), b.read_tiles()
        for part in ("blocks", "units", "image"):
            assert ta[part].tobytes() == tb[part].tobytes(), part + " differs between the CPSR and the CSR path"
        xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl))
        b.load_vector(xw)
        b.run()
        y = b.read_result()
    return cp, xw, y, sb


@pytest.mark.parametrize("fmt", ["pairs", "delta", "owner", "owner24", "bitmap", "sweep"])
@pytest.mark.parametrize("impl", [0, 1, 2])
def test_same_image_as_the_cpsr_path(monkeypatch, fmt, impl):
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", fmt)
    m = cases.random_csr(2500, 700, 0.02, 13, impl)
    csr = host.CSRMatrix.from_scipy(m)
    cp, xw, y, _ = _both(csr, impl, vb=16, ob=8 if impl == 2 else 2)
    want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)
    assert np.array_equal(y, want) if impl == 0 else cases.float_close(y, want)


@pytest.mark.parametrize("impl", [0, 1, 2])
def test_default_choice_shapes(monkeypatch, impl):
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    for kind, rows, cols, kw in [("powerlaw", 1000, 1000, dict(a=10000, b=0.0)), ("powerlaw", 70, 50000, dict(a=35000, b=0.3)),
                                 ("powerlaw", 300000, 64, dict(a=900000, b=0.3)), ("powerlaw", 40000, 40000, dict(a=16e6, b=0.3)),
                                 ("powerlaw", 60000, 90000, dict(a=100000, b=0.0)), ("bernoulli", 512, 8192, dict(b=0.4))]:
        csr = host.CSRMatrix.generate(kind, rows, cols, c=1.5, seed=rows, **kw)
        _both(csr, impl)


def test_unsorted_columns_and_explicit_zeros():
    rng = np.random.default_rng(5)
    m = cases.random_csr(4000, 3000, 0.01, 6, 0)
    ip, ix, dv = m.indptr.astype(np.uint32), m.indices.astype(np.uint32).copy(), m.data.copy()
    for r in range(m.shape[0]):
        a, b = ip[r], ip[r + 1]
        perm = rng.permutation(b - a)
        ix[a:b], dv[a:b] = ix[a:b][perm], dv[a:b][perm]
    dv[::17] = 0.0
    dv[5::23] = -1.0             # negative -> 0 in Q8.24 (sw/data_loader.h:76-84), still a stored element
    for impl in (0, 2):
        csr = host.CSRMatrix.from_arrays(4000, 3000, ip, ix, dv)
        _both(csr, impl)


def test_errors():
    ip = np.array([0, 2, 2, 3], dtype=np.uint32)
    with device.SpmvEngine(0) as eng:
        # the same (row, column) twice: both products count, like in the reference's formatter (formatted on the host then)
        eng.load_matrix_csr((3, 10, ip, np.array([4, 4, 1], dtype=np.uint32), np.array([1.0, 2.0, 4.0], dtype=np.float32)))
        assert not eng.stats()["retiled_on_gpu"]
        eng.load_vector(host.pack_vector(0, np.ones(eng.num_cols, dtype=np.float32)))
        eng.run()
        assert orc.unpack_result(0, eng.read_result())[:3].tolist() == [3.0, 0.0, 4.0]
        with pytest.raises(device.DeviceError, match="column"):
            eng.load_matrix_csr((3, 10, ip, np.array([4, 10, 1], dtype=np.uint32), np.ones(3, dtype=np.float32)))
        with pytest.raises(device.DeviceError, match="indptr"):
            eng.load_matrix_csr((3, 10, np.array([0, 2, 1, 3], dtype=np.uint32), np.array([4, 5, 1], dtype=np.uint32), np.ones(3, dtype=np.float32)))
        with pytest.raises(device.DeviceError, match="indptr"):
            eng.load_matrix_csr((3, 10, np.array([1, 2, 2, 3], dtype=np.uint32), np.array([4, 5, 1], dtype=np.uint32), np.ones(3, dtype=np.float32)))
        # an empty matrix of the right shape is fine
        eng.load_matrix_csr((3, 10, np.zeros(4, dtype=np.uint32), np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.float32)))
        eng.load_vector(np.zeros(eng.num_cols, dtype=np.uint32))
        eng.run()
        assert not eng.read_result().any()


@pytest.mark.parametrize("name", ["ogbl_ppa", "transformer_50", "ogbn_products", "mouse_gene", "ogbl_ppa_rmat"])
def test_full_size_configs(monkeypatch, name):
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    cfg, csr = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    _, _, _, st = _both(csr, impl, skip=cfg.skip_empty_rows)
    print(name, "CSR-path load %.1f ms" % (st["load_seconds"] * 1e3))
