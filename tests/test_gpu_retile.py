"""The load-time re-tiling on the GPU (hisparse_amd/csrc/gpu_tiles.hip) against the host builder (stream_tiles.cpp), byte for byte.

hs_load_matrix runs the per-non-zero passes on the device by default; hs_tiles_build is the host builder with the same planner.
Both must leave the SAME image, Block[] and Unit[] -- the SpMV parity tests then cover whichever built the image, and this file
pins that the two are interchangeable.  (the reference formats on the host only, sw/data_formatter.h.)
"""
import numpy as np
import pytest

from hisparse_amd import datasets, device, host

import cases

pytestmark = pytest.mark.gpu

FORMATS = ["pairs", "delta", "owner", "pairs24", "owner24", "sweep"]


def _set_format(monkeypatch, fmt):
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "pairs" if fmt == "pairs24" else fmt)
    if fmt == "pairs24":
        monkeypatch.setenv("HISPARSE_AUX_BITS", "24")
    monkeypatch.delenv("HISPARSE_RETILE", raising=False)


def _compare(cp, impl, expect_gpu=True):
    with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        got = eng.read_tiles()
    want = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                              st["num_compute_units"])
    assert bool(st["retiled_on_gpu"]) == expect_gpu
    assert device.STREAM_FORMATS[st["stream_format"]] == want["format"]
    assert st["stream_bytes"] == want["image"].size and st["num_blocks"] == want["blocks"].size and st["num_units"] == want["units"].size
    assert st["stream_elements"] == want["elements"] and st["nnz"] == want["nnz"]
    assert got["blocks"].tobytes() == want["blocks"].tobytes(), "Block[] differs"
    assert got["units"].tobytes() == want["units"].tobytes(), "Unit[] differs"
    if not np.array_equal(got["image"], want["image"]):
        bad = np.nonzero(got["image"] != want["image"])[0]
        raise AssertionError(f"image differs at {bad.size} of {want['image'].size} bytes, first at {bad[:8]}")
    return st


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("skip", [False, True])
def test_small_banks_many_partitions(monkeypatch, fmt, impl, skip):
    _set_format(monkeypatch, fmt)
    m = cases.random_csr(2500, 300, 0.03, 11, impl)
    _, cp = cases.formatted(m, impl, 4, 8 if impl == 2 else 1, skip)
    _compare(cp, impl)


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("impl", [0, 1, 2])
def test_default_banks(monkeypatch, fmt, impl):
    _set_format(monkeypatch, fmt)
    # (the host generator: scipy.sparse.random takes 10 s and more at these shapes)
    for rows, cols, density, seed in [(1000, 1000, 0.01, 1), (40000, 9000, 0.002, 5), (70, 50000, 0.01, 7), (300000, 64, 0.05, 9)]:
        csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=rows * cols * density, b=0.3, c=1.5, seed=seed)
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
        _compare(cp, impl)


@pytest.mark.parametrize("impl", [0, 1, 2])
def test_empty_and_ragged(monkeypatch, impl):
    import scipy.sparse as sp
    monkeypatch.delenv("HISPARSE_RETILE", raising=False)
    for shape in [(64, 64), (1, 1), (129, 7)]:
        m = sp.csr_matrix(shape, dtype=np.float32)
        csr = host.CSRMatrix.from_scipy(m)
        for skip in (False, True):
            cp = host.format_matrix(csr, impl, skip_empty_rows=skip)
            _compare(cp, impl)
    # rows far apart (chained markers with skip_empty_rows), one very long row
    rng = np.random.default_rng(3)
    rows = np.concatenate([np.array([0, 5000, 5001, 99999]), np.full(3000, 70000)])
    cols = np.concatenate([np.array([1, 2, 3, 4]), rng.choice(20000, 3000, replace=False)])
    vals = rng.uniform(0.1, 1.0, rows.size).astype(np.float32)
    m = sp.csr_matrix((vals, (rows, cols)), shape=(100000, 20000))
    m.sort_indices()
    csr = host.CSRMatrix.from_scipy(m)
    for skip in (False, True):
        cp = host.format_matrix(csr, impl, skip_empty_rows=skip)
        _compare(cp, impl)


def test_default_format_choice(monkeypatch):
    # no forced format: the planner decides (PAIRS / DELTA / OWNER on the GPU; BITMAP images are built by the host code)
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    monkeypatch.delenv("HISPARSE_RETILE", raising=False)
    for impl in (0, 1, 2):
        csr = host.CSRMatrix.generate("powerlaw", 30000, 30000, a=900000, b=0.2, c=1.5, seed=21)
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
        _compare(cp, impl)
    csr = host.CSRMatrix.generate("bernoulli", 512, 8192, b=0.5, c=1.0, seed=22)
    cp = host.format_matrix(csr, 2, skip_empty_rows=True)
    st = _compare(cp, 2)
    assert device.STREAM_FORMATS[st["stream_format"]] == "bitmap"


def _mfma_image(cp, impl, where):
    import os
    old = os.environ.pop("HISPARSE_BITMAP_BUILD", None)
    if where:
        os.environ["HISPARSE_BITMAP_BUILD"] = where
    try:
        with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
            eng.load_matrix(cp)
            return eng.read_mfma_image(), eng.stats()
    finally:
        os.environ.pop("HISPARSE_BITMAP_BUILD", None)
        if old is not None:
            os.environ["HISPARSE_BITMAP_BUILD"] = old


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("shape", [(512, 8192, 0.5), (3, 40000, 0.3), (16, 2048, 0.6), (9, 2500, 0.2), (130, 4096, 0.15), (2500, 2100, 0.25),
                                   (20000, 2048, 0.13), (1, 70000, 0.9), (700, 3000, 0.02)])
def test_bitmap_images_built_on_the_device(monkeypatch, impl, shape):
    """BITMAP: masks, compacted values, run heads (and, float modes, the matrix-engine image) come from the kernels of gpu_tiles.hip and
    must be byte for byte what the host builder writes: whole-row and partial-row runs, column slices (fewer rows than compute units),
    rows too sparse for the format (forced), one row, float_stall's row padding."""
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bitmap")
    monkeypatch.delenv("HISPARSE_RETILE", raising=False)
    rows, cols, density = shape
    csr = host.CSRMatrix.generate("bernoulli", rows, cols, b=density, c=1.0, seed=rows + cols)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    st = _compare(cp, impl)
    assert device.STREAM_FORMATS[st["stream_format"]] == "bitmap"
    got, _ = _mfma_image(cp, impl, None)
    want, st_host = _mfma_image(cp, impl, "host")
    assert not st_host["retiled_on_gpu"]
    assert (got.size > 0) == (impl != 0) and got.size == want.size
    assert got.tobytes() == want.tobytes(), "matrix-engine image differs"


@pytest.mark.parametrize("slices", [2, 3, 8])
def test_bitmap_column_slices_on_the_device(monkeypatch, slices):
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bitmap")
    monkeypatch.setenv("HISPARSE_COL_SLICES", str(slices))
    csr = host.CSRMatrix.generate("bernoulli", 40, 9000, b=0.4, c=1.0, seed=5)
    for impl in (0, 1):
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
        st = _compare(cp, impl)
        assert st["col_slices"] == slices


def test_bitmap_duplicate_column_falls_back(monkeypatch):
    # a column twice in a row cannot be a bit in a mask: the device builder notices (the bit is set already) and the element formats take over
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bitmap")
    m = cases.random_csr(64, 4096, 0.3, 3, 0)
    ip, ix, dv = m.indptr.astype(np.uint32), m.indices.astype(np.uint32).copy(), m.data.copy()
    ix[ip[5] + 1] = ix[ip[5]]
    csr = host.CSRMatrix.from_arrays(64, 4096, ip, ix, dv)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    st = _compare(cp, 0, expect_gpu=False)
    assert device.STREAM_FORMATS[st["stream_format"]] != "bitmap"


@pytest.mark.parametrize("impl", [0, 1])
def test_bitmap_duplicate_in_a_many_block_matrix_leaves_no_stale_tables(monkeypatch, impl):
    # the device BITMAP builder sees a duplicate only in its mask pass, after blocks / units / max_block_rows / col_slices
    # were laid out; the element-format path that takes over must start from an empty StreamTiles (it push_backs onto the tables and sizes
    # its sort keys by them).  A dense matrix of several thousand rows = hundreds of BITMAP blocks and runs, duplicates in many rows;
    # image and tables must equal the host builder's, and y the oracle's.
    from oracle import oracle as orc
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    monkeypatch.delenv("HISPARSE_RETILE", raising=False)
    m = cases.random_csr(3000, 4096, 0.2, 21, impl)
    ip, ix, dv = m.indptr.astype(np.uint32), m.indices.astype(np.uint32).copy(), m.data.copy()
    for r in range(3, 3000, 97):
        if ip[r + 1] - ip[r] >= 2:
            ix[ip[r] + 1] = ix[ip[r]]
    csr = host.CSRMatrix.from_arrays(3000, 4096, ip, ix, dv)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    st = _compare(cp, impl, expect_gpu=False)
    assert device.STREAM_FORMATS[st["stream_format"]] != "bitmap"
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 9, impl))
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        eng.load_vector(xw)
        eng.run()
        got = eng.read_result()
    want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)


def test_host_opt_out(monkeypatch):
    monkeypatch.setenv("HISPARSE_RETILE", "host")
    m = cases.random_csr(3000, 3000, 0.01, 4, 0)
    cp = host.format_matrix(host.CSRMatrix.from_scipy(m), 0, skip_empty_rows=True)
    _compare(cp, 0, expect_gpu=False)


@pytest.mark.parametrize("name", ["ogbl_ppa", "ogbn_products", "mouse_gene", "ogbl_ppa_rmat"])
def test_full_size_configs(monkeypatch, name):
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    monkeypatch.delenv("HISPARSE_RETILE", raising=False)
    cfg, csr = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
    _compare(cp, impl)


@pytest.mark.parametrize("impl", [0, 2])
def test_duplicate_entries_fall_back_to_host(monkeypatch, impl):
    # the same (row, column) twice: legal input for the reference's formatter; the host builder orders equal positions by value
    # word, the device sort does not, so hs_load_matrix hands such a matrix to the host builder
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    monkeypatch.delenv("HISPARSE_RETILE", raising=False)
    m = cases.random_csr(5000, 5000, 0.004, 8, impl)
    ip, ix, dv = m.indptr.astype(np.uint32), m.indices.astype(np.uint32).copy(), m.data.copy()
    for r in range(0, 5000, 7):
        if ip[r + 1] - ip[r] >= 2:
            ix[ip[r] + 1] = ix[ip[r]]
    csr = host.CSRMatrix.from_arrays(5000, 5000, ip, ix, dv)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    _compare(cp, impl, expect_gpu=False)
