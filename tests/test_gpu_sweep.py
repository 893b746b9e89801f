"""The SWEEP format (round 4; stream_tiles.h, sweep_tiles.cpp, spmv_sweep.hip): column-ordered blocks, x gathered from L2, no units.  Chosen
automatically for very sparse matrices (where its plan is modelled faster than OWNER24's: a mean position gap of ~60-70 K and up on square
matrices, lower on row slabs; > 2 M non-zeros); here: the automatic choice at that
scale against the oracle, the reference's partition loop and chains of blocks per workgroup, fixed-point saturation through the 4-byte sums
with a carry bit, non-finite x, iterate / SpMM on top of the image, and the image byte for byte against the CPU build of the same library.
(tests/test_gpu_parity.py runs its whole case list through the format as "sweep"; tests/test_sweep_cpu.py checks the image on the CPU.)"""
import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases

pytestmark = pytest.mark.gpu


def _oracle(cp, impl, xw):
    return orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)


def _check(impl, got, want):
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)


@pytest.fixture(autouse=True)
def _clean(monkeypatch):
    for k in ("HISPARSE_STREAM_FORMAT", "HISPARSE_SWEEP", "HISPARSE_LIGHT", "HISPARSE_MAX_ROWS", "HISPARSE_COL_SLICES", "HISPARSE_XCD_AFFINITY"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("impl", [0, 1, 2])
def test_very_sparse_matrices_take_sweep_and_match_the_oracle(impl):
    # 1 M x 1 M, 5 M non-zeros (mean position gap 200 K): SWEEP by the planner's own rule -- its plan is modelled (and measured: 31 against 39 us)
    # faster than OWNER24's -- in every numeric mode
    csr = host.CSRMatrix.generate("powerlaw", 1000000, 1000000, a=5.0e6, b=0.4, c=1.0 if impl == 0 else 2.0, seed=21 + impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl) * (30.0 if impl == 0 else 1.0))      # fixed point: hub rows saturate
    want = _oracle(cp, impl, xw)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        assert device.STREAM_FORMATS[st["stream_format"]] == "sweep", st
        assert st["lds_bytes"] <= 160 * 1024 and st["num_units"] == 1
        eng.load_vector(xw)
        eng.run()
        _check(impl, eng.read_result(), want)
        eng.run()                                                # accumulators (and carry bits) are re-armed by every launch
        _check(impl, eng.read_result(), want)
        tiles = eng.read_tiles()
    if impl == 0:
        assert (want == 0xFFFFFFFF).any() and (want != 0xFFFFFFFF).any()
    built = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, st["num_compute_units"])
    assert built["image"].tobytes() == tiles["image"].tobytes() and built["blocks"].tobytes() == tiles["blocks"].tobytes()
    # gap 30 K: OWNER24 until round 5; with the SWEEP model fitted again on the round-5 kernel (sweep_tiles.cpp) SWEEP -- measured 11.0 us against
    # 17.2 us as an OWNER24 image (profiles/r06_format_ab_sweep_owner24_border.txt).  (What stays OWNER24: images beyond the Infinity Cache at this
    # gap -- 1 M x 1 M, 40 M non-zeros: 62.4 against 74.1 us -- tests/test_sweep_cpu.py plans that one on the host.)
    denser = host.CSRMatrix.generate("powerlaw", 300000, 300000, a=3.0e6, b=0.4, c=1.0, seed=5)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix_csr(denser)
        assert device.STREAM_FORMATS[eng.stats()["stream_format"]] == "sweep"


@pytest.mark.parametrize("impl", [0, 2])
@pytest.mark.parametrize("slices", ["1", "3"])
def test_partition_loop_and_block_chains(impl, slices, monkeypatch):
    # small banks: many row partitions (hs_run_partition enters a workgroup's chain at the partition's head); HISPARSE_MAX_ROWS=40: far more
    # blocks than workgroups, so every workgroup walks a chain of them (the LDS sums are zeroed and stored once per block)
    monkeypatch.setenv("HISPARSE_SWEEP", "1")
    monkeypatch.setenv("HISPARSE_MAX_ROWS", "40")
    monkeypatch.setenv("HISPARSE_COL_SLICES", slices)
    m = cases.random_csr(30000, 700, 0.02, 5, impl)
    _, cp = cases.formatted(m, impl, 16, 8 if impl == 2 else 2, True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 8, impl))
    want = _oracle(cp, impl, xw)
    with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        assert device.STREAM_FORMATS[st["stream_format"]] == "sweep" and st["num_blocks"] > st["num_workgroups"] and st["col_slices"] == int(slices)
        eng.load_vector(xw)
        eng.run()
        _check(impl, eng.read_result(), want)
        eng.load_vector(np.zeros_like(xw))
        eng.run()
        assert not eng.read_result().any()
        eng.load_vector(xw)
        for j in range(cp.num_row_partitions):                  # the reference's launch loop (sw/benchmark.cpp:318-338)
            eng.run_partition(j, cp.part_len(j))
        _check(impl, eng.read_result(), want)


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("slices", ["11", "16"])
def test_more_than_eight_column_slices(impl, slices, monkeypatch):
    # round 5: SWEEP images may be cut into up to 16 column slices (a short, wide matrix -- one rank's slab -- wants few row ranges, each of
    # which sweeps all of x, and many slices): the kernel's partial rows, the combine pass over 11 / 16 of them, the carried combine
    # (bursts of launches), the device-built image against the host builder's
    monkeypatch.setenv("HISPARSE_SWEEP", "1")
    monkeypatch.setenv("HISPARSE_COL_SLICES", slices)
    csr = host.CSRMatrix.generate("powerlaw", 30000, 400000, a=1.2e6, b=0.4, c=1.0 if impl == 0 else 2.0, seed=31 + impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 9, impl))
    want = _oracle(cp, impl, xw)
    for carry in ("0", "1"):
        monkeypatch.setenv("HISPARSE_CARRY_COMBINE", carry)
        with device.SpmvEngine(impl) as eng:
            eng.load_matrix(cp)
            st = eng.stats()
            assert device.STREAM_FORMATS[st["stream_format"]] == "sweep" and st["col_slices"] == int(slices)
            eng.load_vector(xw)
            eng.run()
            _check(impl, eng.read_result(), want)
            eng.run_batch(3)                                    # three consecutive launches: the carried combine when it is on
            _check(impl, eng.read_result(), want)
            if carry == "0":
                built = eng.read_tiles()
                monkeypatch.setenv("HISPARSE_RETILE", "host")
                with device.SpmvEngine(impl) as host_built:
                    host_built.load_matrix(cp)
                    ref = host_built.read_tiles()
                monkeypatch.delenv("HISPARSE_RETILE")
                assert built["image"].tobytes() == ref["image"].tobytes() and built["blocks"].tobytes() == ref["blocks"].tobytes()


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("resident", ["0", "1"])
def test_stream_policy_does_not_change_the_result(impl, resident, monkeypatch):
    # SWEEP images that fit the Infinity Cache are streamed without the non-temporal hint (hs_api.cpp: stream_resident; its own instantiation
    # of the kernel): both instantiations against the oracle, single launches and a burst
    monkeypatch.setenv("HISPARSE_SWEEP", "1")
    monkeypatch.setenv("HISPARSE_STREAM_RESIDENT", resident)
    csr = host.CSRMatrix.generate("powerlaw", 120000, 120000, a=1.5e6, b=0.4, c=1.0 if impl == 0 else 2.0, seed=41 + impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 11, impl))
    want = _oracle(cp, impl, xw)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        assert device.STREAM_FORMATS[eng.stats()["stream_format"]] == "sweep"
        eng.load_vector(xw)
        for _ in range(3):
            eng.run()
            _check(impl, eng.read_result(), want)
        eng.run_batch(3)
        _check(impl, eng.read_result(), want)


def test_non_finite_x_reaches_only_the_rows_that_hold_the_column(monkeypatch):
    # padding slots carry value 0 at the block's spare row and gather a real x word: 0 x inf = NaN must never land in a real row
    monkeypatch.setenv("HISPARSE_SWEEP", "1")
    impl = 1
    m = cases.random_csr(3000, 5000, 0.004, 2, impl)
    cp = host.format_matrix(host.CSRMatrix.from_scipy(m), impl, skip_empty_rows=True)
    x = cases.random_x(cp.num_cols, 4, impl)
    x[0] = np.inf
    x[4097] = np.nan
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        assert device.STREAM_FORMATS[eng.stats()["stream_format"]] == "sweep"
        eng.load_vector(host.pack_vector(impl, x))
        eng.run()
        got = eng.read_result().view(np.float32)[:3000]
    touched = np.asarray((m[:, [0, 4097]] != 0).sum(axis=1)).ravel() > 0
    assert np.isfinite(got[~touched]).all() and not np.isfinite(got[touched]).any()


def test_device_builder_hands_duplicates_and_wide_column_gaps_to_the_host_loops(monkeypatch):
    """gpu_tiles.hip sorts and emits SWEEP images on the device (byte for byte the host builder's: tests/test_gpu_retile.py); two cases it
    reports instead of building -- a (row, column) that occurs twice, and 64 consecutive elements of a block spanning more than 65535 columns
    (the host cuts that chunk short) -- go through the host loops, and the SpMV is the same either way."""
    import scipy.sparse as sp
    monkeypatch.setenv("HISPARSE_SWEEP", "1")
    rng = np.random.default_rng(5)
    rows, cols = 64, 400000
    wide = sp.csr_matrix((rng.uniform(0.1, 1.0, rows * 3).astype(np.float32), (np.repeat(np.arange(rows), 3), rng.choice(cols, rows * 3, replace=False))), shape=(rows, cols))
    wide.sort_indices()
    dense_enough = cases.random_csr(3000, 5000, 0.004, 2, 0)
    for m, expect_gpu in ((wide, False), (dense_enough, True)):
        monkeypatch.setenv("HISPARSE_COL_SLICES", "1")      # (one slice: the 192 elements of `wide` lie ~2 000 columns apart, 64 of them span 130 K)
        cp = host.format_matrix(host.CSRMatrix.from_scipy(m), 0, skip_empty_rows=True)
        xw = host.pack_vector(0, cases.random_x(cp.num_cols, 4, 0))
        with device.SpmvEngine(0) as eng:
            eng.load_matrix(cp)
            st = eng.stats()
            assert device.STREAM_FORMATS[st["stream_format"]] == "sweep" and bool(st["retiled_on_gpu"]) == expect_gpu, st
            eng.load_vector(xw)
            eng.run()
            _check(0, eng.read_result(), _oracle(cp, 0, xw))
    monkeypatch.delenv("HISPARSE_COL_SLICES")
    # duplicates: the CSR load path keeps them (two entries of one (row, column) are two products)
    indptr = np.array([0, 3, 3, 5], dtype=np.uint32)
    csr = host.CSRMatrix.from_arrays(3, 10, indptr, np.array([7, 7, 2, 0, 0], dtype=np.uint32), np.array([1.0, 2.0, 0.5, 0.25, 0.25], dtype=np.float32))
    with device.SpmvEngine(0) as eng:
        eng.load_matrix_csr(csr)
        st = eng.stats()
        assert device.STREAM_FORMATS[st["stream_format"]] == "sweep" and not st["retiled_on_gpu"]
        eng.load_vector(host.pack_vector(0, np.arange(1, eng.num_cols + 1, dtype=np.float32) * 0.125))
        eng.run()
        y = orc.unpack_result(0, eng.read_result())
    assert y[0] == 3.0 * 1.0 + 0.5 * 0.375 and y[1] == 0.0 and y[2] == 0.5 * 0.125


def test_spmm_runs_over_a_sweep_image(monkeypatch):
    # hs_spmm runs one SpMV per column over formats without a fused kernel (hs_iterate on a sliced sweep image: tests/test_gpu_parity.py's
    # PageRank test under the "sweep" fixture -- the feedback is folded into the slice-combine launch)
    monkeypatch.setenv("HISPARSE_SWEEP", "1")
    monkeypatch.setenv("HISPARSE_COL_SLICES", "2")
    for impl in (0, 1):
        m = cases.random_csr(4096, 4096, 0.003, 9, impl)
        cp = host.format_matrix(host.CSRMatrix.from_scipy(m), impl, skip_empty_rows=True)
        with device.SpmvEngine(impl) as eng:
            eng.load_matrix(cp)
            assert device.STREAM_FORMATS[eng.stats()["stream_format"]] == "sweep" and eng.stats()["col_slices"] == 2
            X = np.stack([host.pack_vector(impl, cases.random_x(cp.num_cols, 20 + j, impl)) for j in range(3)])
            Y = eng.spmm(X)
            for j in range(3):
                _check(impl, Y[j], _oracle(cp, impl, X[j]))
