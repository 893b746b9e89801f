"""The LIGHT plan (round 4; stream_tiles.h, spmv_kernels.hip: spmv_light_kernel): small matrices in ONE launch of 256-thread workgroups over
a PAIRS image, x gathered from L2.  Chosen automatically below kLightMaxNnz non-zeros and 16 x sub-tiles; every case against the oracle
(bit-exact in fixed point), the reference's partition loop, chains of blocks per workgroup, and the image byte for byte against the host
builder.  (tests/test_gpu_parity.py runs its whole case list through this plan as the stream format "light".)"""
import numpy as np
import pytest

from hisparse_amd import datasets, device, host
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
    for k in ("HISPARSE_STREAM_FORMAT", "HISPARSE_LIGHT", "HISPARSE_MAX_ROWS", "HISPARSE_COL_SLICES", "HISPARSE_RETILE"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("shape", [(1000, 1000, 0.01), (5638, 45101, 0.007), (512, 33288, 0.05), (300000, 64, 0.05), (70, 50000, 0.01), (8, 8, 1.0), (20000, 20000, 0.00002)])
def test_small_matrices_take_the_light_plan_and_match_the_oracle(impl, shape):
    rows, cols, density = shape
    csr = host.CSRMatrix.generate("bernoulli", rows, cols, b=density, c=1.0 if impl == 0 else 0.5, seed=rows + impl)
    if impl == 0:
        ip, ix, dv = csr.arrays()
        csr = host.CSRMatrix.from_arrays(rows, cols, ip, ix, np.abs(dv))
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl))
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        dense_rows = st["nnz"] >= 0.125 * rows * cols and cols >= 2048
        assert st["light_kernel"] == (0 if dense_rows else 1), st
        if st["light_kernel"]:
            assert st["col_slices"] == 1 and device.STREAM_FORMATS[st["stream_format"]] == "pairs"
            assert st["num_blocks"] <= 6 * st["num_compute_units"] and st["lds_bytes"] <= 24576 + 16
        eng.load_vector(xw)
        eng.run()
        want = _oracle(cp, impl, xw)
        _check(impl, eng.read_result(), want)
        eng.run()                                                # accumulators are re-armed by every launch
        _check(impl, eng.read_result(), want)
        tiles = eng.read_tiles()
    built = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, st["num_compute_units"])
    assert built["image"].tobytes() == tiles["image"].tobytes() and built["blocks"].tobytes() == tiles["blocks"].tobytes() and built["units"].tobytes() == tiles["units"].tobytes()


@pytest.mark.parametrize("impl", [0, 2])
def test_partition_loop_and_block_chains(impl, monkeypatch):
    # small banks: many row partitions (hs_run_partition enters a workgroup's chain at the partition's head); HISPARSE_MAX_ROWS=3: far more
    # blocks than workgroups, so every workgroup walks a chain of them
    monkeypatch.setenv("HISPARSE_LIGHT", "1")
    monkeypatch.setenv("HISPARSE_MAX_ROWS", "3")
    m = cases.random_csr(9000, 700, 0.02, 5, impl)
    _, cp = cases.formatted(m, impl, 16, 8 if impl == 2 else 2, True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 8, impl))
    want = _oracle(cp, impl, xw)
    with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        assert st["light_kernel"] == 1 and st["num_blocks"] > st["num_workgroups"]
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


def test_non_finite_x_reaches_only_the_rows_that_hold_the_column():
    # padding slots carry value 0 at the block's spare row: 0 x inf = NaN must never land in a real row
    impl = 1
    m = cases.random_csr(3000, 5000, 0.004, 2, impl)
    csr = host.CSRMatrix.from_scipy(m)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    x = cases.random_x(cp.num_cols, 4, impl)
    x[0] = np.inf
    x[4097] = np.nan
    xw = host.pack_vector(impl, x)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        assert eng.stats()["light_kernel"] == 1
        eng.load_vector(xw)
        eng.run()
        got = eng.read_result().view(np.float32)[:3000]
    touched = np.asarray((m[:, [0, 4097]] != 0).sum(axis=1)).ravel() > 0
    assert np.isfinite(got[~touched]).all() and not np.isfinite(got[touched]).any()


def test_pruned_nn_layers_below_bitmap_density_are_one_launch():
    # transformer-90 / -95 (sw/bm.sh:26-27): 1.7 M / 0.85 M non-zeros -- 5 column slices + a combine launch in round 3 (16.4 us).
    # Round 4: both the LIGHT plan.  Round 5: in FIXED point the 10 %-dense layer runs as a sliced DELTA plan whose combine pass is carried
    # into the next step's kernel (7.9 against 8.6 us; stream_tiles.cpp) -- still one launch per step in a run; the float modes and the
    # 5 %-dense layer keep the LIGHT plan
    for name, impl, light in (("transformer_90", 0, False), ("transformer_95", 0, True), ("transformer_90", 1, True), ("transformer_95", 1, True)):
        cfg, csr = datasets.load(name)
        with device.SpmvEngine(impl) as eng:
            eng.load_matrix_csr(csr)
            st = eng.stats()
            if light:
                assert st["light_kernel"] == 1 and st["col_slices"] == 1, (name, impl, st)
            else:
                assert st["light_kernel"] == 0 and device.STREAM_FORMATS[st["stream_format"]] == "delta" and st["col_slices"] == 5, (name, impl, st)
                assert st["stream_bytes"] < 48 << 20          # small enough for the carried combine: one launch per step
