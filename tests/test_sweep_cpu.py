"""The SWEEP format (hisparse_amd/csrc/sweep_tiles.cpp, stream_tiles.h) checked without a GPU: the image hs_load_matrix would upload is
walked by the numpy emulation of spmv_sweep_kernel (tests/tile_emulator.py: column order inside a block, padding rules, slice bounds are
asserted there) and compared with the oracle."""
import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases
import tile_emulator


def build(cp, impl, workgroups):
    return device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                              cp.num_col_partitions, workgroups)


def oracle_y(cp, impl, xw):
    return orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                    cp.num_col_partitions, cp.ob_bank, cp.vb_bank)


@pytest.fixture(autouse=True)
def forced(monkeypatch):
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "sweep")


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("rows,cols,nnz,wgs,slices", [(60000, 90000, 200000, 16, None), (30000, 200000, 150000, 8, 4), (9000, 70000, 20000, 3, 2), (20000, 400000, 300000, 24, 12), (8000, 300000, 100000, 16, 16),
                                                      (300, 50, 700, 4, None), (1000, 1000, 10000, 256, None)])
def test_sweep_structure_and_parity(impl, rows, cols, nnz, wgs, slices, monkeypatch):
    if slices:
        monkeypatch.setenv("HISPARSE_COL_SLICES", str(slices))
    csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=nnz, b=0.5, c=1.0 if impl == 0 else 2.0, seed=12)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 12, impl) * (40.0 if impl == 0 else 1.0))      # fixed point: some rows saturate
    t = build(cp, impl, wgs)
    assert t["format"] == "sweep" and t["nnz"] == cp.nnz and t["elements"] >= cp.nnz
    assert not slices or t["col_slices"] == slices
    blocks = t["blocks"]
    assert (blocks["nrows"] <= (39715 if impl == 0 else 20479)).all() and t["max_block_rows"] == blocks["nrows"].max()
    # streams of whole steps (one 512-byte chunk per wavefront) followed by the chunk-base tables (one word per wavefront and step)
    W = tile_emulator.SWEEP_WAVES
    steps = blocks["total_steps"][:, 0].astype(np.int64)
    assert len(t["image"]) == steps.sum() * W * (512 + 4)
    assert t["elements"] == steps.sum() * W * 64
    # the slices of a row range are contiguous, disjoint, cover the columns and start on 128-byte lines of x
    order = np.lexsort((blocks["first_col0"], blocks["row0"]))
    per_range = t["col_slices"]
    for b in range(0, len(blocks), per_range):
        mine = blocks[order[b: b + per_range]]
        assert (mine["row0"] == mine["row0"][0]).all() and mine["first_col0"][0] == 0
        assert (mine["first_col0"] % 32 == 0).all()
        assert np.array_equal(mine["first_col0"][1:], (mine["first_col0"] + mine["first_ncols"])[:-1])
        assert mine["first_col0"][-1] + mine["first_ncols"][-1] == cp.num_cols
        for k in range(per_range):
            assert mine["out_offset"][k] == (k * cp.num_rows if per_range > 1 else 0) + mine["row0"][k]
    got = tile_emulator.run(t, impl, xw, cp.num_rows)
    want = oracle_y(cp, impl, xw)
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
    if impl == 0 and rows >= 9000:
        assert (want == 0xFFFFFFFF).any()


@pytest.mark.parametrize("cross", [0, 1])
def test_sweep_row_partitions_and_partition_filter(cross, monkeypatch):
    # float_pob-style small output banks: several row partitions, run one at a time like hs_run_partition; cross = 1 (default since
    # round 5): blocks may reach over partition borders, cross = 0: they end there
    monkeypatch.setenv("HISPARSE_CROSS_PARTITIONS", str(cross))
    m = cases.random_csr(2500, 300, 0.03, 21, 0)
    _, cp = cases.formatted(m, 0, 4, 1, True)
    assert cp.num_row_partitions > 3
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 21, 0))
    t = build(cp, 0, 16)
    assert t["format"] == "sweep"
    blocks = t["blocks"]
    assert (blocks["last_part"] != blocks["row_part"]).any() == bool(cross)
    for g in range(t["num_workgroups"]):
        chain = t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]]
        parts = blocks["row_part"][chain]
        assert (np.diff(parts.astype(np.int64)) >= 0).all()
        last = (blocks["flags"][chain] & 2) != 0
        assert np.array_equal(last, np.append(parts[1:] > blocks["last_part"][chain][:-1], True))
    full = tile_emulator.run(t, 0, xw, cp.num_rows)
    y = np.zeros(cp.num_rows, dtype=np.uint32)
    for j in range(cp.num_row_partitions):
        y = tile_emulator.run(t, 0, xw, cp.num_rows, row_part_filter=j, y_init=y, rows_per_part=128 * cp.ob_bank)
    assert np.array_equal(y, full) and np.array_equal(full, oracle_y(cp, 0, xw))


@pytest.mark.parametrize("impl", [0, 2])
def test_sweep_column_gaps_beyond_16_bits_cut_the_chunk(impl):
    """A chunk's 16-bit column offsets reach 65535 columns beyond its first element: a block whose columns lie further apart gets its
    chunks cut short and padded (value 0, spare accumulator)."""
    rows, cols = 64, 400000
    rng = np.random.default_rng(5)
    indptr = np.arange(0, rows * 3 + 1, 3, dtype=np.uint32)
    indices = np.sort(rng.choice(cols, size=(rows, 3), replace=True), axis=1).astype(np.uint32).ravel()      # ~192 columns spread over 400 K: gaps of thousands, some > 65535 with 1 row range
    data = rng.uniform(0.1, 1.0, rows * 3).astype(np.float32)
    csr = host.CSRMatrix.from_arrays(rows, cols, indptr, indices, data)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 2, impl))
    t = build(cp, impl, 64)      # few non-zeros: few blocks, each with columns far apart
    assert t["format"] == "sweep"
    steps = t["blocks"]["total_steps"][:, 0].astype(np.int64)
    assert steps.sum() * tile_emulator.SWEEP_WAVES * 64 > 4 * cp.nnz      # (mostly padding: the cuts happened)
    got, want = tile_emulator.run(t, impl, xw, cp.num_rows), oracle_y(cp, impl, xw)
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)


def test_sweep_duplicates_and_empty_matrix():
    # duplicate (row, column) entries are separate elements (their products add up); a matrix without non-zeros gives blocks without steps
    indptr = np.array([0, 3, 3, 5], dtype=np.uint32)
    indices = np.array([7, 7, 2, 0, 0], dtype=np.uint32)
    data = np.array([1.0, 2.0, 0.5, 0.25, 0.25], dtype=np.float32)
    csr = host.CSRMatrix.from_arrays(3, 10, indptr, indices, data)
    cp = host.format_matrix(csr, 0, skip_empty_rows=False)
    xw = host.pack_vector(0, np.arange(1, cp.num_cols + 1, dtype=np.float32) * 0.125)
    t = build(cp, 0, 8)
    got = tile_emulator.run(t, 0, xw, cp.num_rows)
    assert np.array_equal(got, oracle_y(cp, 0, xw)) and orc.unpack_result(0, got)[0] == 3.0 * 1.0 + 0.5 * 0.375
    empty = host.CSRMatrix.from_arrays(5, 9, np.zeros(6, dtype=np.uint32), np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.float32))
    cpe = host.format_matrix(empty, 0, skip_empty_rows=False)
    te = build(cpe, 0, 8)
    assert te["format"] == "sweep" and te["nnz"] == 0 and (te["blocks"]["total_steps"][:, 0] == 0).all()
    assert not tile_emulator.run(te, 0, host.pack_vector(0, np.ones(cpe.num_cols, dtype=np.float32)), cpe.num_rows).any()


def test_planner_takes_sweep_where_its_plan_is_modelled_faster(monkeypatch):
    """Unforced: SWEEP wherever OWNER24 would be taken and SWEEP's plan is modelled faster (stream_tiles.cpp) -- a very sparse square, and a row
    slab of a matrix that stays OWNER24 as a whole (few row ranges: few lines of x to gather, while OWNER24's units cost what they cost)."""
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    sparse = host.CSRMatrix.generate("powerlaw", 1000000, 1000000, a=5.0e6, b=0.4, c=1.0, seed=3)          # mean position gap 200 K
    # gap 25 K and an image beyond the Infinity Cache (1 M x 1 M, 40 M non-zeros, 324 MB): OWNER24 (measured 62.4 us against 74.1 as a SWEEP image,
    # profiles/r06_format_ab_sweep_owner24_border.txt) ...
    denser = host.CSRMatrix.generate("powerlaw", 1000000, 1000000, a=4.0e7, b=0.35, c=1.0, seed=9)
    ip, ix, dv = denser.arrays()
    cut = int(np.searchsorted(ip, ip[-1] // 8))                                                           # ... its first eighth of the non-zeros as a slab: SWEEP
    slab = host.CSRMatrix.from_arrays(cut, denser.num_cols, ip[:cut + 1].copy(), ix[:ip[cut]].copy(), dv[:ip[cut]].copy())
    # ... and since round 6 (SWEEP model fitted again on the round-5 kernel) a gap-32 K matrix whose image stays in the cache: SWEEP (35.1 against 41.7 us)
    resident = host.CSRMatrix.generate("powerlaw", 800000, 800000, a=2.0e7, b=0.4, c=1.0, seed=4)
    for csr, want in ((sparse, "sweep"), (denser, "owner24"), (slab, "sweep"), (resident, "sweep")):
        cp = host.format_matrix(csr, 0, skip_empty_rows=True)
        t = build(cp, 0, 256)
        assert t["format"] == want, (csr.num_rows, csr.nnz, t["format"])
    monkeypatch.setenv("HISPARSE_SWEEP", "0")
    cp = host.format_matrix(sparse, 0, skip_empty_rows=True)
    assert build(cp, 0, 256)["format"] == "owner24"


def test_planner_takes_sweep_for_short_wide_slabs_that_fit_the_infinity_cache(monkeypatch):
    """Round 5 (stream_tiles.cpp): a fixed-point row slab of >= 6 columns per row, mean position gap in (8 K, 20 K], > 2 M non-zeros and an image below
    256 MiB -- one rank's slab of hollywood split 8 ways -- is a SWEEP image; the same shape in a float mode, a denser one and a squarer one keep
    the row-block planner's choice; the emulated kernel agrees with the oracle on the slab."""
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    slab = host.CSRMatrix.generate("powerlaw", 50000, 400000, a=2.2e6, b=0.4, c=1.0, seed=5)             # gap 9.1 K, 8 columns per row
    cp = host.format_matrix(slab, 0, skip_empty_rows=True)
    t = build(cp, 0, 256)
    assert t["format"] == "sweep", t["format"]
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 5, 0))
    assert np.array_equal(tile_emulator.run(t, 0, xw, cp.num_rows), oracle_y(cp, 0, xw))
    assert build(host.format_matrix(slab, 1, skip_empty_rows=True), 1, 256)["format"] != "sweep"         # float modes: not measured, not switched
    denser = host.CSRMatrix.generate("powerlaw", 50000, 400000, a=4.0e6, b=0.4, c=1.0, seed=6)           # gap 5 K
    assert build(host.format_matrix(denser, 0, skip_empty_rows=True), 0, 256)["format"] != "sweep"
    squarer = host.CSRMatrix.generate("powerlaw", 100000, 400000, a=4.2e6, b=0.4, c=1.0, seed=7)         # gap 9.5 K, 4 columns per row
    assert build(host.format_matrix(squarer, 0, skip_empty_rows=True), 0, 256)["format"] != "sweep"
