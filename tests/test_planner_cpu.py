"""The load-time planner's round-6 rules, checked on the host builder (hs_tiles_build: no GPU): the tile census (structured matrices are planned by
where their elements are), hub rows, one-slice float plans, tiny units.  The measurements behind each rule: DESIGN.md section 3 "The planner",
profiles/r06_planner_check_*.txt; the timed counterpart of this file: tests/test_gpu_planner.py.  The emulated kernel (tests/tile_emulator.py) checks the
resulting images against the oracle where the matrix is small enough."""
import os
import sys

import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases            # noqa: E402
import planner_check as pc      # noqa: E402
import tile_emulator    # noqa: E402

PLAN_KEYS = ("HISPARSE_STREAM_FORMAT", "HISPARSE_COL_SLICES", "HISPARSE_MAX_ROWS", "HISPARSE_SWEEP", "HISPARSE_LIGHT", "HISPARSE_PLAN_CENSUS", "HISPARSE_ROW_RUNS", "HISPARSE_AUX_BITS")


@pytest.fixture(autouse=True)
def clean_plan_environment(monkeypatch):
    for k in PLAN_KEYS:
        monkeypatch.delenv(k, raising=False)


def plan(m, impl, workgroups=256):
    cp = host.format_matrix(host.CSRMatrix.from_scipy(m), impl, skip_empty_rows=True)
    t = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, workgroups)
    return cp, t


def parity(cp, t, impl, seed=5):
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, seed, impl))
    want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    got = tile_emulator.run(t, impl, xw, cp.num_rows)
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)


def test_banded_and_block_diagonal_matrices_are_not_sliced(monkeypatch):
    """A row range of a banded matrix meets two or three x sub-tiles: column slices would leave most (range x slice) blocks empty -- 102 of 255 hold anything
    on the 400 K case, 56.9 us against 21.3 us in one slice.  With the census the plan is one slice; without it (the planner of rounds 1-5) it is five."""
    m = pc.banded(200_000, 20, 1_000, 1, 0)
    cp, t = plan(m, 0)
    assert t["format"] in ("delta", "pairs") and t["col_slices"] == 1
    busy = [b for b in t["blocks"] if b["unit_end"] > b["unit_begin"]]
    assert len(busy) >= 0.95 * len(t["blocks"])                      # every workgroup has work
    monkeypatch.setenv("HISPARSE_PLAN_CENSUS", "0")
    _, old = plan(m, 0)
    assert old["col_slices"] > 1
    busy_old = [b for b in old["blocks"] if b["unit_end"] > b["unit_begin"]]
    assert len(busy_old) < 0.6 * len(old["blocks"])                  # ... which the uniform picture could not see
    monkeypatch.delenv("HISPARSE_PLAN_CENSUS")
    small = pc.banded(30_000, 12, 300, 2, 0)                         # small enough for the emulator: the one-slice image against the oracle
    cp, t = plan(small, 0, 32)
    assert t["col_slices"] == 1
    parity(cp, t, 0)
    bd = pc.block_diagonal(120_000, 512, 0.10, 3, 0)
    assert plan(bd, 0)[1]["col_slices"] == 1


def test_effective_gap_keeps_structured_matrices_off_sweep():
    """1 M x 1 M with 12 non-zeros per row has a mean position gap of 85 K -- hyper-sparse by the plain rule, SWEEP -- but a band of 100 K columns: inside the
    cells that hold anything the gap is ~10 K and the row-block kernels run it 2.3 x faster (63.1 -> 27.0 us)."""
    m = pc.banded(400_000, 6, 20_000, 2, 2)
    _, t = plan(m, 2)
    assert t["format"] != "sweep"
    scattered = pc.uniform(400_000, 400_000, 6, 3, 2)                # the same shape without structure: SWEEP
    assert plan(scattered, 2)[1]["format"] == "sweep"


def test_hub_rows_get_per_lane_sums():
    """Rows that hold a large part of their block would put most lanes of a step on one LDS accumulator: DELTA blocks with such a row are flagged for
    per-lane register sums, and DELTA is kept where hub rows hold >= 30 % of the matrix (89 -> 44.9 us on the 500 K case)."""
    m = pc.hubs(200_000, 12, 24, 80_000, 14, 0)
    cp, t = plan(m, 0)
    assert t["format"] == "delta"
    rn = np.diff(m.indptr)
    flagged = 0
    for b in t["blocks"]:
        mine = rn[b["row0"]:min(b["row0"] + b["nrows"], len(rn))]
        hub = mine.size and mine.max() >= 4096 and 8 * int(mine.max()) >= int(mine.sum())
        assert bool(b["flags"] & 1) == bool(hub), (int(b["row0"]), int(b["nrows"]))
        flagged += bool(hub)
    assert 0 < flagged < len(t["blocks"])                            # per block, not everywhere (per-lane sums cost an ordinary graph 20 %)
    small = pc.hubs(20_000, 6, 3, 9_000, 15, 0)
    cp, t = plan(small, 0, 16)
    parity(cp, t, 0)


def test_float_one_slice_plans_become_owner24():
    """float modes: a one-slice PAIRS-family plan of >= 8 M non-zeros is planned again as OWNER24 (10-30 % on five matrices, both float modes); fixed point and
    small matrices keep their plans"""
    m = pc.block_diagonal(360_000, 64, 0.5, 4, 1)
    assert m.nnz >= (8 << 20)
    assert plan(m, 1)[1]["format"] == "owner24"
    fixed = pc.block_diagonal(360_000, 64, 0.5, 4, 0)
    assert plan(fixed, 0)[1]["format"] in ("delta", "pairs")
    small = pc.block_diagonal(100_000, 64, 0.5, 4, 2)
    assert plan(small, 2)[1]["format"] in ("delta", "pairs")


def test_tiny_units_go_to_sweep():
    """Few long rows over millions of columns: below every gap rule, 131 elements per (row range x sub-tile) unit -- 67.5 us as PAIRS, 22.3 us as SWEEP.  Taken
    where the chosen row-block plan has < 1024 elements per non-empty unit AND SWEEP's modelled step is under 60 % of the row-block plan's."""
    m = pc.uniform(2_048, 4_000_000, 800, 27, 0)
    _, t = plan(m, 0)
    assert t["format"] == "sweep"
    slab = pc.reference("mouse_gene_slab8")                          # 14 K elements per unit: SWEEP's model alone would take it (9.0 against 15.1 us modelled)
    assert plan(slab, 0)[1]["format"] == "pairs"                     # ... and be wrong: 8.3 us as PAIRS, 12.5 as SWEEP -- the unit-size condition keeps it
