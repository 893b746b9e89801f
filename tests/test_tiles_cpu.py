"""Load-time re-tiling (hisparse_amd/csrc/stream_tiles.cpp) checked without a GPU: the image hs_load_matrix would
upload is walked by a numpy emulation of the kernel (tests/tile_emulator.py) and compared with the oracle."""
import os

import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases
import tile_emulator


@pytest.fixture(autouse=True, params=["pairs", "delta", "bitmap", "pairs24"])
def stream_format(request, monkeypatch):
    # every test of this module runs once per device stream format (stream_tiles.h); "pairs24" = PAIRS with 24-bit position words
    # (opt-in, HISPARSE_AUX_BITS=24; taken whenever no block has more than 2046 rows)
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", request.param[:5] if request.param.startswith("pairs") else request.param)
    if request.param == "pairs24":
        monkeypatch.setenv("HISPARSE_AUX_BITS", "24")
    return request.param


def build(cp, impl, workgroups):
    return device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                              cp.num_col_partitions, workgroups)


def oracle_y(cp, impl, xw):
    return orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                    cp.num_col_partitions, cp.ob_bank, cp.vb_bank)


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("skip", [False, True])
@pytest.mark.parametrize("rows,cols,density,vb,ob,wgs", [(300, 50, 0.05, 2, 8, 4), (3000, 700, 0.02, 16, 8, 7), (1000, 1000, 0.01, 4096, 8192, 256)])
def test_emulated_kernel_matches_oracle(impl, skip, rows, cols, density, vb, ob, wgs):
    if impl == 1 and ob == 8192:
        ob = 1024
    m = cases.random_csr(rows, cols, density, 5, impl)
    _, cp = cases.formatted(m, impl, vb, ob, skip)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 5, impl))
    tiles = build(cp, impl, wgs)
    assert tiles["nnz"] == m.nnz
    got = tile_emulator.run(tiles, impl, xw, cp.num_rows)
    want = oracle_y(cp, impl, xw)
    if impl == 0:
        assert np.array_equal(got, want)
    else:
        assert cases.float_close(got, want)


def test_structure_invariants(stream_format, monkeypatch):
    if stream_format == "bitmap":
        pytest.skip("element-stream structure; BITMAP has its own test below")
    monkeypatch.setenv("HISPARSE_COL_SLICES", "1")      # the unsliced structure (the planner would slice this 7-sub-tile matrix; sliced plans: test_column_slices*)
    csr = host.CSRMatrix.generate("powerlaw", 30000, 50000, a=600000, b=0.4, c=1.0, seed=3)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    t = build(cp, 0, 64)
    units = t["units"]
    # the kernel's walk: workgroup g starts at blocks[g] and follows `next`; same blocks as the host-side wg_first/block_order
    for g in range(t["num_workgroups"]):
        chain, b = [], g
        while True:
            chain.append(b)
            b = int(t["blocks"][b]["next"])
            if b == 0:
                break
        assert chain == t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]].tolist()
    blocks = t["blocks"][np.argsort(t["blocks"]["row0"], kind="stable")]
    # blocks tile the rows exactly, within the LDS budget, never across a row partition
    assert t["col_slices"] == 1 and t["ring_buffers"] == 4
    assert blocks["row0"][0] == 0 and (blocks["row0"][1:] == blocks["row0"][:-1] + blocks["nrows"][:-1]).all()
    assert blocks["row0"][-1] + blocks["nrows"][-1] == cp.num_rows
    assert (blocks["out_offset"] == blocks["row0"]).all()
    assert blocks["nrows"].max() <= 4095 and t["max_block_rows"] == blocks["nrows"].max()
    # every block is owned by exactly one workgroup
    assert sorted(t["block_order"].tolist()) == list(range(len(blocks)))
    assert t["wg_first"][0] == 0 and t["wg_first"][-1] == len(blocks) and t["num_workgroups"] <= 64
    # sub-tiles: multiple-of-8 widths inside one column partition, wave streams monotone
    assert (units["ncols"] % 8 == 0).all() and (units["ncols"] <= 8192).all() and (units["ncols"] > 0).all()
    assert ((units["col0"] // 32768) == ((units["col0"] + units["ncols"] - 1) // 32768)).all()
    for b in blocks[:5]:
        es = units["end_step"][b["unit_begin"]:b["unit_end"]]
        assert (np.diff(es, axis=0) >= 0).all()
        assert (b["total_steps"] == es[-1]).all() and (b["first_end"] == es[0]).all() and b["first_col0"] == units["col0"][b["unit_begin"]] and b["first_ncols"] == units["ncols"][b["unit_begin"]]
    assert t["format"] == stream_format and t["elements"] >= t["nnz"]
    if stream_format in ("pairs", "pairs24"):
        # bytes: 8 (7 with 24-bit position words) per element slot, padding below 64 slots per unit
        assert len(t["image"]) == t["elements"] * (8 if stream_format == "pairs" else 7)
        assert t["elements"] - t["nnz"] < 64 * len(units)
    else:
        # bytes: 768 per record = two slots per lane, 64 x (2 x u32 value + 2 x u16 gap); per (unit, wavefront) that has work one head slot
        # and, when that makes the slot count odd, one dead slot
        records = sum(int(units["end_step"][b["unit_end"] - 1].sum()) for b in blocks if b["unit_end"] > b["unit_begin"])
        assert len(t["image"]) == records * 768
        extra = 2 * records - t["elements"] // 64           # head + dead slots
        assert 0 < extra <= 2 * 14 * len(units)
    # balance: no workgroup carries more than ~1.5x the mean (power-law rows, 64 groups)
    loads = []
    for g in range(t["num_workgroups"]):
        bs = t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]]
        loads.append(sum(int(units["end_step"][blocks[b]["unit_end"] - 1].sum()) if blocks[b]["unit_end"] > blocks[b]["unit_begin"] else 0 for b in bs))
    assert max(loads) < 1.5 * (sum(loads) / len(loads)) + 64


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("slices", [2, 4])
def test_column_slices(impl, slices, monkeypatch):
    # 2-D decomposition: (row range x column slice) blocks writing per-slice partials + combine pass
    if os.environ.get("HISPARSE_STREAM_FORMAT") == "bitmap":
        pytest.skip("sub-tile slices; BITMAP slices are covered by test_bitmap_structure_and_parity")
    monkeypatch.setenv("HISPARSE_COL_SLICES", str(slices))
    csr = host.CSRMatrix.generate("powerlaw", 20000, 60000, a=300000, b=0.4, c=1.0 if impl == 0 else 2.0, seed=11)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl))
    t = build(cp, impl, 32)
    assert t["col_slices"] == slices and t["nnz"] == cp.nnz
    units = t["units"]
    blocks = t["blocks"][np.lexsort((t["blocks"]["out_offset"], t["blocks"]["row0"]))]   # row range major, slice minor
    assert len(blocks) % slices == 0 and blocks["nrows"].max() <= 12287
    # the slices of one row range own disjoint sub-tiles, every touched sub-tile exactly once, and carry about the same work
    for b in range(0, len(blocks), slices):
        seen, loads = [], []
        for k in range(slices):
            blk = blocks[b + k]
            u = units[blk["unit_begin"]:blk["unit_end"]]
            seen += (u["col0"] // 8192).tolist()
            loads.append(int(u["end_step"][-1].sum()) if len(u) else 0)
            assert blk["out_offset"] == k * cp.num_rows + blocks[b]["row0"] and blk["row0"] == blocks[b]["row0"]
        assert len(seen) == len(set(seen))
        assert max(loads) <= 1.25 * (sum(loads) / slices) + 14
    got = tile_emulator.run(t, impl, xw, cp.num_rows)
    want = oracle_y(cp, impl, xw)
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)


def test_format_choice(monkeypatch):
    # unforced: DELTA is tried for mean position gaps rows*cols/nnz in [8, 20000] and kept when it needs <= 2 % bridge slots and saves
    # more than 23 MiB of stream against PAIRS (stream_tiles.cpp); gaps beyond 20000: OWNER24; everything else is PAIRS
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    for rows, cols, nnz, want in [(20000, 60000, 150000, "pairs"),       # right gap, but a 1 MB image: nothing to save
                                  (40000, 40000, 16000000, "delta"),     # 128 MB in PAIRS, 100 MB in DELTA
                                  (300000, 400000, 200000, "owner")]:    # hyper-sparse, too wide for the LIGHT plan (52 sub-tiles): OWNER24 (fixed point: saturating 32-bit accumulators)
        csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=nnz, b=0.0, c=1.0, seed=2)
        cp = host.format_matrix(csr, 0, skip_empty_rows=True)
        t = build(cp, 0, 16)
        assert t["format"][:5] == want          # "pairs" or its 7-byte form "pairs24" (blocks of <= 2046 rows)
        if want == "delta":
            assert t["image"].size + (23 << 20) < 8 * cp.nnz
    # inside a DELTA matrix, blocks of heavy rows are flagged for per-lane register sums, the sparse bulk is not
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "delta")
    monkeypatch.setenv("HISPARSE_COL_SLICES", "1")       # short row ranges (the planner would slice this matrix: long ranges, no dense block)
    csr = host.CSRMatrix.generate("powerlaw", 30000, 60000, a=2400000, b=0.8, c=1.0, seed=3)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    t = build(cp, 0, 64)
    flags = t["blocks"]["flags"] & 1
    assert t["format"] == "delta" and flags.any() and not flags.all()
    # flagged: the blocks of FEW long rows (mean gap below kDenseMeanGap) -- and, since round 6, a block whose heaviest row holds an eighth of it
    # (a hub row would put most lanes of a step on one LDS accumulator: stream_tiles.cpp, "hub rows")
    ip = csr.arrays()[0].astype(np.int64)
    row_nnz = np.diff(ip)
    few_rows = t["blocks"]["nrows"][flags == 0].min()
    for b in t["blocks"][flags == 1]:
        if b["nrows"] >= few_rows:
            mine = row_nnz[b["row0"]:min(b["row0"] + b["nrows"], len(row_nnz))]
            assert mine.max() >= 4096 and 8 * mine.max() >= mine.sum(), (int(b["row0"]), int(b["nrows"]))
    for b in t["blocks"][flags == 0]:
        mine = row_nnz[b["row0"]:min(b["row0"] + b["nrows"], len(row_nnz))]
        assert not (mine.max() >= 4096 and 8 * mine.max() >= mine.sum())
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bogus")
    with pytest.raises(device.DeviceError):
        build(cp, 0, 16)


def test_delta_bridges(stream_format):
    # a hyper-sparse unit: most gaps exceed 16 bits and need bridge slots; rows near the end of a 12287-row block
    if stream_format != "delta":
        pytest.skip("bridges exist in the DELTA format only")
    for impl in (0, 2):
        csr = host.CSRMatrix.generate("powerlaw", 30000, 40000, a=15000, b=0.0, c=1.0 if impl == 0 else 2.0, seed=8)
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
        xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 9, impl))
        t = build(cp, impl, 4)
        assert t["elements"] > 1.3 * t["nnz"]            # bridges are there
        got = tile_emulator.run(t, impl, xw, cp.num_rows)
        want = oracle_y(cp, impl, xw)
        assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)


@pytest.mark.parametrize("cross", [0, 1])
def test_many_row_partitions_and_partition_filter(cross, monkeypatch):
    # float_pob-style small output banks: several row partitions, run one at a time like hs_run_partition.
    # cross = 1 (the default since round 5): row ranges are cut by non-zeros and the LDS cap alone and may reach over partition borders;
    # cross = 0: they end at the borders, and every partition is balanced over the workgroups by itself
    monkeypatch.setenv("HISPARSE_CROSS_PARTITIONS", str(cross))
    m = cases.random_csr(2500, 300, 0.03, 21, 0)
    _, cp = cases.formatted(m, 0, 4, 1, True)
    assert cp.num_row_partitions > 3
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 21, 0))
    t = build(cp, 0, 16)
    blocks = t["blocks"]
    crossing = (blocks["last_part"] != blocks["row_part"]).any()
    assert crossing == bool(cross)
    assert (blocks["row_part"] == blocks["row0"] // (128 * cp.ob_bank)).all()
    assert (blocks["last_part"][blocks["nrows"] > 0] == ((blocks["row0"] + blocks["nrows"] - 1) // (128 * cp.ob_bank))[blocks["nrows"] > 0]).all()
    # a workgroup's chain visits the row partitions in order; the last-of-partition flag: the next block begins beyond where this one ends
    for g in range(t["num_workgroups"]):
        chain = t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]]
        parts = blocks["row_part"][chain]
        assert (np.diff(parts.astype(np.int64)) >= 0).all()
        last = (blocks["flags"][chain] & 2) != 0
        assert np.array_equal(last, np.append(parts[1:] > blocks["last_part"][chain][:-1], True))
    if not cross:      # each partition spread over the workgroups (block counts per workgroup differ by at most one)
        for p in range(cp.num_row_partitions):
            per_wg = [int((blocks["row_part"][t["block_order"][t["wg_first"][g]:t["wg_first"][g + 1]]] == p).sum()) for g in range(t["num_workgroups"])]
            assert max(per_wg) - min(per_wg) <= 1
    else:              # the whole SpMV spread over the workgroups
        per_wg = [int(t["wg_first"][g + 1] - t["wg_first"][g]) for g in range(t["num_workgroups"])]
        assert max(per_wg) - min(per_wg) <= 1
    full = tile_emulator.run(t, 0, xw, cp.num_rows)
    y = np.zeros(cp.num_rows, dtype=np.uint32)
    for j in range(cp.num_row_partitions):
        before = y.copy()
        y = tile_emulator.run(t, 0, xw, cp.num_rows, row_part_filter=j, y_init=y, rows_per_part=128 * cp.ob_bank)
        lo, hi = j * 128 * cp.ob_bank, min(cp.num_rows, (j + 1) * 128 * cp.ob_bank)
        assert np.array_equal(y[:lo], before[:lo]) and np.array_equal(y[hi:], before[hi:])      # rows of other partitions keep their contents
    assert np.array_equal(y, full) and np.array_equal(full, oracle_y(cp, 0, xw))


def corrupt(cp, fn):
    chans = [cp.channel(c) for c in range(16)]
    fn(chans)
    return chans


def test_malformed_images_are_rejected():
    m = cases.random_csr(600, 80, 0.05, 2, 0)
    _, cp = cases.formatted(m, 0, 4, 2, True)
    args = (0, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, 8)

    def col_out_of_range(ch):
        payload = cp.num_partitions * 2
        rows, lanes = np.nonzero(ch[0][payload:, :8] != 0xFFFFFFFF)
        ch[0][payload + rows[0], lanes[0]] = 40          # partition-local column beyond the 32-column partition
    with pytest.raises(device.DeviceError) as e:
        device.build_tiles(corrupt(cp, col_out_of_range), *args)
    assert e.value.code == -4 and "column" in str(e.value)

    def truncated(ch):
        ch[3] = ch[3][: cp.num_partitions * 2 + 1]
    with pytest.raises(device.DeviceError) as e:
        device.build_tiles(corrupt(cp, truncated), *args)
    assert e.value.code == -4

    def huge_marker(ch):
        payload = cp.num_partitions * 2
        rows, lanes = np.nonzero(ch[1][payload:, :8] == 0xFFFFFFFF)
        ch[1][payload + rows[0], 8 + lanes[0]] = 200 << 24   # skip 200 rounds: the next non-zero of that lane leaves the partition
    with pytest.raises(device.DeviceError) as e:
        device.build_tiles(corrupt(cp, huge_marker), *args)
    assert e.value.code == -4 and "row" in str(e.value)

    def short_headers(ch):
        ch[0] = ch[0][:1]
    with pytest.raises(device.DeviceError):
        device.build_tiles(corrupt(cp, short_headers), *args)


def marker_limit_matrix():
    import scipy.sparse as sp
    rows = 128 * 400
    r = np.array([5, 5 + 128 * 300, 5 + 128 * 350])          # one lane stream; 299 empty rounds between the first two rows
    return sp.csr_matrix((np.array([1.0, 2.0, 3.0], dtype=np.float32), (r, np.array([3, 4, 5]))), shape=(rows, 64)), r


def test_marker_skip_count_beyond_255_is_chained():
    """SURVEY.md Appendix B.7 / spmv/libfpga/spmv_cluster.h:81-82: in fixed mode the skip count travels in the 8 integer
    bits of a Q8.24 word.  The reference's formatter writes a jump of >= 256 rounds as ONE saturated word, which every
    decoder (its own included) reads as 255: the following rows land 45 rounds too early, silently.  The product's
    formatter emits a chain of markers (255 + 45) instead -- decoders add consecutive markers up -- so the image is valid
    and y lands on the right rows; the float modes carry the count as a raw integer and need no chain."""
    m, r = marker_limit_matrix()
    for impl in (0, 1):
        csr = host.CSRMatrix.from_scipy(m)
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
        xw = host.pack_vector(impl, np.ones(cp.num_cols, dtype=np.float32))
        want = oracle_y(cp, impl, xw)
        got = tile_emulator.run(build(cp, impl, 8), impl, xw, cp.num_rows)
        assert np.array_equal(got, want)
        assert np.nonzero(orc.unpack_result(impl, want))[0].tolist() == r.tolist()
        assert orc.unpack_result(impl, want)[r].tolist() == [1.0, 2.0, 3.0]


def test_reference_formatter_marker_limit_quirk_is_documented():
    """The same matrix through the ORACLE's restatement of the reference formatter (saturating marker word): the oracle's
    decoder puts rows 2 and 3 of the lane stream 45 rounds early.  This pins what the reference does; the product does not
    reproduce it (previous test)."""
    from oracle import cpsr_format as of
    m, r = marker_limit_matrix()
    ref, rows, cols, rp, cpn = of.format_matrix(0, m.shape[0], m.shape[1], m.data, m.indices, m.indptr, 8192, 4096, True)
    xw = host.pack_vector(0, np.ones(cols, dtype=np.float32))
    y = orc.spmv(0, ref, xw, rows, cols, rp, cpn, 8192, 4096)
    assert np.nonzero(orc.unpack_result(0, y))[0].tolist() == [5, 5 + 128 * 255, 5 + 128 * 305]


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("rows,cols,density,wgs", [(512, 4200, 0.5, 256), (40, 9000, 0.3, 64), (3000, 2500, 0.2, 16), (129, 2049, 0.9, 8)])
def test_bitmap_structure_and_parity(impl, rows, cols, density, wgs, monkeypatch):
    """Dense-row matrices pick the BITMAP format by themselves; image = one 64-bit mask per (row, 64-column group) + the compacted
    values; every (row, group) belongs to exactly one wavefront run (checked inside the emulator); y matches the oracle."""
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    m = cases.random_csr(rows, cols, density, 31, impl)
    ob = 1024 if impl == 1 else 8192
    _, cp = cases.formatted(m, impl, 4096, ob, True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 31, impl))
    t = build(cp, impl, wgs)
    if t["format"] == "delta":
        # round 6: a FIXED-point dense-row layer over ONE x sub-tile is a DELTA image by the planner's own rule (no refills, no unit barriers, per-lane
        # row sums; until round 5 the rule needed two sub-tiles) -- 2048 x 8192 at 15 %: 6.7 us against 11.6 us as a BITMAP image
        # (profiles/r06_planner_check_before.txt / _after.txt).  The BITMAP structure below is then checked on the forced image.
        assert impl == 0 and cols <= 8192 and m.nnz >= (1 << 20) and t["col_slices"] == 1 and (rows, cols) == (3000, 2500)      # (where the fitted costs say so: stream_tiles.cpp)
        monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bitmap")
        t = build(cp, impl, wgs)
    assert t["format"] == "bitmap" and t["nnz"] == m.nnz and t["elements"] == m.nnz
    blocks, slices = t["blocks"], t["col_slices"]
    groups = (cp.num_cols + 63) // 64
    # fewer rows than workgroups: the columns are cut into slices so that every workgroup has a block
    assert slices == max(1, min(8, 1 << int(np.floor(np.log2(max(1, wgs // cp.num_rows))))))
    assert len(t["units"]) == 80 * len(blocks) and (blocks["unit_end"] - blocks["unit_begin"] == 80).all()   # 16 run headers x 5 slots
    # masks: rows x groups x 8 bytes in total (every row belongs to `slices` blocks that split its groups) + zero padding of
    # at most 23 masks per wavefront run; values: 4 bytes, padded to 8 per block
    mask_bytes = int((blocks["nrows"].astype(np.int64) * blocks["first_ncols"]).sum()) * 8
    assert mask_bytes == cp.num_rows * groups * 8
    runs = int(np.where(blocks["nrows"] * 2 <= 16, blocks["nrows"] * (16 // np.maximum(blocks["nrows"], 1)), blocks["nrows"]).sum())
    assert mask_bytes + 4 * m.nnz + 16 * 8 * runs <= len(t["image"]) <= mask_bytes + 4 * m.nnz + 4 * len(blocks) + 23 * 8 * runs
    assert len(t["image"]) < 0.75 * 8 * m.nnz + 23 * 8 * runs      # the point of the format: well under 8 bytes per non-zero
    assert blocks["nrows"].max() <= 8191 and t["max_block_rows"] == blocks["nrows"].max()
    for b in blocks:
        assert b["out_offset"] % cp.num_rows == b["row0"] and b["first_col0"] % 64 == 0
    got = tile_emulator.run(t, impl, xw, cp.num_rows)
    want = oracle_y(cp, impl, xw)
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)


@pytest.mark.parametrize("impl,ob,rows,cols", [(0, 8192, 700, 5000), (1, 1024, 300, 9000), (2, 8192, 1500, 4100)])
def test_bitmap_walk_by_lane_gives_the_same_image(impl, ob, rows, cols, monkeypatch):
    """The host BITMAP builder decodes the CPSR image with one task per (channel, packet lane) on hosts with many threads and one per
    channel otherwise: same image, Block[] and run headers either way (column partitions of a row still arrive in ascending order)."""
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bitmap")
    m = cases.random_csr(rows, cols, 0.3, 77, impl)
    _, cp = cases.formatted(m, impl, 256, ob, True)      # 2048 columns per column partition
    built = []
    for lanes in ("0", "1"):
        monkeypatch.setenv("HISPARSE_WALK_LANES", lanes)
        built.append(build(cp, impl, 64))
    a, b = built
    assert a["format"] == "bitmap" and cp.num_col_partitions > 1
    assert np.array_equal(a["image"], b["image"]) and a["blocks"].tobytes() == b["blocks"].tobytes() and a["units"].tobytes() == b["units"].tobytes()


def test_bitmap_choice_unsorted_rows_and_duplicates(monkeypatch):
    import scipy.sparse as sp
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    # below 1/8 density or with short rows the element streams stay
    for rows, cols, density, want in [(256, 4096, 0.10, False), (256, 1024, 0.5, False), (256, 4096, 0.13, True)]:
        m = cases.random_csr(rows, cols, density, 4, 0)
        _, cp = cases.formatted(m, 0, 4096, 8192, True)
        assert (build(cp, 0, 16)["format"] == "bitmap") == want
    # CSR input with unsorted columns inside the rows (legal for the reference): the bitmap builder sorts them
    rng = np.random.default_rng(6)
    m = cases.random_csr(200, 3000, 0.3, 6, 0)
    for r in range(m.shape[0]):
        a, b = m.indptr[r], m.indptr[r + 1]
        perm = rng.permutation(b - a)
        m.indices[a:b], m.data[a:b] = m.indices[a:b][perm], m.data[a:b][perm]
    csr = host.CSRMatrix.from_arrays(200, 3000, m.indptr.astype(np.uint32), m.indices.astype(np.uint32), m.data)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 6, 0))
    t = build(cp, 0, 16)
    assert t["format"] == "bitmap"
    assert np.array_equal(tile_emulator.run(t, 0, xw, cp.num_rows), oracle_y(cp, 0, xw))
    # the same column twice in a row cannot be one bit: the element streams take over, result still right
    ip = np.array([0, 3] + [3] * 127, dtype=np.uint32)
    ix = np.array([5, 5, 9], dtype=np.uint32)
    dv = np.array([1.0, 2.0, 4.0], dtype=np.float32)
    csr = host.CSRMatrix.from_arrays(128, 4096, ip, ix, dv)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bitmap")
    xw = host.pack_vector(0, np.ones(cp.num_cols, dtype=np.float32))
    t = build(cp, 0, 4)
    assert t["format"] != "bitmap"
    y = tile_emulator.run(t, 0, xw, cp.num_rows)
    assert orc.unpack_result(0, y)[0] == 7.0 and np.array_equal(y, oracle_y(cp, 0, xw))


@pytest.mark.parametrize("impl", [1, 2])
@pytest.mark.parametrize("rows,cols,nnz,wgs,slices", [(60000, 90000, 200000, 16, None), (30000, 200000, 150000, 8, 4), (9000, 70000, 20000, 3, 2)])
def test_owner_structure_and_parity(impl, rows, cols, nnz, wgs, slices, monkeypatch):
    """Hyper-sparse float matrices pick the OWNER format: float accumulators, wavefront-private rows (checked inside the emulator:
    sorted lane-major runs, one owner per row, padding at the wavefront's spare accumulator); y matches the oracle."""
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    monkeypatch.setenv("HISPARSE_LIGHT", "0")      # (matrices this small would take the LIGHT plan: the choice among the row-block kernel's formats is what is tested)
    if slices:
        monkeypatch.setenv("HISPARSE_COL_SLICES", str(slices))
    csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=nnz, b=0.5, c=2.0, seed=12)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    assert cp.num_rows * cp.num_cols / cp.nnz > 20000
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 12, impl))
    auto = build(cp, impl, wgs)
    assert auto["format"] in ("owner", "owner24")      # the 7-byte records unless the 11-bit row cap makes them larger than 8-byte chunks
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "owner24")
    t = build(cp, impl, wgs)
    assert t["format"] == "owner24" and t["nnz"] == cp.nnz and t["elements"] >= cp.nnz       # 24-bit position words: shares <= 2047 rows
    # whole records of four steps per wavefront and block: 7 bytes per slot + at most 3 dead steps per (block, wavefront)
    assert len(t["image"]) % 1792 == 0 and t["elements"] * 7 <= len(t["image"]) <= t["elements"] * 7 + len(t["blocks"]) * 14 * 3 * 448
    if auto["format"] == "owner24":
        assert np.array_equal(auto["image"], t["image"])
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "owner")
    t32 = build(cp, impl, wgs)
    assert t32["format"] == "owner" and len(t32["image"]) == t32["elements"] * 8
    assert cases.float_close(tile_emulator.run(t32, impl, xw, cp.num_rows), oracle_y(cp, impl, xw))
    assert t["ring_buffers"] in (2, 3, 4) and (not slices or t["col_slices"] == slices)
    assert (t["max_block_rows"] + 14) * 4 + t["ring_buffers"] * 32768 <= 160 * 1024
    got = tile_emulator.run(t, impl, xw, cp.num_rows)
    assert cases.float_close(got, oracle_y(cp, impl, xw))
    # fixed point: the 8-byte OWNER form is float only, OWNER24 takes hyper-sparse fixed-point matrices too (saturating 32-bit sums)
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "owner")
    csr0 = host.CSRMatrix.generate("powerlaw", 9000, 70000, a=20000, b=0.5, c=1.0, seed=12)
    cp0 = host.format_matrix(csr0, 0, skip_empty_rows=True)
    assert build(cp0, 0, 4)["format"] in ("pairs", "pairs24")
    for forced in ("owner24", None):
        if forced:
            monkeypatch.setenv("HISPARSE_STREAM_FORMAT", forced)
        else:
            monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
        t0 = build(cp0, 0, 4)
        assert t0["format"] == "owner24"
        x0 = host.pack_vector(0, cases.random_x(cp0.num_cols, 12, 0) * 40.0)      # large x: some rows saturate
        want0 = oracle_y(cp0, 0, x0)
        assert np.array_equal(tile_emulator.run(t0, 0, x0, cp0.num_rows), want0)


def test_empty_environment_switches_count_as_unset(monkeypatch):
    """`HISPARSE_MAX_ROWS=` (set but empty) once meant "one row per block": 4.9 M blocks for ogbn-products.  Empty = not set."""
    for name in ("HISPARSE_STREAM_FORMAT", "HISPARSE_MAX_ROWS", "HISPARSE_COL_SLICES", "HISPARSE_ROW_RUNS", "HISPARSE_AUX_BITS"):
        monkeypatch.delenv(name, raising=False)
    csr = host.CSRMatrix.generate("powerlaw", 20000, 30000, a=200000, b=0.3, c=1.0, seed=4)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    want = build(cp, 0, 32)
    for name in ("HISPARSE_STREAM_FORMAT", "HISPARSE_MAX_ROWS", "HISPARSE_COL_SLICES", "HISPARSE_ROW_RUNS", "HISPARSE_AUX_BITS"):
        monkeypatch.setenv(name, "")
    got = build(cp, 0, 32)
    assert got["format"] == want["format"] and np.array_equal(got["image"], want["image"]) and got["blocks"].tobytes() == want["blocks"].tobytes()


def test_worker_pool(tmp_path):
    """parallel_for of the load-time builders runs on parked worker threads (tiles_common.h: WorkerPool): tests/cpp/test_worker_pool.cpp."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "worker_pool"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", f"-I{root}/include", f"-I{root}/hisparse_amd/csrc", f"{root}/tests/cpp/test_worker_pool.cpp", "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "WORKER POOL OK" in out.stdout, out.stdout + out.stderr


@pytest.mark.parametrize("impl,ob", [(0, 8), (2, 64)])
def test_blocks_go_to_xcds_by_column_slice(impl, ob, monkeypatch):
    """Column-sliced matrices whose x outgrows an XCD's L2 (forced here: HISPARSE_XCD_AFFINITY=1): logical workgroups
    [x * G/8, (x+1) * G/8) -- XCD x -- hold blocks of at most two column slices, every workgroup has work, every block is run
    once, the chains stay ordered by row partition (hs_run_partition), and y is what the oracle says."""
    if impl == 2 and os.environ.get("HISPARSE_STREAM_FORMAT") in ("pairs", "pairs24", "delta", "bitmap"):
        monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "owner24")
    if os.environ.get("HISPARSE_STREAM_FORMAT") == "bitmap":
        pytest.skip("BITMAP images have their own builder")
    monkeypatch.setenv("HISPARSE_COL_SLICES", "5")
    monkeypatch.setenv("HISPARSE_MAX_ROWS", "500")
    monkeypatch.setenv("HISPARSE_XCD_AFFINITY", "1")
    csr = host.CSRMatrix.generate("powerlaw", 30000, 90000, a=400000, b=0.4, c=1.0 if impl == 0 else 2.0, seed=3)
    cp = host.format_matrix(csr, impl, ob_bank=ob, skip_empty_rows=True)      # several row partitions
    assert cp.num_row_partitions > 1
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl))
    G = 16
    t = build(cp, impl, G)
    assert t["col_slices"] == 5 and t["num_workgroups"] == G and len(t["blocks"]) >= G
    slice_of = t["blocks"]["out_offset"] // cp.num_rows
    for xcd in range(8):
        seen = set()
        for g in range(xcd * G // 8, (xcd + 1) * G // 8):
            mine = t["block_order"][t["wg_first"][g]: t["wg_first"][g + 1]]
            assert len(mine) >= 1
            parts = t["blocks"]["row_part"][mine]
            assert (np.diff(parts.astype(np.int64)) >= 0).all()
            seen |= set(int(s) for s in slice_of[mine])
        assert len(seen) <= 2, seen
    want = oracle_y(cp, impl, xw)
    got = tile_emulator.run(t, impl, xw, cp.num_rows)
    assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
    y = np.zeros(cp.num_rows, dtype=np.uint32)
    for j in range(cp.num_row_partitions):      # the reference's launch loop: one row partition at a time
        y = tile_emulator.run(t, impl, xw, cp.num_rows, row_part_filter=j, y_init=y, rows_per_part=128 * cp.ob_bank)
    assert np.array_equal(y, want) if impl == 0 else cases.float_close(y, want)


def test_plans_of_small_and_narrow_matrices(monkeypatch):
    """Round-3 planner rules (stream_tiles.cpp, "tile plan"), each on the shape it was measured on, emulated against the oracle:
    a few-row PAIRS block puts many lanes of a step into one row (LDS atomics collide) -> one slice per sub-tile; a matrix of at most 16
    sub-tiles may take any slice count, preferably one that divides the sub-tiles; unsliced plans pay per unit boundary; DELTA's fixed
    cost shrinks with the units per block."""
    for k in ("HISPARSE_STREAM_FORMAT", "HISPARSE_COL_SLICES", "HISPARSE_MAX_ROWS"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("HISPARSE_LIGHT", "0")      # the second case (1.4 M non-zeros) would take the LIGHT plan: these are the row-block kernel's plans
    # one rank's slab of mouse_gene split 4 ways: 6 live sub-tiles -> 6 slices (15-27 us in one slice, 13.3-13.9 in six)
    csr = host.CSRMatrix.generate("powerlaw", 11264, 45101, a=7.2e6, b=0.30, c=0.1, seed=44)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    t = build(cp, 0, 256)
    assert t["col_slices"] == 6 and len(t["blocks"]) == 252 and len(t["units"]) == 252
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 3, 0))
    assert np.array_equal(tile_emulator.run(t, 0, xw, cp.num_rows), oracle_y(cp, 0, xw))
    # 14 sub-tiles (gplus's shape, a tenth of its non-zeros): 7 slices = two sub-tiles each, not 6 or 8
    csr = host.CSRMatrix.generate("powerlaw", 107614, 107614, a=1.4e6, b=0.35, c=1.0, seed=7)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    t = build(cp, 0, 256)
    assert t["col_slices"] == 7
    per_block = t["blocks"]["unit_end"] - t["blocks"]["unit_begin"]
    assert per_block.max() <= 2
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 4, 0))
    assert np.array_equal(tile_emulator.run(t, 0, xw, cp.num_rows), oracle_y(cp, 0, xw))
    # a wide matrix: through round 4 only the power-of-two slice counts (HISPARSE_POW2_SLICES=1 brings the rule back); since round 5 whatever
    # the cost model likes (ogbl-ppa and its R-MAT stand-in run 3-5 % faster in 5 slices than in 4: profiles/r05_any_slice_count.txt)
    csr = host.CSRMatrix.generate("powerlaw", 60000, 300000, a=3.0e6, b=0.3, c=1.0, seed=9)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    t = build(cp, 0, 256)
    assert 1 <= t["col_slices"] <= 8
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 5, 0))
    assert np.array_equal(tile_emulator.run(t, 0, xw, cp.num_rows), oracle_y(cp, 0, xw))
    monkeypatch.setenv("HISPARSE_POW2_SLICES", "1")
    assert build(cp, 0, 256)["col_slices"] in (1, 2, 4, 8)
    monkeypatch.delenv("HISPARSE_POW2_SLICES")
    # round 4: without the switch the 1.4 M non-zero matrix takes the LIGHT plan -- one slice, up to 4 blocks per CU, strided dealing,
    # the image an ordinary PAIRS image (the emulator runs it with the row-block kernel's dealing: same sums)
    monkeypatch.delenv("HISPARSE_LIGHT")
    csr = host.CSRMatrix.generate("powerlaw", 107614, 107614, a=1.4e6, b=0.35, c=1.0, seed=7)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    t = build(cp, 0, 256)
    assert t["format"] == "pairs" and t["col_slices"] == 1 and 256 < len(t["blocks"]) <= 1024 and t["max_block_rows"] <= 3071
    assert not (t["blocks"]["flags"] & 1).any()
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 4, 0))
    assert np.array_equal(tile_emulator.run(t, 0, xw, cp.num_rows), oracle_y(cp, 0, xw))


def test_small_fixed_point_dense_row_layers_take_the_sliced_delta_plan(monkeypatch):
    """Round 5 planner rule (stream_tiles.cpp): a pruned-NN layer of 10-30 % density in FIXED point -- 1 ... 5.5 M non-zeros, 2 ... 8 x
    sub-tiles -- runs as DELTA with one column slice per sub-tile (combine carried into the next step's kernel: one launch), where the
    fitted cost beats BITMAP's and LIGHT's; the float modes keep BITMAP.  The emulated kernel must still match the oracle."""
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
    for name in ("HISPARSE_LIGHT", "HISPARSE_COL_SLICES", "HISPARSE_ROW_RUNS"):
        monkeypatch.delenv(name, raising=False)
    csr = host.CSRMatrix.generate("bernoulli", 512, 33288, b=0.2, c=0.05, seed=80)      # transformer-80's shape and density
    for impl, want_format, want_slices in ((0, "delta", 5), (1, "bitmap", 1)):
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
        t = build(cp, impl, 256)
        assert t["format"] == want_format and t["col_slices"] == want_slices, (impl, t["format"], t["col_slices"])
        if impl == 0:
            assert (t["blocks"]["flags"] & 1).all()                     # per-lane row sums (kBlockDenseRows) in every block
            assert len(t["image"]) < 48 << 20                           # below the carried-combine limit (stream_tiles.h: kCarryMaxImageBytes)
            units_per_block = (t["blocks"]["unit_end"] - t["blocks"]["unit_begin"]).max()
            assert units_per_block == 1                                 # a slice per sub-tile: no x refills, no unit barriers
            xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 80, impl))
            got = tile_emulator.run(t, impl, xw, cp.num_rows)
            assert np.array_equal(got, oracle_y(cp, impl, xw))
    # below a million non-zeros the measured range ends: LIGHT, as before
    small = host.CSRMatrix.generate("bernoulli", 512, 33288, b=0.05, c=0.05, seed=95)
    t = build(host.format_matrix(small, 0, skip_empty_rows=True), 0, 256)
    assert t["format"] == "pairs" and t["col_slices"] == 1
