"""GPU parity: libhisparse_hip.so (through the C-ABI) against the oracle on the same seeded inputs.

Bit-exact for the fixed-point mode; float modes within 1e-4 relative + 1e-4 absolute (north_star:
1e-4 rel-err; the reference's own verify uses 1e-4 absolute, spmv_csim/csim.cpp:162,172).
"""
import os

import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases

pytestmark = pytest.mark.gpu

IMPLS = [0, 1, 2]


@pytest.fixture(autouse=True, params=["pairs", "delta", "delta-lane-sums", "delta-no-lane-sums", "bitmap", "owner", "pairs24", "owner24", "light", "sweep"])
def stream_format(request, monkeypatch):
    # every parity test runs once per device stream format (hisparse_amd/csrc/stream_tiles.h); DELTA additionally with the
    # per-lane register sums of long-row blocks forced on and off (by default the block's density decides); BITMAP (normally
    # chosen for dense rows only) forced onto every matrix small enough for a mask per 64 columns of every row
    # "pairs24": the opt-in 7-byte form of PAIRS (HISPARSE_AUX_BITS=24), taken where the row counts allow; "owner24": OWNER in records of
    # four steps with 24-bit position words (the default for hyper-sparse float matrices)
    # "light" (round 4): the small-matrix plan -- the PAIRS image cut into up to 4 x CUs blocks and run by spmv_light_kernel (one launch,
    # 256-thread workgroups, x gathered from L2); taken by every case of at most 16 x sub-tiles, plain PAIRS otherwise
    # "sweep" (round 4): column-ordered blocks, x gathered from L2, no units (spmv_sweep.hip)
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "pairs" if request.param in ("pairs24", "light") else request.param.split("-")[0])
    monkeypatch.setenv("HISPARSE_LIGHT", "1" if request.param == "light" else "0")
    if request.param == "pairs24":
        monkeypatch.setenv("HISPARSE_AUX_BITS", "24")
    if request.param.endswith("-lane-sums"):
        monkeypatch.setenv("HISPARSE_ROW_RUNS", "0" if "-no-" in request.param else "1")
    return request.param


def _run_case(impl, m, vb, ob, skip, seed):
    csr, cp = cases.formatted(m, impl, vb, ob, skip)
    x = cases.random_x(cp.num_cols, seed, impl)
    xw = host.pack_vector(impl, x)
    chans = [cp.channel(c) for c in range(16)]
    want = orc.spmv(impl, chans, xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    eng = device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank)
    eng.load_matrix(cp)
    eng.load_vector(xw)
    eng.run()
    got = eng.read_result()
    # a second run must give the same answer (accumulators re-armed by the finalize pass / memset)
    eng.run()
    again = eng.read_result()
    stats = eng.stats()
    eng.close()
    assert stats["nnz"] == m.nnz
    sub_tiles = cp.num_col_partitions * max(1, -(-(8 * cp.vb_bank) // 8192))      # column partitions x sub-tiles of 8192 columns per partition
    sliced = os.environ.get("HISPARSE_COL_SLICES", "1") not in ("", "1")      # a forced sliced plan is the row-block kernel's
    assert stats["light_kernel"] == (1 if os.environ.get("HISPARSE_LIGHT") == "1" and sub_tiles <= 16 and m.nnz > 0 and not sliced else 0)
    forced = os.environ["HISPARSE_STREAM_FORMAT"]
    if forced == "owner" and impl == 0:
        forced = "pairs"                 # the 8-byte OWNER form is float only (fixed point: OWNER24 with saturating 32-bit accumulators)
    if forced != "bitmap" or cp.num_rows * ((cp.num_cols + 63) // 64) * 8 <= (1 << 30):    # a forced bitmap gives way above 1 GiB of masks
        got_format = device.STREAM_FORMATS[stats["stream_format"]]
        assert got_format == forced or (os.environ.get("HISPARSE_AUX_BITS") == "24" and got_format == forced + "24")
    if impl == 0:
        assert np.array_equal(got, want), f"fixed-point mismatch at {np.nonzero(got != want)[0][:8]}"
        assert np.array_equal(again, want)
    else:
        assert cases.float_close(got, want)
        assert cases.float_close(again, want)
    return cp, got, want


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("skip", [False, True])
def test_multi_partition_small_banks(impl, skip):
    # tiny banks: many row/column partitions, header/start offsets, last-partition part_len, F=8 interleave
    ob = 8 if impl == 2 else 1
    m = cases.random_csr(2500, 300, 0.03, 11, impl)
    cp, _, _ = _run_case(impl, m, vb=4, ob=ob, skip=skip, seed=11)
    assert cp.num_col_partitions > 1 and cp.num_row_partitions > 1


@pytest.mark.parametrize("impl", IMPLS)
def test_baseline_config_1k(impl):
    # BASELINE.json configs[0]: 1k x 1k, 1 % dense, default banks => one partition
    m = cases.random_csr(1000, 1000, 0.01, 1, impl)
    v, o = host.default_banks(impl)
    cp, _, _ = _run_case(impl, m, vb=v, ob=o, skip=True, seed=1)
    assert cp.num_partitions == 1


@pytest.mark.parametrize("impl", IMPLS)
def test_medium_default_banks(impl):
    # 40k x 70k power-law stand-in, three column partitions at the default 32768-column tile, ~1.4 M non-zeros
    csr = host.CSRMatrix.generate("powerlaw", 40000, 70000, a=1.4e6, b=0.35, c=1.0 if impl == 0 else 2.0, seed=7)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)   # signed values for the float modes
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(40000, 70000))
    v, o = host.default_banks(impl)
    cp, _, _ = _run_case(impl, m, vb=v, ob=o, skip=True, seed=7)
    assert cp.num_col_partitions == 3


@pytest.mark.parametrize("impl", IMPLS)
def test_dense_row_blocks(impl):
    # pruned-NN shape: few rows, thousands of non-zeros each => row blocks of 1-2 rows, the dense-row (wave-combine) layout
    csr = host.CSRMatrix.generate("bernoulli", 64, 20000, b=0.3, c=0.05, seed=3)
    ip, ix, dv = csr.arrays()
    if impl == 0:
        dv = np.abs(dv)
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(64, 20000))
    v, o = host.default_banks(impl)
    _run_case(impl, m, vb=v, ob=o, skip=True, seed=13)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("slices", [2, 4, 8])
def test_column_slices(impl, slices, monkeypatch):
    # force the 2-D (row range x column slice) decomposition + combine pass on a matrix small enough for the oracle
    monkeypatch.setenv("HISPARSE_COL_SLICES", str(slices))
    csr = host.CSRMatrix.generate("powerlaw", 30000, 70000, a=900000, b=0.4, c=1.0 if impl == 0 else 2.0, seed=17)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(30000, 70000))
    v, o = host.default_banks(impl)
    _run_case(impl, m, vb=v, ob=o, skip=True, seed=17)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("slices", [12, 16])
def test_column_slices_beyond_eight(impl, slices, monkeypatch):
    # a forced plan of more than eight column slices (stream_tiles.h: kMaxForcedColSlices; the planner itself stops at eight for row-block
    # images): 18 sub-tiles of x, the combine pass instantiated for up to 16 partial vectors
    monkeypatch.setenv("HISPARSE_COL_SLICES", str(slices))
    csr = host.CSRMatrix.generate("powerlaw", 20000, 140000, a=1200000, b=0.4, c=1.0 if impl == 0 else 2.0, seed=19)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(20000, 140000))
    v, o = host.default_banks(impl)
    _run_case(impl, m, vb=v, ob=o, skip=True, seed=19)


def test_repeated_runs_stay_exact():
    # soak: the hand-counted waits of the stream ring must hold on every launch, not just most (a compiler-inserted
    # register copy ahead of a wait once made one record in a few million wrong, on some launches only)
    csr = host.CSRMatrix.generate("powerlaw", 150000, 150000, a=6000000, b=0.35, c=1.0, seed=23)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    xw = host.pack_vector(0, cases.random_x(cp.num_cols, 23, 0))
    want = orc.spmv(0, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                    cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    eng = device.SpmvEngine(0)
    eng.load_matrix(cp)
    eng.load_vector(xw)
    for i in range(40):
        eng.run()
        got = eng.read_result()
        assert np.array_equal(got, want), f"run {i}: {int((got != want).sum())} rows differ"
    eng.close()


@pytest.mark.parametrize("impl", [0, 2])
def test_heavy_rows_inside_a_sparse_matrix(impl):
    # power-law head: a few blocks of very long rows (DELTA: per-lane register sums; PAIRS: dense-row chunks where rows <= 32)
    # next to a sparse bulk, in one matrix
    csr = host.CSRMatrix.generate("powerlaw", 30000, 60000, a=600000, b=0.8, c=1.0 if impl == 0 else 2.0, seed=3)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(30000, 60000))
    v, o = host.default_banks(impl)
    _run_case(impl, m, vb=v, ob=o, skip=True, seed=29)


@pytest.mark.parametrize("impl", IMPLS)
def test_hyper_sparse_bridges(impl, stream_format):
    # most position gaps exceed 16 bits: DELTA needs bridge slots (fixed point: zero-value elements that advance 65535;
    # float: dead slots whose row is clamped), PAIRS just pads; rows near the end of tall blocks
    csr = host.CSRMatrix.generate("powerlaw", 30000, 40000, a=15000, b=0.0, c=1.0 if impl == 0 else 2.0, seed=8)
    ip, ix, dv = csr.arrays()
    if impl != 0:
        dv = (dv - 1.0).astype(np.float32)
    import scipy.sparse as sp
    m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(30000, 40000))
    v, o = host.default_banks(impl)
    _run_case(impl, m, vb=v, ob=o, skip=True, seed=8)


@pytest.mark.parametrize("impl", [1, 2])
def test_conflicting_lds_atomics_are_complete_before_the_store(impl, monkeypatch):
    # regression (found by tests/gpu_fuzz_soak.py): a 15 %-dense float matrix in the DELTA format WITHOUT per-lane sums makes
    # many lanes of one ds_add_f64 hit the same accumulator; such no-return atomics were still queued when the accumulators
    # were stored, and ~40 % of the launches lost one record's worth of products.  Every launch must be right.
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "delta")
    monkeypatch.setenv("HISPARSE_ROW_RUNS", "0")
    m = cases.random_csr(3004, 1423, 0.15, 543808340, impl)
    _, cp = cases.formatted(m, impl, 64, 16 if impl == 2 else 2, False)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 543808340, impl))
    want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)
    eng = device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank)
    eng.load_matrix(cp)
    eng.load_vector(xw)
    for i in range(40):
        eng.run()
        assert cases.float_close(eng.read_result(), want), f"launch {i}"
    eng.close()


def test_context_reuse():
    # one context, several matrices in a row (column-sliced and not, different sizes): no state of the previous matrix survives,
    # and a vector left over from a differently sized matrix is refused instead of being read out of bounds
    eng = device.SpmvEngine(0)
    last_cols = None
    for k, (rows, cols, nnz) in enumerate([(30000, 70000, 900000), (900, 500, 9000), (60000, 40000, 400000)]):
        csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=nnz, b=0.4, c=1.0, seed=50 + k)
        cp = host.format_matrix(csr, 0, skip_empty_rows=True)
        xw = host.pack_vector(0, cases.random_x(cp.num_cols, 60 + k, 0))
        eng.load_matrix(cp)
        if last_cols is not None and last_cols != cp.num_cols:
            with pytest.raises(device.DeviceError):
                eng.run()
        eng.load_vector(xw)
        eng.run()
        want = orc.spmv(0, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                        cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
        assert np.array_equal(eng.read_result(), want)
        last_cols = cp.num_cols
    eng.close()


def test_random_shapes_and_banks():
    # seeded fuzz over shapes, densities, bank sizes (hence partition counts), skip_empty_rows and numeric modes
    rng = np.random.default_rng(20240607)
    for case in range(36):
        impl = int(rng.integers(0, 3))
        rows = int(rng.integers(1, 6000))
        cols = int(rng.integers(1, 6000))
        density = float(rng.choice([0.0005, 0.003, 0.02, 0.15]))
        vb = int(rng.choice([1, 2, 16, 64, 4096]))
        ob = int(rng.choice([1, 2, 8, 64])) * (8 if impl == 2 else 1)
        skip = bool(rng.integers(0, 2))
        try:
            _run_case(impl, cases.random_csr(rows, cols, density, 100 + case, impl), vb=vb, ob=ob, skip=skip, seed=200 + case)
        except AssertionError as e:
            raise AssertionError(f"case {case}: impl {impl} {rows}x{cols} density {density} vb {vb} ob {ob} skip {skip}: {e}")


@pytest.mark.parametrize("impl", IMPLS)
def test_tiny_matrix(impl):
    # wavefronts with fewer records than the pipeline depth, and wavefronts with none at all
    v, o = host.default_banks(impl)
    _run_case(impl, cases.random_csr(300, 200, 0.05, 3, impl), vb=v, ob=o, skip=True, seed=3)


def test_fixed_rounding_and_saturation():
    # values/x chosen so products need AP_RND and rows overflow AP_SAT (sum >= 256 => 0xffffffff)
    rng = np.random.default_rng(3)
    import scipy.sparse as sp
    rows, cols = 256, 512
    dense = np.zeros((rows, cols), dtype=np.float32)
    dense[0, :] = 200.0                      # single products 200*3 saturate, so does the row
    dense[1, :300] = 1.0                     # 300 * x(=1..) saturates the running sum
    dense[2, ::7] = rng.uniform(0, 1, len(range(0, cols, 7))).astype(np.float32) * 2.0 ** -20  # tiny: rounding to 0/1 LSB
    dense[3:, :] = (rng.uniform(0, 1, (rows - 3, cols)) < 0.05) * rng.uniform(0, 1.5, (rows - 3, cols))
    m = sp.csr_matrix(dense.astype(np.float32))
    _, got, want = _run_case(0, m, vb=4096, ob=8192, skip=True, seed=3)
    assert got[0] == 0xFFFFFFFF and want[0] == 0xFFFFFFFF


def test_run_partition_matches_run():
    impl = 0
    m = cases.random_csr(2500, 300, 0.03, 21, impl)
    csr, cp = cases.formatted(m, impl, 4, 1, True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 21, impl))
    eng = device.SpmvEngine(impl, ob_bank=1, vb_bank=4)
    eng.load_matrix(cp)
    eng.load_vector(xw)
    eng.run()
    full = eng.read_result()
    # re-load to reset y, then go partition by partition with the reference's scalars
    eng.load_matrix(cp)
    for j in range(cp.num_row_partitions):
        eng.run_partition(j, cp.part_len(j))
    stepped = eng.read_result()
    with pytest.raises(device.DeviceError):
        eng.run_partition(0, cp.part_len(0) + 8)
    eng.close()
    assert np.array_equal(full, stepped)


def test_run_partition_with_column_slices(monkeypatch):
    # per-partition stepping must also combine only that partition's rows when the matrix is column-sliced
    monkeypatch.setenv("HISPARSE_COL_SLICES", "2")
    impl = 1
    m = cases.random_csr(2500, 300, 0.03, 23, impl)
    csr, cp = cases.formatted(m, impl, 4, 1, True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 23, impl))
    want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, 1, 4)
    eng = device.SpmvEngine(impl, ob_bank=1, vb_bank=4)
    eng.load_matrix(cp)
    assert eng.stats()["col_slices"] == 2
    eng.load_vector(xw)
    for j in reversed(range(cp.num_row_partitions)):
        eng.run_partition(j, cp.part_len(j))
    stepped = eng.read_result()
    eng.run()
    full = eng.read_result()
    eng.close()
    assert cases.float_close(stepped, want) and cases.float_close(full, want)


def test_empty_rows_and_empty_matrix_rows():
    # rows with no entries at all, an all-empty column partition, and skip counts > 1
    import scipy.sparse as sp
    rows, cols = 1024, 96
    dense = np.zeros((rows, cols), dtype=np.float32)
    dense[5, 3] = 1.5
    dense[5 + 128 * 4, 40] = 0.25          # same lane stream, three empty rounds in between
    dense[1000, 95] = 2.0
    m = sp.csr_matrix(dense)
    for impl in IMPLS:
        _run_case(impl, m, vb=4, ob=8, skip=True, seed=5)
        _run_case(impl, m, vb=4, ob=8, skip=False, seed=5)


def test_marker_skip_count_beyond_255():
    # fixed mode, >= 256 skipped rounds: the product's formatter chains markers (tests/test_tiles_cpu.py), rows land where they belong
    from test_tiles_cpu import marker_limit_matrix
    m, _ = marker_limit_matrix()
    for impl in (0, 2):
        v, o = host.default_banks(impl)
        _run_case(impl, m, vb=v, ob=o, skip=True, seed=1)


def test_csim_large_sparse_known_answer():
    # spmv_csim/csim.cpp:468-479 at full size: uniform 100 000 x 100 000, 10 ones per row, x = glibc rand() % 2 after the
    # 128 + 1024 draws of the two earlier cases; expected y = exact integer row sums (compute_ref) -- all three modes
    import ctypes
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    for _ in range(128 + 1024):
        libc.rand()
    draws = np.array([libc.rand() % 2 for _ in range(100000)], dtype=np.float32)
    for impl in IMPLS:
        csr = host.CSRMatrix.generate("uniform", 100000, 100000, a=10)
        cp = host.format_matrix(csr, impl, skip_empty_rows=False)
        x = np.zeros(cp.num_cols, dtype=np.float32)
        x[:100000] = draws
        eng = device.SpmvEngine(impl)
        eng.load_matrix(cp)
        eng.load_vector(host.pack_vector(impl, x))
        eng.run()
        y = host.unpack_result(impl, eng.read_result())
        eng.close()
        ip, ix, dv = csr.arrays()
        ref = orc.compute_ref(cp.num_rows, ip, ix, dv, x)
        assert orc.verify(ref, y) == -1 and np.array_equal(y, ref)


def _feedback_reference(impl, y_words, x_words, scale, shift):
    n = min(len(y_words), len(x_words))
    out = x_words.copy()
    if impl == 0:
        wide = y_words[:n].astype(object) * int(scale) + (1 << 23)          # exact integers
        prod = np.array([min(int(w) >> 24, 0xFFFFFFFF) for w in wide], dtype=object)
        out[:n] = np.array([min(int(p) + int(shift), 0xFFFFFFFF) for p in prod], dtype=np.uint32)
    else:
        s, b = np.uint32(scale).view(np.float32), np.uint32(shift).view(np.float32)
        out[:n] = ((s * y_words[:n].view(np.float32)).astype(np.float32) + b).astype(np.float32).view(np.uint32)
    return out


@pytest.mark.parametrize("slices", ["1", "2"])
@pytest.mark.parametrize("graph", ["0", "1"])
@pytest.mark.parametrize("impl", IMPLS)
def test_pagerank_iterations(impl, graph, slices, monkeypatch):
    monkeypatch.setenv("HISPARSE_ITERATE_GRAPH", graph)   # plain launches / captured hipGraphs of 32 iterations
    monkeypatch.setenv("HISPARSE_COL_SLICES", slices)     # 2: the feedback rides in the slice-combine launch
    # iterative caller (hisparse_hip.h extension): PageRank over util_normalize_csr_matrix_by_outdegree, x fed back on
    # the device through one replayed hipGraph, against the same loop on the CPU (oracle SpMV + exact update arithmetic)
    n, iters, damping = 20000, 40, 0.85
    csr = host.CSRMatrix.generate("powerlaw", n, n, a=300000, b=0.3, c=1.0, seed=41)
    csr.normalize_by_outdegree()
    v, o = host.default_banks(impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    scale = int(host.pack_vector(impl, np.array([damping], dtype=np.float32))[0])
    shift = int(host.pack_vector(impl, np.array([(1.0 - damping) / n], dtype=np.float32))[0])
    x0 = host.pack_vector(impl, np.full(cp.num_cols, 1.0 / n, dtype=np.float32))
    chans = [cp.channel(c) for c in range(16)]
    x, y = x0.copy(), None
    for _ in range(iters):
        y = orc.spmv(impl, chans, x, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
        x = _feedback_reference(impl, y, x, scale, shift)
    eng = device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank)
    eng.load_matrix(cp)
    eng.load_vector(x0)
    eng.iterate(iters, scale, shift)
    got_y = eng.read_result()
    # one more explicit step through the non-graph entry points must continue the same sequence
    eng.run()
    eng.feedback(scale, shift)
    got_y2 = eng.read_result()
    eng.close()
    y2 = orc.spmv(impl, chans, x, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    if impl == 0:
        assert np.array_equal(got_y, y) and np.array_equal(got_y2, y2)
        assert got_y[:n].astype(np.float64).sum() / 2 ** 24 > 0.5       # ranks still carry mass (not all rounded away)
    else:
        assert cases.float_close(got_y, y) and cases.float_close(got_y2, y2)


@pytest.mark.parametrize("name", ["ogbl_ppa", "transformer_50", "ogbn_products", "mouse_gene"])
def test_full_size_exact_known_answer(name, stream_format, monkeypatch):
    # BASELINE.json's configurations at FULL size (stand-in generators of hisparse_amd/datasets.py), with inputs that make the
    # answer independent of summation order and rounding, so it must match an integer CSR product bit for bit in every
    # numeric mode: all matrix values 2^-10, x in {0, 1, 2, 3}  =>  y[r] = 2^-10 * (sum of the selected x), exact in Q8.24
    # (no product needs rounding, no row reaches 256) and in fp32 (every partial sum is a multiple of 2^-10 below 2^14).
    if stream_format != "pairs":
        pytest.skip("one pass over the big matrices, in the format the library picks by itself")
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    monkeypatch.delenv("HISPARSE_AUX_BITS", raising=False)
    import scipy.sparse as sp
    from hisparse_amd import datasets
    cfg, csr = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    csr.fill(2.0 ** -10)
    ip, ix, _ = csr.arrays()
    rows, cols = csr.num_rows, csr.num_cols
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    x_int = np.random.default_rng(77).integers(0, 4, cp.num_cols).astype(np.int64)
    pattern = sp.csr_matrix((np.ones(len(ix), dtype=np.int64), ix.astype(np.int64), ip.astype(np.int64)), shape=(rows, cols))
    sums = pattern @ x_int[:cols]                                    # exact integers
    assert sums.max() < (1 << 17)                                    # < 256 * 2^10 / ... : far from saturation and from 2^24
    want = np.zeros(cp.num_rows, dtype=np.uint32)
    if impl == 0:
        want[:rows] = (sums << 14).astype(np.uint32)                 # 2^-10 * s in Q8.24 = s * 2^14 LSB
    else:
        want[:rows] = (sums.astype(np.float64) / 1024.0).astype(np.float32).view(np.uint32)
    eng = device.SpmvEngine(impl)
    eng.load_matrix(cp)
    eng.load_vector(host.pack_vector(impl, x_int.astype(np.float32)))
    eng.run()
    got = eng.read_result()
    # the same through the reference's partition-by-partition launch sequence
    for j in range(cp.num_row_partitions):
        eng.run_partition(j, cp.part_len(j))
    again = eng.read_result()
    stats = eng.stats()
    eng.close()
    assert stats["nnz"] == len(ix)
    assert np.array_equal(got, want), f"{int((got != want).sum())} rows differ"
    assert np.array_equal(again, want)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("rows,cols,density", [(512, 33288, 0.5), (40, 9000, 0.3), (3000, 2500, 0.2), (129, 2049, 0.9), (20000, 4096, 0.15)])
def test_dense_rows_pick_bitmap(impl, rows, cols, density, stream_format, monkeypatch):
    """Pruned-NN shaped matrices (sw/bm.sh:21-27) choose the BITMAP format unforced: whole-run, partition-by-partition and
    repeated launches against the oracle."""
    if stream_format != "pairs":
        pytest.skip("format chosen by the library here; one pass")
    monkeypatch.delenv("HISPARSE_STREAM_FORMAT")
    monkeypatch.delenv("HISPARSE_AUX_BITS", raising=False)
    m = cases.random_csr(rows, cols, density, 17, impl)
    if impl != 0:
        m.data *= np.float32(0.05)       # pruned-NN weights (datasets.py: N(0, 0.05)): the ORACLE's fp32 running sum of 16 K unit-sized
                                         # terms would itself be further than 1e-4 from the exact product
    v, o = host.default_banks(impl)
    if rows == 20000:
        o = 16 if impl != 2 else 16      # several row partitions: 20000 rows over LOGICAL_OB = 2048
    csr, cp = cases.formatted(m, impl, v, o, True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 17, impl))
    want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)
    with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
        eng.load_matrix(cp)
        eng.load_vector(xw)
        st = eng.stats()
        assert st["nnz"] == m.nnz
        # rows padded to 1024 (float_stall) can leave so many empty rows that their masks outweigh the saving: element streams then
        expect_bitmap = cp.num_rows * ((cp.num_cols + 63) // 64) * 8 <= 2 * m.nnz
        # round 6: a FIXED-point dense layer over ONE x sub-tile of >= 2^20 non-zeros is a one-slice DELTA image where the fitted costs say so
        # (stream_tiles.cpp: sliced_delta_possible; 2048 x 8192 at 15 %: 6.7 us against 11.6 us as a BITMAP image)
        if impl == 0 and (rows, cols) == (3000, 2500):
            assert device.STREAM_FORMATS[st["stream_format"]] == "delta" and st["col_slices"] == 1
            expect_bitmap = False
        assert (device.STREAM_FORMATS[st["stream_format"]] == "bitmap") == expect_bitmap
        if expect_bitmap:
            assert st["stream_bytes"] < 0.8 * 8 * m.nnz + (1 << 20)
        results = []
        for _ in range(3):
            eng.run()
            results.append(eng.read_result())
        for j in range(cp.num_row_partitions):
            eng.run_partition(j, cp.part_len(j))
        results.append(eng.read_result())
    for got in results:
        assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
