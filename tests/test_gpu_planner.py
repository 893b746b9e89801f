"""Out-of-sample check of the load-time planner (tools/planner_check.py holds the generators and the measurement):
over matrices from families none of the planner's constants was measured on -- banded, block-diagonal, R-MAT with other quadrant
probabilities, wide bipartite, tall-narrow, Erdos-Renyi, dense-row layers of other shapes, 2- / 4- / 8-way row slabs -- the plan the library
takes by itself must run within 10 % of the best plan that can be FORCED (every format, the planner's format at half / twice its slices),
and every forced plan must give the planner's y.  The reference's analogue is its design-space sweep
(performance_model/design_space_exp.cpp:496-547).

The full list (24 matrices; `python tools/planner_check.py`, profiles/r06_planner_check_after.txt): 23 within 10 %, median 1.016, one known miss asserted
here at its measured ratio + a margin so that it is SEEN, not hidden: a 60 %-dense float layer of one sub-tile (DELTA 1.14 x: the sliced-DELTA rule is
fixed-point only).  Before round 6 the same list read: 14 of 24 within 10 %, six between 2.1 x and 5.9 x (profiles/r06_planner_check_before.txt); what
closed the gap: the tile census (banded / block-diagonal matrices and their slabs), the SWEEP model fitted again, SWEEP for small wide matrices, per-lane
row sums for DELTA blocks with a hub row (a matrix whose 50 hub rows hold 52 % of the non-zeros: 89 -> 44.9 us), OWNER24 for one-slice float plans."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# (name, bound on planner time / best forced time).  A subset that builds in a few seconds each; the whole list: tools/planner_check.py.
WITHIN_10_PERCENT = ["banded_400k_d40_w2k", "blockdiag_200k_b512_p10", "rmat19_45_15_15", "bipartite_20k_x_2m_200", "bipartite_100k_x_4m_60", "tall_2m_x_50k_10",
                     "er_300k_30", "er_1500k_8", "dense_2048_x_8k_15", "dense_256_x_64k_8", "slab8_of_rmat19", "slab4_of_banded_400k", "slab8_of_er_300k",
                     "slab4_of_bipartite_100k", "slab8_of_tall_2m", "tall_3m_x_8k_6", "hubs_500k_15_plus_50x200k", "blockdiag_600k_b64_p50",
                     # from the SECOND list (tools/planner_check.py --second: written after the rules were final; 9 of 12 within 10 %, profiles/r06_planner_check_second_list.txt)
                     "wide_2k_x_8m_2000", "hubcols_500k_20_100x30", "stencil_1m_7x4"]
KNOWN_MISSES = {"dense_4096_x_4k_60": 1.25, "stencil_300k_3x20": 1.45,      # (second list: a fixed-point one-slice plan that OWNER24 would run 1.31 x faster; no rule separates it from the ones it would slow down)
                "banded_1m_d12_w50k": 1.16}      # (measured 1.02-1.09: OWNER24 in 4 slices where 2 would be 8 % faster -- inside 10 %, but too close to assert at 1.10)


@pytest.mark.parametrize("name", WITHIN_10_PERCENT + sorted(KNOWN_MISSES))
def test_planner_choice_against_the_best_forced_plan(name, record_property):
    import planner_check as pc
    case = next(c for c in pc.CASES + pc.SECOND if c[0] == name)
    res = pc.check(name, case[1], case[2](), steps=200, log=lambda s: None)
    record_property("planner", res["planner"])
    record_property("planner_us", res["planner_us"])
    record_property("best_forced", f"{res['best_forced']} {res['best_forced_us']} us")
    assert not res["wrong_results"], f"{name}: forced plans {res['wrong_results']} give another y than the planner's plan"
    timed = [v for v in res["variants"] if v["us"] is not None]
    assert len(timed) >= 3, f"{name}: only {len(timed)} forced plans loaded"
    bound = KNOWN_MISSES.get(name, 1.12)      # (the profile runs: every one of these <= 1.08; two points of margin for the box's noise on 8-30 us steps)
    assert res["planner_over_best"] <= bound, (f"{name}: planner {res['planner']} {res['planner_us']} us, best forced {res['best_forced']} {res['best_forced_us']} us "
                                               f"= {res['planner_over_best']:.3f} x (bound {bound})")


@pytest.mark.parametrize("name,impl_want", [("stencil_300k_3x20", "owner24"), ("rmat19_45_15_15", None), ("dense_2048_x_8k_15", None)])
def test_autotune_keeps_the_fastest_measured_plan(name, impl_want):
    """hs_set_option "autotune" = 1 (round 6, opt-in): the load measures the planner's own image against every other element format and keeps the fastest.  On the
    known close call of the second list (a fixed-point one-slice plan that OWNER24 runs 1.3 x faster) it must switch; elsewhere it must end within 3 % of the
    planner's plan or better; the result is the oracle's either way, and the caller's options are as they were."""
    import numpy as np
    import planner_check as pc
    from hisparse_amd import device, host
    from oracle import oracle as orc
    import cases
    case = next(c for c in pc.CASES + pc.SECOND if c[0] == name)
    impl, m = case[1], case[2]()
    csr = host.CSRMatrix.from_scipy(m)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 3, impl))
    want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    times = {}
    for tune in ("0", "1"):
        with device.SpmvEngine(impl) as eng:
            eng.set_option("autotune", tune)
            eng.set_option("light", "0")                     # a caller's own option must survive the tuning loads
            eng.load_matrix(cp)
            st = eng.stats()
            eng.load_vector(xw)
            eng.run()
            got = eng.read_result()
            assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
            for _ in range(300):
                eng.run()
            eng.sync()
            times[tune] = (min(eng.time_runs(5, 200, kernel=False)[0] / 200 for _ in range(3)) * 1e3, device.STREAM_FORMATS[st["stream_format"]])
            eng.set_option("autotune", "0")                  # ... and the tuning loads left no forced format behind: without the option the next load is
            eng.load_matrix(cp)                              # the planner's own plan again (two formats within 3 % of each other may swap between tuned loads)
            planner_fmt = device.STREAM_FORMATS[eng.stats()["stream_format"]]
            assert planner_fmt == times.get("0", (None, planner_fmt))[1]
    (plain_us, plain_fmt), (tuned_us, tuned_fmt) = times["0"], times["1"]
    if impl_want:
        assert tuned_fmt == impl_want and tuned_us < 0.9 * plain_us, times
    assert tuned_us <= 1.08 * plain_us, times      # (ties within 3 % go to the planner's plan; a wrong pick at that margin costs no more than this)


STRUCTURED = [("banded", lambda impl: __import__("planner_check").banded(150_000, 16, 800, 41, impl)),
              ("block_diagonal", lambda impl: __import__("planner_check").block_diagonal(90_000, 256, 0.12, 42, impl)),
              ("hub_rows", lambda impl: __import__("planner_check").hubs(150_000, 10, 20, 70_000, 43, impl)),
              ("few_long_rows_over_4m_columns", lambda impl: __import__("planner_check").uniform(2_048, 4_000_000, 600, 44, impl)),
              ("float_one_slice", lambda impl: __import__("planner_check").block_diagonal(360_000, 64, 0.5, 45, impl)),
              ("stencil", lambda impl: __import__("planner_check").diagonals(200_000, (-40_000, 0, 40_000), 30, 12, 46, impl))]


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("kind", [k for k, _ in STRUCTURED])
def test_structured_matrices_match_the_oracle_under_the_planners_own_plan(kind, impl):
    """The plans round 6 added (one-slice plans from the census, per-lane sums for hub-row blocks, SWEEP for tiny units, OWNER24 for one-slice float plans)
    against oracle/cpu_ref.c: whole run, a burst (the carried combine) and the reference's partition-by-partition loop -- bit-exact in fixed point,
    1e-4 in the float modes."""
    import numpy as np
    from hisparse_amd import device, host
    from oracle import oracle as orc
    import cases
    m = dict(STRUCTURED)[kind](impl)
    if impl != 0:
        m.data *= np.float32(0.25)
    cp = host.format_matrix(host.CSRMatrix.from_scipy(m), impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 7, impl))
    want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    with device.SpmvEngine(impl) as eng:
        eng.load_matrix(cp)
        st = eng.stats()
        eng.load_vector(xw)
        eng.run()
        one = eng.read_result()
        eng.run_batch(3)
        burst = eng.read_result()
        for j in range(cp.num_row_partitions):
            eng.run_partition(j, cp.part_len(j))
        parts = eng.read_result()
    for y in (one, burst, parts):
        assert np.array_equal(y, want) if impl == 0 else cases.float_close(y, want), (kind, device.STREAM_FORMATS[st["stream_format"]], st["col_slices"])
    fmt = device.STREAM_FORMATS[st["stream_format"]]
    if kind in ("banded", "block_diagonal"):
        assert st["col_slices"] == 1 or fmt in ("owner24", "sweep"), (fmt, st["col_slices"])
    if kind == "few_long_rows_over_4m_columns":
        assert fmt == "sweep"
    if kind == "float_one_slice" and impl != 0:
        assert fmt == "owner24"
