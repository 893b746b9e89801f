"""Out-of-sample check of the load-time planner (tools/planner_check.py holds the generators and the measurement):
over matrices from families none of the planner's constants was measured on -- banded, block-diagonal, R-MAT with other quadrant
probabilities, wide bipartite, tall-narrow, Erdos-Renyi, dense-row layers of other shapes, 2- / 4- / 8-way row slabs -- the plan the library
takes by itself must run within 10 % of the best plan that can be FORCED (every format, the planner's format at half / twice its slices),
and every forced plan must give the planner's y.  The reference's analogue is its design-space sweep
(performance_model/design_space_exp.cpp:496-547).

The full list (24 matrices; `python tools/planner_check.py`, profiles/r06_planner_check_after.txt) has four known misses, asserted here at
their measured ratio + a margin so that they are SEEN, not hidden: a matrix whose 50 hub rows hold 64 % of the non-zeros (PAIRS: all lanes
of a step on one LDS accumulator; SWEEP would be 1.9 x faster), a float block-diagonal matrix of 64 x 64 blocks (OWNER24 1.44 x), a 3 M x 8 K
tall matrix of 6 non-zeros per row (OWNER24 1.17 x) and a 60 %-dense float layer (DELTA 1.13 x: the sliced-DELTA rule is fixed-point only).
Before the tile census of round 6 the same list read: 14 of 24 within 10 %, six between 2.1 x and 5.9 x (profiles/r06_planner_check_before.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu

# (name, bound on planner time / best forced time).  A subset that builds in a few seconds each; the whole list: tools/planner_check.py.
WITHIN_10_PERCENT = ["banded_400k_d40_w2k", "blockdiag_200k_b512_p10", "rmat19_45_15_15", "bipartite_20k_x_2m_200", "bipartite_100k_x_4m_60", "tall_2m_x_50k_10",
                     "er_300k_30", "er_1500k_8", "dense_2048_x_8k_15", "dense_256_x_64k_8", "slab8_of_rmat19", "slab4_of_banded_400k", "slab8_of_er_300k",
                     "slab4_of_bipartite_100k", "slab8_of_tall_2m", "banded_1m_d12_w50k"]
KNOWN_MISSES = {"tall_3m_x_8k_6": 1.30, "dense_4096_x_4k_60": 1.25}      # (+ hubs 1.93 and blockdiag_600k 1.44 in the full list: too slow to build here)


@pytest.mark.parametrize("name", WITHIN_10_PERCENT + sorted(KNOWN_MISSES))
def test_planner_choice_against_the_best_forced_plan(name, record_property):
    import planner_check as pc
    case = next(c for c in pc.CASES if c[0] == name)
    res = pc.check(name, case[1], case[2](), steps=200, log=lambda s: None)
    record_property("planner", res["planner"])
    record_property("planner_us", res["planner_us"])
    record_property("best_forced", f"{res['best_forced']} {res['best_forced_us']} us")
    assert not res["wrong_results"], f"{name}: forced plans {res['wrong_results']} give another y than the planner's plan"
    timed = [v for v in res["variants"] if v["us"] is not None]
    assert len(timed) >= 3, f"{name}: only {len(timed)} forced plans loaded"
    bound = KNOWN_MISSES.get(name, 1.10)
    assert res["planner_over_best"] <= bound, (f"{name}: planner {res['planner']} {res['planner_us']} us, best forced {res['best_forced']} {res['best_forced_us']} us "
                                               f"= {res['planner_over_best']:.3f} x (bound {bound})")
