"""round 6: the hub-row miss of tools/planner_check.py under DELTA with per-lane register sums forced on (row_runs = 1)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host
import planner_check as pc
for name in ("hubs_500k_15_plus_50x200k", "rmat19_45_15_15"):
    case = next(c for c in pc.CASES if c[0] == name)
    m = case[2](); impl = case[1]
    csr = host.CSRMatrix.from_scipy(m)
    rng = np.random.default_rng(99)
    cols8 = (m.shape[1] + 7) // 8 * 8
    xw = host.pack_vector(impl, rng.uniform(0.0, 2.0, cols8).astype(np.float32))
    own_us, own_plan, y = pc.time_plan(impl, csr, xw, {}, 200)
    print(f"{name}: planner {own_plan} {own_us:.2f} us", flush=True)
    for opts in ({"stream_format": "delta"}, {"stream_format": "delta", "row_runs": "1"}, {"stream_format": "delta", "row_runs": "1", "col_slices": "1"},
                 {"stream_format": "delta", "row_runs": "1", "col_slices": "3"}, {"stream_format": "delta", "row_runs": "1", "col_slices": "12"},
                 {"stream_format": "pairs", "col_slices": "12"}, {"stream_format": "sweep"}):
        us, plan, _ = pc.time_plan(impl, csr, xw, dict(opts, light="0"), 200, want=y)
        print(f"    {str(opts):80s} {plan:12s} {us if us is None else round(us, 2)}", flush=True)
