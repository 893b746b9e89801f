"""round 6: fixed point -- one-slice row-block plans against OWNER24 forced to one slice"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host
import planner_check as pc
names = ["banded_400k_d40_w2k", "blockdiag_200k_b512_p10", "tall_2m_x_50k_10", "slab4_of_banded_400k", "dense_2048_x_8k_15"]
second = ["road_2m_d3_w1k", "stencil_300k_3x20", "dense5_20k", "wide_2k_x_8m_2000", "tiny_20k_x_200k_5"]
cases = [c for c in pc.CASES if c[0] in names] + [c for c in pc.SECOND if c[0] in second] + [("ref_mouse_gene", 0, lambda: pc.reference("mouse_gene")), ("ref_mouse_gene_slab2", 0, lambda: pc.reference("mouse_gene_slab2"))]
for name, impl, build in cases:
    m = build()
    csr = host.CSRMatrix.from_scipy(m)
    rng = np.random.default_rng(99)
    cols8 = (m.shape[1] + 7) // 8 * 8
    x = rng.uniform(0.0, 2.0, cols8).astype(np.float32) if impl == 0 else rng.normal(size=cols8).astype(np.float32)
    xw = host.pack_vector(impl, x)
    own_us, own_plan, y = pc.time_plan(impl, csr, xw, {}, 200)
    line = f"{name:26s} {impl} planner {own_plan:12s} {own_us:8.2f} |"
    for opts in ({"stream_format": "owner24", "col_slices": "1"}, {"stream_format": "owner24"}, {"stream_format": "pairs", "col_slices": "1"}, {"stream_format": "delta", "col_slices": "1"}, {"stream_format": "sweep"}):
        us, plan, _ = pc.time_plan(impl, csr, xw, dict(opts, light="0"), 200, want=y)
        line += f" {plan:11s} {('%8.2f' % us) if us else '   -    '} |"
    print(line, flush=True)
