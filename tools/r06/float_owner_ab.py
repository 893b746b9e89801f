"""round 6: float modes -- where the row-block planner ends up with a PAIRS image (ds_add_f64 row sums), is OWNER24 (owned rows, plain read-modify-write) faster?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import datasets, host
import planner_check as pc

def ref(name):
    return lambda: pc.reference(name)
CASES = [("mouse_gene_slab2", ref("mouse_gene_slab2")), ("mouse_gene_slab4", ref("mouse_gene_slab4")), ("mouse_gene_slab8", ref("mouse_gene_slab8")), ("gplus", ref("gplus")),
         ("ogbl_ppa", ref("ogbl_ppa")), ("er_300k_30", lambda: pc.uniform(300_000, 300_000, 30, 12, 1)), ("rmat19", lambda: pc.rmat(19, 20_000_000, 0.45, 0.15, 0.15, 5, 1)),
         ("banded_400k", lambda: pc.banded(400_000, 40, 2_000, 1, 1)), ("blockdiag_200k", lambda: pc.block_diagonal(200_000, 512, 0.10, 3, 1)),
         ("tall_2m_x_50k", lambda: pc.uniform(2_000_000, 50_000, 10, 10, 1)), ("slab8_of_er_300k", lambda: pc.slab(pc.uniform(300_000, 300_000, 30, 12, 1), 8, 5)),
         ("slab4_of_banded", lambda: pc.slab(pc.banded(400_000, 40, 2_000, 1, 1), 4, 1))]
for impl in (1, 2):
    for name, build in CASES:
        m = build()
        csr = host.CSRMatrix.from_scipy(m)
        rng = np.random.default_rng(99)
        cols8 = (m.shape[1] + 7) // 8 * 8
        xw = host.pack_vector(impl, rng.normal(size=cols8).astype(np.float32))
        own_us, own_plan, y = pc.time_plan(impl, csr, xw, {}, 200)
        line = f"{name:20s} {['fixed','float_pob','float_stall'][impl]:11s} planner {own_plan:12s} {own_us:8.2f} |"
        for f in ("pairs", "delta", "owner24", "sweep"):
            us, plan, _ = pc.time_plan(impl, csr, xw, {"stream_format": f, "light": "0"}, 200, want=y)
            line += f" {plan:11s} {('%8.2f' % us) if us else '   -    '} |"
        print(line, flush=True)
