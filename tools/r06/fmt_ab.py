"""round 6: forced-format A/B on a few matrices (whole step, tools/planner_check.time_plan):  python tools/r06/fmt_ab.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import datasets, host
import planner_check as pc

CASES = [("powerlaw 300K^2 3M (gap 30K)", 0, lambda: host.CSRMatrix.generate("powerlaw", 300000, 300000, a=3.0e6, b=0.4, c=1.0, seed=5)),
         ("powerlaw 800K^2 20M (gap 32K)", 0, lambda: host.CSRMatrix.generate("powerlaw", 800000, 800000, a=2.0e7, b=0.4, c=1.0, seed=4)),
         ("powerlaw 500K^2 10M (gap 25K)", 0, lambda: host.CSRMatrix.generate("powerlaw", 500000, 500000, a=1.0e7, b=0.35, c=1.0, seed=8)),
         ("powerlaw 1M^2 40M (gap 25K, 320 MB)", 0, lambda: host.CSRMatrix.generate("powerlaw", 1000000, 1000000, a=4.0e7, b=0.35, c=1.0, seed=9)),
         ("powerlaw 800K^2 20M float_pob", 1, lambda: host.CSRMatrix.generate("powerlaw", 800000, 800000, a=2.0e7, b=0.4, c=2.0, seed=4)),
         ("hollywood", 0, lambda: datasets.load("hollywood")[1]), ("hollywood again", 0, lambda: datasets.load("hollywood")[1]),
         ("ogbn_products fixed", 0, lambda: datasets.load("ogbn_products")[1])]
for name, impl, build in CASES:
    csr = build()
    rng = np.random.default_rng(99)
    cols8 = (csr.num_cols + 7) // 8 * 8
    x = rng.uniform(0.0, 2.0, cols8).astype(np.float32) if impl == 0 else rng.normal(size=cols8).astype(np.float32)
    xw = host.pack_vector(impl, x)
    own_us, own_plan, y = pc.time_plan(impl, csr, xw, {}, 200)
    line = f"{name:38s} planner {own_plan:12s} {own_us:8.2f} us |"
    for f in ("delta", "owner24", "sweep", "pairs"):
        us, plan, _ = pc.time_plan(impl, csr, xw, {"stream_format": f, "light": "0"}, 200, want=y)
        line += f" {plan:11s} {('%8.2f' % us) if us else '   -    '} |"
    print(line, flush=True)
