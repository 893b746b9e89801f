"""round 6: SWEEP ring depth on images that are STREAMED from HBM (beyond the Infinity Cache, `nt` loads): ogbn-products as a forced SWEEP image, whole step,
under variant builds (HISPARSE_HIP_LIB); python tools/r06/sweep_depth_streamed.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import datasets, host
import planner_check as pc
for name, impl in (("ogbn_products", 0), ("ogbn_products", 2), ("pokec", 0)):
    cfg, csr = datasets.load(name)
    rng = np.random.default_rng(99)
    cols8 = (csr.num_cols + 7) // 8 * 8
    x = rng.uniform(0.0, 2.0, cols8).astype(np.float32) if impl == 0 else rng.normal(size=cols8).astype(np.float32)
    xw = host.pack_vector(impl, x)
    line = f"{os.path.basename(os.environ.get('HISPARSE_HIP_LIB', 'product')):28s} {name}/{impl}:"
    for opts in ({"stream_format": "sweep"}, {"stream_format": "sweep", "col_slices": "8"}, {"stream_format": "sweep", "stream_resident": "1"}, {"stream_format": "owner24"}):
        us, plan, _ = pc.time_plan(impl, csr, xw, dict(opts, light="0"), 100)
        line += f" {str(sorted(opts.items())[-1][1]) if len(opts) > 1 else ''}{plan:11s} {('%7.2f' % us) if us else '  -  '} |"
    print(line, flush=True)
