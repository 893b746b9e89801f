"""CPU only (the host builder, hs_tiles_build): the plan the library takes for the reference's matrices and for the out-of-sample cases of
tools/planner_check.py, WITH the tile census (round 6) and with HISPARSE_PLAN_CENSUS=0 (the planner of rounds 1-5) -- which plans changed?
    python tools/plan_diff.py [dataset names / planner_check case names ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hisparse_amd import datasets, device, host
import planner_check as pc

REFERENCE = [(n, None) for n, _ in datasets.BM_LIST] + [("ogbl_ppa_rmat", None), ("mouse_gene_slab8", None), ("mouse_gene_slab4", None), ("mouse_gene_slab2", None),
                                                        ("pokec", "float_pob"), ("ogbn_products", "float_stall"), ("mouse_gene", "float_pob"), ("transformer_50", "float_pob"), ("transformer_80", "float_pob")]
names = sys.argv[1:]


def plan(csr, impl):
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    out = {}
    for census in ("1", "0"):
        os.environ["HISPARSE_PLAN_CENSUS"] = census
        t0 = time.perf_counter()
        t = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, 256)
        out[census] = (f"{t['format']} x{t['col_slices']}, {len(t['blocks'])} blocks, {len(t['units'])} units, {len(t['image'])/1e6:.0f} MB", time.perf_counter() - t0)
    return out


for name, impl_name in REFERENCE:
    if names and name not in names:
        continue
    cfg, csr = datasets.load(name)
    impl = host.impl_id(impl_name or "fixed")
    p = plan(csr, impl)
    print(f"{name + '/' + (impl_name or 'fixed'):34s} census: {p['1'][0]:52s} ({p['1'][1]:.2f} s) | without: {p['0'][0]:52s} ({p['0'][1]:.2f} s) {'' if p['1'][0] == p['0'][0] else '  <-- CHANGED'}", flush=True)
for name, impl, build in pc.CASES:
    if names and name not in names:
        continue
    m = build()
    p = plan(host.CSRMatrix.from_scipy(m), impl)
    print(f"{name + '/' + ['fixed', 'float_pob', 'float_stall'][impl]:34s} census: {p['1'][0]:52s} ({p['1'][1]:.2f} s) | without: {p['0'][0]:52s} ({p['0'][1]:.2f} s) {'' if p['1'][0] == p['0'][0] else '  <-- CHANGED'}", flush=True)
