"""PageRank-style iterative caller on one GPU: y = M x on the device, x = d*y + (1-d)/n fed back
in HBM (hs_feedback); `iters` iterations through one hs_iterate call vs run + feedback issued one by one from Python
(HISPARSE_ITERATE_GRAPH=1 makes hs_iterate replay captured hipGraphs).

  python tools/pagerank.py [config] [iters]        config from hisparse_amd/datasets.py (default ppa_small)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1] if len(sys.argv) > 1 else "ppa_small"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
n = csr.num_rows
if csr.num_cols != n:
    sys.exit(f"{name} is {n} x {csr.num_cols}: PageRank needs a square matrix")
csr.normalize_by_outdegree()
t0 = time.perf_counter()
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
t_format = time.perf_counter() - t0
eng = device.SpmvEngine(impl)
t0 = time.perf_counter()
eng.load_matrix(cp)
t_load = time.perf_counter() - t0
d = 0.85
scale = int(host.pack_vector(impl, np.array([d], dtype=np.float32))[0])
shift = int(host.pack_vector(impl, np.array([(1 - d) / n], dtype=np.float32))[0])
x0 = host.pack_vector(impl, np.full(cp.num_cols, 1.0 / n, dtype=np.float32))

def timed(fn):
    eng.load_vector(x0); eng.sync()
    t0 = time.perf_counter(); fn(); eng.sync()
    return (time.perf_counter() - t0) / iters * 1e6

def plain():
    for _ in range(iters):
        eng.run(); eng.feedback(scale, shift)

for _ in range(2):
    us_graph = timed(lambda: eng.iterate(iters, scale, shift))
    us_plain = timed(plain)
ranks = host.unpack_result(impl, eng.read_result())[:n]
print(f"{name}: n {n}, nnz {cp.nnz}, format {t_format:.2f} s + load {t_load:.2f} s once; per iteration {us_graph:.1f} us (hs_iterate) "
      f"vs {us_plain:.1f} us (run + feedback from Python); rank mass {ranks.sum():.4f}, top {np.sort(ranks)[-3:][::-1]}")
