"""What one SpMV costs when x comes from and y goes back to HOST memory every time (the reference's migrate pattern,
sw/host.cpp:355-370): hs_load_vector + hs_run + hs_read_result per step.  Never the bench `value`; noted in DESIGN.md."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1] if len(sys.argv) > 1 else "ogbl_ppa"
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
t0 = time.perf_counter(); eng.load_matrix(cp); t_load = time.perf_counter() - t0
xw = host.pack_vector(impl, np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32))
for _ in range(5):
    eng.load_vector(xw); eng.run(); eng.read_result()
steps = 50
t0 = time.perf_counter()
for _ in range(steps):
    eng.load_vector(xw); eng.run(); y = eng.read_result()
t = (time.perf_counter() - t0) / steps
print(f"{name}: host x -> SpMV -> host y: {t*1e6:.1f} us per step = {8*cp.nnz/t/1e9:.0f} GB/s (8 B per non-zero); "
      f"x {xw.nbytes/1e6:.1f} MB up, y {y.nbytes/1e6:.1f} MB down per step; matrix load once: {t_load:.2f} s for {eng.stats()['stream_bytes']/1e6:.0f} MB")
