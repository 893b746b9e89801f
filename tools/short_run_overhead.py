"""What a short timed region costs beyond its steps: wall time of K back-to-back SpMVs bracketed by a sync on both sides, K = 1 ... 200,
fitted as a + b K (a = launch latency of the first kernel + the wake-up after the last; the driver's bench run uses K = 20).
python tools/short_run_overhead.py [config]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1] if len(sys.argv) > 1 else "ogbl_ppa"
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
eng.load_vector(host.pack_vector(impl, np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32)))
for _ in range(400):
    eng.run()
eng.sync()
ks = [1, 2, 5, 10, 20, 50, 100, 200]
best = {}
for rep in range(7):
    for k in ks:
        for _ in range(5):
            eng.run()
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(k):
            eng.run()
        eng.sync()
        us = (time.perf_counter() - t0) * 1e6
        best[k] = min(best.get(k, 1e18), us)
b, a = np.polyfit(ks, [best[k] for k in ks], 1)
for k in ks:
    print(f"{name}: K = {k:3d}: {best[k]:9.1f} us = {best[k] / k:7.2f} us per step")
print(f"{name}: fit {a:.1f} us + {b:.2f} us x K")
# the same K steps between two events recorded inside the bracket (device time only)
for k in (20, 200):
    region_ms, _ = eng.time_runs(5, k, kernel=False)
    print(f"{name}: K = {k}: two events around the K launches: {region_ms / k * 1e3:.2f} us per step")
