"""Perf probe on the ogbl-ppa stand-in (fixed point).  HISPARSE_ABLATE=<bits> selects a profiling variant."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device

n = 576289
csr = host.CSRMatrix.generate("powerlaw", n, n, a=42463862, b=0.35, c=1.0, seed=42)
cp = host.format_matrix(csr, 0, skip_empty_rows=True)
eng = device.SpmvEngine(0)
eng.load_matrix(cp)
st = eng.stats()
x = np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32)
eng.load_vector(host.pack_vector(0, x))
runs = int(os.environ.get("RUNS", "50"))
for k in range(2):
    tot, kern = eng.time_runs(5, runs)
print("ablate", os.environ.get("HISPARSE_ABLATE", "0"), "depth", os.environ.get("HISPARSE_DEPTH", "8"), "ms/run %.4f kernel ms %.4f | algorithmic %.0f GB/s | stream %.0f GB/s" % (
    tot / runs, kern / runs, 8 * cp.nnz / (kern / runs * 1e-3) / 1e9, st["stream_bytes"] / (kern / runs * 1e-3) / 1e9))
