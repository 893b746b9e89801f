"""Every row slab of an N-way split of one matrix (sharding.split_rows_by_nnz: what rank r of `bench.py --gpus N` loads), timed on ONE GPU under
several planner overrides:   python tools/slab_probe.py <config> <N> "TAG:VAR=value,..." ...
Whole-step time (K back-to-back steps / K), best of 3."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets, sharding

name, n = sys.argv[1], int(sys.argv[2])
variants = []
for spec in sys.argv[3:] or ["default:"]:
    tag, _, envs = spec.partition(":")
    variants.append((tag, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
managed = sorted({k for _, env in variants for k in env})
cfg, full = datasets.load(name)
impl = host.impl_id(cfg.impl)
granule = 128 * (8 if impl == host.IMPL_FLOAT_STALL else 1)
indptr, indices, data = full.arrays()
cols8 = (full.num_cols + 7) // 8 * 8
rng = np.random.default_rng(2024)
x = rng.uniform(0.0, 2.0, cols8).astype(np.float32) if impl == host.IMPL_FIXED else rng.normal(size=cols8).astype(np.float32)
xw = host.pack_vector(impl, x)
bounds = sharding.split_rows_by_nnz(indptr, n, granule)
steps = int(os.environ.get("RUNS", "200"))
for r in range(n):
    lo, hi = bounds[r], bounds[r + 1]
    ip, ix, dv = sharding.slab_arrays(indptr, indices, data, lo, hi)
    csr = host.CSRMatrix.from_arrays(hi - lo, full.num_cols, ip, ix, dv)
    y0 = None
    for tag, env in variants:
        for k in managed:
            os.environ.pop(k, None)
        os.environ.update(env)
        with device.SpmvEngine(impl) as eng:
            eng.load_matrix_csr(csr)
            eng.load_vector(xw)
            st = eng.stats()
            for _ in range(150):
                eng.run()
            eng.sync()
            y = eng.read_result()
            if y0 is None:
                y0 = y
            best = min(eng.time_runs(5, steps, kernel=False)[0] / steps for _ in range(3)) * 1e3
        print(f"{name} {n}-way slab {r}: rows {hi - lo} nnz {int(ip[-1])} {tag:14s} {best:7.2f} us  {device.STREAM_FORMATS[st['stream_format']]:7s} "
              f"{st['col_slices']} slices {st['num_blocks']} blocks {st['num_units']} units ring {st['ring_buffers']}  y {'same' if np.array_equal(y, y0) else 'DIFFERS'}", flush=True)
