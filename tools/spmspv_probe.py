"""SpMSpV throughput (extension, spmspv.hip): python tools/spmspv_probe.py [config]     (HISPARSE_SPMSPV=atomic: the direct scatter path)
For x with 0.1 % / 1 % / 10 % / 50 % of the columns set: time per hs_spmspv call (host-synchronous: upload of x, one stream sync for the
product count, the passes), products per second, and the rate over the bytes the selected columns hold (8 B per product: row index + value
word) -- the operator's own "touched bytes".  Checked against the SpMV of the same matrix with x densified (bit-exact, fixed point)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1] if len(sys.argv) > 1 else "ogbl_ppa"
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
indptr, ridx, words = host.csr_to_csc(csr, impl)
rows, cols = csr.num_rows, csr.num_cols
eng = device.SpmvEngine(impl)
eng.load_matrix_csc(indptr, ridx, words, rows)
spmv = device.SpmvEngine(impl)
spmv.load_matrix_csr(csr)
rng = np.random.default_rng(1)
print(f"{name}: {rows} x {cols}, nnz {csr.nnz}, {'atomic scatter' if os.environ.get('HISPARSE_SPMSPV') == 'atomic' else 'binned (no global atomics)'} path")
for frac in (0.001, 0.01, 0.1, 0.5):
    n = max(1, int(cols * frac))
    xi = np.sort(rng.choice(cols, size=n, replace=False)).astype(np.uint32)
    xv = rng.uniform(0.0, 2.0, n).astype(np.float32) if impl == 0 else rng.normal(size=n).astype(np.float32)
    xw = host.pack_vector(impl, xv)
    products = int((indptr[xi + 1].astype(np.int64) - indptr[xi].astype(np.int64)).sum())
    y = eng.spmspv(xi, xw)
    x = np.zeros(spmv.num_cols, dtype=np.float32)
    x[xi] = xv
    spmv.load_vector(host.pack_vector(impl, x))
    spmv.run()
    want = spmv.read_result()[:rows]
    ok = np.array_equal(y, want) if impl == 0 else bool(np.allclose(y.view(np.float32), want.view(np.float32), rtol=1e-4, atol=1e-4))
    pairs = np.empty((n, 2), dtype=np.uint32); pairs[:, 0] = xi; pairs[:, 1] = xw
    import ctypes as C
    reps = 20
    for _ in range(3):
        device.lib().hs_spmspv(eng._h, pairs.ctypes.data, n)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        device.lib().hs_spmspv(eng._h, pairs.ctypes.data, n)
    eng.sync()
    t = (time.perf_counter() - t0) / reps
    print(f"  {frac*100:5.1f} % of the columns ({n:8d} entries, {products:10d} products): {t*1e6:9.1f} us per call, {products/t/1e9:7.2f} G products/s, "
          f"{products*8/t/1e9:8.1f} GB/s over the selected columns' bytes; result {'matches the SpMV' if ok else 'DIFFERS'}")
