"""SpMSpV throughput (extension, spmspv.hip): python tools/spmspv_probe.py [config]
For x with 0.05 % ... 20 % of the columns set: time per call of (a) hs_spmspv_device -- the entries resident on the device: the two launches
and nothing else --, (b) hs_spmspv with host entries on the sparse path (pinned staging + async H2D in front), (c) the dense dispatch (x
scattered into a zero vector + the dense SpMV of the same matrix), and hs_run itself; the crossover is where (a)/(b) pass (c).  Products per
second and the rate over the bytes the selected columns hold (8 B per product: row index + value word).  Checked against the dense SpMV of
the same matrix with x densified (bit-exact in fixed point)."""
import ctypes as C
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
eng.load_matrix_csr(csr)                      # the dense matrix on the same context: the dispatch target
eng.load_matrix_csc(indptr, ridx, words, rows)
rng = np.random.default_rng(1)
rt = C.CDLL("libamdhip64.so")
rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
lib = device.lib()


def timed(fn, reps=100):
    for _ in range(10):
        fn()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.sync()
    return (time.perf_counter() - t0) / reps * 1e6


x0 = rng.uniform(0.0, 2.0, eng.num_cols).astype(np.float32) if impl == 0 else rng.normal(size=eng.num_cols).astype(np.float32)
eng.load_vector(host.pack_vector(impl, x0))
t_dense = timed(eng.run)
print(f"{name}: {rows} x {cols}, nnz {csr.nnz}; hs_run (dense SpMV) {t_dense:.1f} us")
print(f"{'columns':>9s} {'entries':>9s} {'products':>10s} | {'device entries':>14s} {'host entries':>13s} {'dense dispatch':>14s} | G products/s   GB/s over the selected columns (device entries) | parity")
FRACS = [float(f) for f in os.environ["FRACS"].split(",")] if os.environ.get("FRACS") else (0.0005, 0.001, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2)
for frac in FRACS:
    n = max(1, int(cols * frac))
    xi = np.sort(rng.choice(cols, size=n, replace=False)).astype(np.uint32)
    xv = rng.uniform(0.0, 2.0, n).astype(np.float32) if impl == 0 else rng.normal(size=n).astype(np.float32)
    xw = host.pack_vector(impl, xv)
    products = int((indptr[xi + 1].astype(np.int64) - indptr[xi].astype(np.int64)).sum())
    pairs = np.empty((n, 2), dtype=np.uint32); pairs[:, 0] = xi; pairs[:, 1] = xw
    d = C.c_void_p()
    assert rt.hipMalloc(C.byref(d), max(pairs.nbytes, 8)) == 0 and rt.hipMemcpy(d, pairs.ctypes.data, pairs.nbytes, 1) == 0
    os.environ["HISPARSE_SPMSPV"] = "sparse"
    t_dev = timed(lambda: lib.hs_spmspv_device(eng._h, d, n))
    y_dev = eng.read_spmspv_result()
    t_host = timed(lambda: lib.hs_spmspv(eng._h, pairs.ctypes.data, n))
    y_host = eng.read_spmspv_result()
    os.environ["HISPARSE_SPMSPV"] = "dense"
    t_disp = timed(lambda: lib.hs_spmspv(eng._h, pairs.ctypes.data, n))
    y_disp = eng.read_spmspv_result()
    os.environ["HISPARSE_SPMSPV"] = "auto"      # (opt-in since round 5: both loads hold the same matrix)
    before = timed(lambda: lib.hs_spmspv(eng._h, pairs.ctypes.data, n), reps=3)      # (the first call times the dense SpMV once)
    t_auto = timed(lambda: lib.hs_spmspv(eng._h, pairs.ctypes.data, n))
    same = (lambda a, b: np.array_equal(a, b)) if impl == 0 else (lambda a, b: bool(np.allclose(a.view(np.float32), b.view(np.float32), rtol=1e-4, atol=1e-4)))
    ok = same(y_dev, y_disp) and same(y_host, y_disp)
    print(f"{frac*100:8.2f}% {n:9d} {products:10d} | {t_dev:11.1f} us {t_host:10.1f} us {t_disp:11.1f} us  auto {t_auto:7.1f} us | {products/t_dev/1e3:10.2f} {products*8/t_dev/1e3:10.1f} | {'all three agree' if ok else 'DIFFER'}", flush=True)
