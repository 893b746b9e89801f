"""hs_spmm_device on a named config: k columns through the fused BITMAP kernel against k SpMVs (HISPARSE_SPMM_FUSED=0 in a second process):
python tools/spmm_probe.py <config> [k]"""
import ctypes as C
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1] if len(sys.argv) > 1 else "transformer_50"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
rng = np.random.default_rng(0)
X = np.stack([host.pack_vector(impl, rng.normal(size=cp.num_cols).astype(np.float32) * 0.1 if impl else rng.uniform(0, 1, cp.num_cols).astype(np.float32))
              for _ in range(k)])
device.lib()
rt = C.CDLL("libamdhip64.so")
rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
xd, yd = C.c_void_p(), C.c_void_p()
assert rt.hipMalloc(C.byref(xd), X.nbytes) == 0 and rt.hipMalloc(C.byref(yd), k * cp.num_rows * 4) == 0
assert rt.hipMemcpy(xd, X.ctypes.data, X.nbytes, 1) == 0
with device.SpmvEngine(impl) as eng:
    eng.load_matrix(cp)
    st = eng.stats()
    for _ in range(50):
        eng.spmm_device(xd.value, cp.num_cols, yd.value, cp.num_rows, k)
    eng.sync()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.spmm_device(xd.value, cp.num_cols, yd.value, cp.num_rows, k)
    eng.sync()
    us = (time.perf_counter() - t0) / reps * 1e6
print("%-16s %s k=%d fused=%s: %.1f us per SpMM = %.1f us per column = %.0f GB/s on the reference's 8 B per non-zero and column" % (
    name, device.STREAM_FORMATS[st["stream_format"]], k, os.environ.get("HISPARSE_SPMM_FUSED", "1"), us, us / k, 8.0 * cp.nnz * k / us / 1e3))
