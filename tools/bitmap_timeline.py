"""Per-wavefront timeline of spmv_bitmap_kernel (profiling build HISPARSE_ABLATE=64): python tools/bitmap_timeline.py [config]
Prints, in microseconds from the first wavefront's entry, the median / min / max over all wavefronts of every timestamp."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "bitmap_timeline.bin")
os.environ["HISPARSE_ABLATE"] = "64"
os.environ["HISPARSE_TIMELINE_OUT"] = path
os.environ.setdefault("HISPARSE_STREAM_FORMAT", "bitmap")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _prof
_prof.use_profiling_library()      # the timeline / ablation instantiations are not in the product library
from hisparse_amd import host, device, datasets

name = sys.argv[1] if len(sys.argv) > 1 else "transformer_50"
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
eng.load_vector(host.pack_vector(impl, np.random.default_rng(0).normal(size=cp.num_cols).astype(np.float32)))
for _ in range(20):
    eng.run()
eng.sync()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 16, 8).astype(np.int64)
t0 = t[:, :, 0].min()
names = ["entry", "descriptors read", "first masks in", "first batch consumed", "run finished", "row sums in LDS", "after barrier", "stored"]
print(f"{name}: {t.shape[0]} workgroups x 16 wavefronts; microseconds after the first wavefront's entry (100 MHz clock)")
for i, n in enumerate(names):
    v = (t[:, :, i] - t0) / 100.0
    print(f"  {n:22s} median {np.median(v):6.2f}   min {v.min():6.2f}   max {v.max():6.2f}")
d = (t[:, :, 1:] - t[:, :, :-1]) / 100.0
print("  per-wavefront phase durations (median): " + ", ".join(f"{names[i]}->{names[i+1]} {np.median(d[:, :, i]):.2f}" for i in range(7)))
# systematic order inside a workgroup?  median over the workgroups, per wavefront index
for i in (0, 2, 3, 4):
    v = np.median((t[:, :, i] - t0) / 100.0, axis=0)
    print(f"  {names[i]:22s} by wavefront index: " + " ".join(f"{x:5.2f}" for x in v))
# and per XCD (workgroup id % 8)
fin = (t[:, :, 4] - t0) / 100.0
print("  run finished, median per XCD (blockIdx % 8): " + " ".join(f"{np.median(fin[k::8]):5.2f}" for k in range(8)))
print("  run finished, max per workgroup: median %.2f, min %.2f, max %.2f" % (np.median(fin.max(axis=1)), fin.max(axis=1).min(), fin.max(axis=1).max()))
