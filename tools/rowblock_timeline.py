"""Phase timeline of spmv_rowblock_kernel (profiling build HISPARSE_ABLATE=512: full work, correct results):
    python tools/rowblock_timeline.py <config>          (HISPARSE_* switches select format / plan as usual)
Per workgroup and block: entered -> prologue done -> main loop finished (consumer wavefront 0 / loader wavefront 14) -> all wavefronts
finished -> result stores issued.  100 MHz clock; microseconds after the first workgroup's entry."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "rowblock_timeline.bin")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _prof
_prof.use_profiling_library()      # the timeline / ablation instantiations are not in the product library
from hisparse_amd import host, device, datasets

name = sys.argv[1]
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
x = np.random.default_rng(0).normal(size=cp.num_cols).astype(np.float32) if impl else np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32)
eng.load_vector(host.pack_vector(impl, x))
for _ in range(30):
    eng.run()
eng.sync()
_, kern = eng.time_runs(5, 50)
os.environ["HISPARSE_ABLATE"] = "512"
os.environ["HISPARSE_TIMELINE_OUT"] = path
for _ in range(3):
    eng.run()
eng.sync()
del os.environ["HISPARSE_ABLATE"]
st = eng.stats()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4, 2, 8).astype(np.int64)        # [workgroup][block][consumer 0 / loader 14][stamp]
live = t[:, :, 0, 0] > 0
t0 = t[:, 0, 0, 0][live[:, 0]].min()
us = lambda v: (v - t0) / 100.0
print(f"{name}: {device.STREAM_FORMATS[st['stream_format']]}, {st['col_slices']} slices, {st['num_blocks']} blocks on {t.shape[0]} workgroups; kernel {kern / 50 * 1e3:.1f} us (HIP events, product build)")
names = ["entered", "prologue done", "own loop done", "all loops done", "stores issued"]
for k in range(4):
    m = live[:, k]
    if not m.any():
        break
    print(f" block {k} of its workgroup ({int(m.sum())} workgroups):")
    for who, w in (("consumer 0", 0), ("loader 14", 1)):
        for i, n in enumerate(names):
            v = us(t[m, k, w, i])
            print(f"   {who:10s} {n:15s} median {np.median(v):7.2f}   min {v.min():7.2f}   max {v.max():7.2f}")
    d = t[m, k, 0, :5]
    print("   durations, consumer 0, median us: prologue %.2f  main loop %.2f  wait for the others %.2f  stores %.2f" % tuple(np.median((d[:, i + 1] - d[:, i]) / 100.0) for i in range(4)))
    dl = t[m, k, 1, :5]
    print("   loader 14: its loop ends %.2f us (median) before the slowest consumer's" % np.median((dl[:, 3] - dl[:, 2]) / 100.0))
last = np.array([t[g, live[g].sum() - 1, 0, 4] for g in range(t.shape[0]) if live[g].any()])
print(" last stamp of a workgroup: median %.2f  min %.2f  max %.2f us after the first entry" % (np.median(us(last)), us(last).min(), us(last).max()))
