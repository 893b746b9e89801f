"""Where the wavefronts of the OWNER-format kernel spend their time (profiling build HISPARSE_ABLATE=256):
python tools/owner_profile.py [config] [launches]"""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.path.join(tempfile.gettempdir(), "owner_profile.bin")
os.environ["HISPARSE_ABLATE"] = "256"
os.environ["HISPARSE_TIMELINE_OUT"] = path
os.environ.setdefault("HISPARSE_STREAM_FORMAT", "owner")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _prof
_prof.use_profiling_library()      # the timeline / ablation instantiations are not in the product library
from hisparse_amd import host, device, datasets

name = sys.argv[1] if len(sys.argv) > 1 else "ogbn_products"
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
eng.load_vector(host.pack_vector(impl, np.random.default_rng(0).normal(size=cp.num_cols).astype(np.float32)))
for _ in range(launches):
    eng.run()
eng.sync()
st = eng.stats()
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 16, 8).astype(np.float64) / launches
clk = 100e6 if t[:, :14, 0].max() < 1e6 else 2.4e9      # s_memtime ticks: report both readings
print(f"{name}: {t.shape[0]} workgroups, {st['num_units']} units, ring {st['ring_buffers']}, slices {st['col_slices']}; per launch, in clock ticks")
cons, load = t[:, :14, :], t[:, 14:, :]
print(f"  consumers: phase {np.median(cons[:, :, 0]):.0f} (max {cons[:, :, 0].max():.0f})  flush {np.median(cons[:, :, 1]):.0f}  barrier wait {np.median(cons[:, :, 2]):.0f}"
      f"  units {np.median(cons[:, :, 3]):.0f}  steps {np.median(cons[:, :, 4]):.0f} (min {cons[:, :, 4].min():.0f} max {cons[:, :, 4].max():.0f})")
print(f"  loaders:   phase {np.median(load[:, :, 0]):.0f}  waiting for refills {np.median(load[:, :, 1]):.0f}  barrier wait {np.median(load[:, :, 2]):.0f}  issuing refills (+ descriptor waits) {np.median(load[:, :, 4]):.0f}")
print(f"  consumer shares: flush {np.median(cons[:, :, 1] / cons[:, :, 0]) * 100:.1f} %  barrier {np.median(cons[:, :, 2] / cons[:, :, 0]) * 100:.1f} %;"
      f"  loader: refill wait {np.median(load[:, :, 1] / load[:, :, 0]) * 100:.1f} %  barrier {np.median(load[:, :, 2] / load[:, :, 0]) * 100:.1f} %")
