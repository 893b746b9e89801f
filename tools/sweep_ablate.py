"""Where the SWEEP kernel's time goes: the kernel ALONE (hs_time_kernel) under the profiling build's ablations (libhisparse_hip_prof.so; wrong
results by design).   python tools/sweep_ablate.py <config> [impl] [ablations ...]
bits: 1 no LDS accumulation, 2 the gather reads a line near the chunk base, 4 no zeroing / no result store, 8 no gather at all."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _prof
_prof.use_profiling_library()
from hisparse_amd import host, device, datasets

name = sys.argv[1]
cfg, csr = datasets.load(name)
impl = host.impl_id(sys.argv[2] if len(sys.argv) > 2 else cfg.impl)
ablations = [int(a) for a in sys.argv[3:]] or [0, 1, 2, 3, 4, 7, 9, 13]
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
st = eng.stats()
rng = np.random.default_rng(2024)
x = rng.uniform(0, 2, cp.num_cols).astype(np.float32) if impl == 0 else rng.normal(size=cp.num_cols).astype(np.float32)
eng.load_vector(host.pack_vector(impl, x))
print(f"{name}: {device.STREAM_FORMATS[st['stream_format']]} {st['col_slices']} slices {st['num_blocks']} blocks, image {st['stream_bytes']/1e6:.1f} MB, nnz {cp.nnz}")
for a in ablations + ablations[:1]:
    os.environ["HISPARSE_ABLATE"] = str(a)
    eng.time_kernel(300, 10)
    best = min(eng.time_kernel(0, 300) / 300 for _ in range(3))
    print(f"{name} ablate={a:2d} kernel alone {best*1e3:7.2f} us   image at {st['stream_bytes']/best/1e9:6.0f} GB/s")
