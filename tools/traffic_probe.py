"""The command bench.py puts under rocprofv3's counter passes (bench_extras.live_traffic): load one named configuration the way bench.py
does and launch a few SpMVs -- nothing else.   python tools/traffic_probe.py <config> [impl] [launches] [sync|batch]
sync (default; the counter passes): one step at a time.  batch (the --kernel-trace --stats pass, bench_extras.live_kernel_trace): the launches
back to back from hs_run_batch, the way bench.py's timed region enqueues them, in four batches with a synchronisation between them."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1]
cfg, csr = datasets.load(name)
impl = host.impl_id(sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] else cfg.impl)
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rng = np.random.default_rng(2024)
with device.SpmvEngine(impl) as eng:
    eng.load_matrix_csr(csr)                                    # (the image the CPSR path builds, byte for byte: tests/test_gpu_retile.py)
    padded_cols = eng.num_cols
    x = rng.uniform(0, 2, padded_cols).astype(np.float32) if impl == 0 else rng.normal(size=padded_cols).astype(np.float32)
    eng.load_vector(host.pack_vector(impl, x))
    if len(sys.argv) > 4 and sys.argv[4] == "batch":
        for _ in range(4):
            eng.run_batch(max(1, launches // 4))
            eng.sync()
    else:
        for _ in range(launches):
            eng.run()
            eng.sync()                                           # one step at a time: no carried combine, every launch a whole step
print("done", name, launches)
