"""Diagnostic: ogbl-ppa stand-in, x as in bench.py; repeated single runs compared against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
from hisparse_amd import host, device, datasets
from oracle import oracle as orc

cfg, csr = datasets.load(os.environ.get("CONFIG", "ogbl_ppa"))
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
x = np.random.default_rng(2024).uniform(0.0, 2.0, cp.num_cols).astype(np.float32)
xw = host.pack_vector(impl, x)
want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
eng.load_vector(xw)
st = eng.stats()
fails = []
for i in range(int(os.environ.get("RUNS", "30"))):
    eng.run(); eng.sync()
    got = eng.read_result()
    if impl == 0:
        bad = np.nonzero(got != want)[0]
    else:
        bad = np.nonzero(~np.isclose(got.view(np.float32), want.view(np.float32), rtol=1e-4, atol=1e-4))[0]
    if len(bad):
        fails.append((i, len(bad), int(bad[0]), int(bad[-1])))
        if len(fails) <= 8:
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez(f"gpurun_out/diag_bad_{len(fails)}.npz", rows=bad, got=got[bad], want=want[bad])
print(os.environ.get("TAG", ""), "format", st["stream_format"], "slices", st["col_slices"], "failing runs", len(fails), fails[:6])
