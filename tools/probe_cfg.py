"""Kernel-time probe on any named config: python tools/probe_cfg.py <config> [impl]  (HISPARSE_* env selects variants;
an explicit impl (fixed|float_pob|float_stall) runs the config's matrix in another numeric mode, e.g. to use the
fixed-point-only HISPARSE_ABLATE builds on a float config)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1]
cfg, csr = datasets.load(name)
impl = host.impl_id(sys.argv[2] if len(sys.argv) > 2 else cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
st = eng.stats()
x = np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32)
eng.load_vector(host.pack_vector(impl, x))
runs = int(os.environ.get("RUNS", "50"))
best = whole = 1e9
for k in range(3):
    tot, kern = eng.time_runs(5, runs)
    best = min(best, kern / runs)
    whole = min(whole, tot / runs)
print("%-16s %-28s kernel us %8.1f whole step us %8.1f (best of 3 x %d) | %s slices %d ring %d blocks %d units %d | ablate %s" % (
    name, os.environ.get("TAG", ""), best * 1e3, whole * 1e3, runs, device.STREAM_FORMATS[st["stream_format"]], st["col_slices"], st["ring_buffers"],
    st["num_blocks"], st["num_units"], os.environ.get("HISPARSE_ABLATE", "0")))
print("%-16s load %.1f ms (%s re-tile), image %.1f MB" % (name, st["load_seconds"] * 1e3, "gpu" if st["retiled_on_gpu"] else "host", st["stream_bytes"] / 1e6))
if os.environ.get("PROBE_JSON"):      # tools/profile_cfg.sh: what the profiled image looked like
    import json
    with open(os.environ["PROBE_JSON"], "w") as f:
        json.dump({"config": name, "impl": int(impl), "nnz": int(cp.nnz), "stream_bytes": int(st["stream_bytes"]),
                   "stream_format": device.STREAM_FORMATS[st["stream_format"]], "col_slices": int(st["col_slices"]),
                   "kernel_us_hip_events_best": best * 1e3}, f)
