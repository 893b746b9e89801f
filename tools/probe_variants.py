"""Several builds / environment variants of one configuration in ONE process (the matrix is generated and formatted once):
    python tools/probe_variants.py <config> "TAG:VAR=value,VAR=value" ...      e.g.  ogbn_products "owner24:" "owner:HISPARSE_STREAM_FORMAT=owner"
Variants are measured round-robin, ROUNDS times (default 3), so that clock / box drift hits them alike; prints the best kernel and
whole-step time of every variant and checks every variant's y against the first one's."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _prof

name = sys.argv[1]
variants = []
for spec in sys.argv[2:]:
    tag, _, envs = spec.partition(":")
    variants.append((tag, dict(kv.split("=", 1) for kv in envs.split(",") if kv)))
if any(_prof.needs_profiling_library(env) for _, env in variants) or _prof.needs_profiling_library(os.environ):
    _prof.use_profiling_library()      # HISPARSE_ABLATE / HISPARSE_DEPTH variants live in libhisparse_hip_prof.so only
from hisparse_amd import host, device, datasets
cfg, csr = datasets.load(name)
impl = host.impl_id(os.environ.get("IMPL", cfg.impl))
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
x = np.random.default_rng(0).normal(size=cp.num_cols).astype(np.float32) if impl else np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32)
xw = host.pack_vector(impl, x)
runs, rounds = int(os.environ.get("RUNS", "50")), int(os.environ.get("ROUNDS", "3"))
managed = sorted({k for _, env in variants for k in env})
engines, y0 = [], None
for tag, env in variants:
    for k in managed:
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = device.SpmvEngine(impl)
    eng.load_matrix(cp)
    eng.load_vector(xw)
    eng.run()
    y = eng.read_result()
    if y0 is None:
        y0 = y
    same = np.array_equal(y, y0) if impl == 0 else bool(np.allclose(y.view(np.float32), y0.view(np.float32), rtol=1e-4, atol=1e-4))
    engines.append([tag, env, eng, 1e9, 1e9, same])
for _ in range(rounds):
    for e in engines:
        for k in managed:
            os.environ.pop(k, None)
        os.environ.update(e[1])         # launch-time switches (HISPARSE_DEPTH, HISPARSE_ABLATE) are read per launch
        tot, kern = e[2].time_runs(5, runs)
        e[3], e[4] = min(e[3], kern / runs), min(e[4], tot / runs)
for tag, env, eng, kern, whole, same in engines:
    st = eng.stats()
    print("%-16s %-22s kernel us %8.1f whole step us %8.1f | %-8s %6.1f MB slices %d ring %d blocks %d units %d | y %s" % (
        name, tag, kern * 1e3, whole * 1e3, device.STREAM_FORMATS[st["stream_format"]], st["stream_bytes"] / 1e6, st["col_slices"], st["ring_buffers"],
        st["num_blocks"], st["num_units"], "same as the first variant's" if same else "DIFFERS"), flush=True)
