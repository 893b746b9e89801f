"""Kernel-time probe on any named config: python tools/probe_cfg.py <config> [impl]  (HISPARSE_* env selects variants; an explicit impl
(fixed|float_pob|float_stall) runs the config's matrix in another numeric mode).

This is the command tools/profile_cfg.sh puts under rocprofv3, so it launches the way bench.py's timed loop does (round 4):
the same spin-up (untimed steps until the step time stops improving), then PLAIN back-to-back hs_run calls -- no HIP event pair around a
launch anywhere (a pair adds ~2 us per launch and, under the profiler, read 317 us for a 200 us kernel) -- so that a rocprofv3 kernel
average taken from this process is a kernel inside the same kind of step the driver times, and must come out <= that step.  Reported:
wall-clock step over RUNS launches (best of 3) and the two-events-around-K-launches step; PROBE_PAIRS=1 adds the per-launch event pairs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _prof
if _prof.needs_profiling_library(os.environ):
    _prof.use_profiling_library()
from hisparse_amd import host, device, datasets

name = sys.argv[1]
cfg, csr = datasets.load(name)
impl = host.impl_id(sys.argv[2] if len(sys.argv) > 2 else cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=True)
eng = device.SpmvEngine(impl)
eng.load_matrix(cp)
st = eng.stats()
rng = np.random.default_rng(2024)
x = rng.uniform(0, 2, cp.num_cols).astype(np.float32) if impl == 0 else rng.normal(size=cp.num_cols).astype(np.float32)
eng.load_vector(host.pack_vector(impl, x))
runs = int(os.environ.get("RUNS", "200"))


def spin_up(batch=200, max_batches=40, at_least=1000):      # bench.py: spin_up
    best, flat, n = 1e9, 0, 0
    for _ in range(max_batches):
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(batch):
            eng.run()
        eng.sync()
        t = (time.perf_counter() - t0) / batch
        n += batch
        flat = 0 if t < 0.997 * best else flat + 1
        best = min(best, t)
        if flat >= 3 and n >= at_least:
            break
    return n


spun = spin_up()
wall = 1e9
for _ in range(3):
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(runs):
        eng.run()
    eng.sync()
    wall = min(wall, (time.perf_counter() - t0) / runs)
region_ms, _ = eng.time_runs(0, runs, kernel=False)
pairs = None
if os.environ.get("PROBE_PAIRS"):
    _, kern = eng.time_runs(0, runs)
    pairs = kern / runs * 1e3
nnz = int(cp.nnz)
print("%-16s %-24s step us %8.2f wall (best of 3 x %d) | %8.2f two events around K launches%s | %.1f %% of 8 TB/s whole job | %s slices %d ring %d blocks %d units %d | spin-up %d" % (
    name, os.environ.get("TAG", ""), wall * 1e6, runs, region_ms / runs * 1e3, "" if pairs is None else " | %8.2f kernel, event pairs" % pairs,
    8.0 * nnz / wall / 8e12 * 100, device.STREAM_FORMATS[st["stream_format"]], st["col_slices"], st["ring_buffers"], st["num_blocks"], st["num_units"], spun))
print("%-16s load %.1f ms (%s re-tile), image %.1f MB" % (name, st["load_seconds"] * 1e3, "gpu" if st["retiled_on_gpu"] else "host", st["stream_bytes"] / 1e6))
if os.environ.get("PROBE_JSON"):      # tools/profile_cfg.sh: what the profiled image looked like
    import json
    with open(os.environ["PROBE_JSON"], "w") as f:
        json.dump({"config": name, "impl": int(impl), "nnz": nnz, "stream_bytes": int(st["stream_bytes"]),
                   "stream_format": device.STREAM_FORMATS[st["stream_format"]], "col_slices": int(st["col_slices"]),
                   "step_us_wall_best": wall * 1e6, "step_us_two_events": region_ms / runs * 1e3, "spin_up_steps": spun, "runs": runs}, f)
