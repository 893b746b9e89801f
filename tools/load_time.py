"""hs_load_matrix wall time on a named config, three loads in one process (the first pays the code-object loads):
python tools/load_time.py <config>   (HISPARSE_PLAN_DEBUG=1 prints the phases, HISPARSE_RETILE=host the host builder)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device, datasets

name = sys.argv[1]
cfg, csr = datasets.load(name)
impl = host.impl_id(cfg.impl)
cp = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
if os.environ.get("LOAD_CSR"):       # hs_load_matrix_csr: no csr2cpsr at all
    with device.SpmvEngine(impl) as eng:
        for k in range(3):
            print("---- CSR load %d" % k, file=sys.stderr, flush=True)
            t0 = time.perf_counter()
            eng.load_matrix_csr(csr)
            wall = time.perf_counter() - t0
            st = eng.stats()
            print("%-16s CSR load %d: %.1f ms (python wall %.1f ms incl. the array copies out of the CSR handle), %s, image %.1f MB" % (
                name, k, st["load_seconds"] * 1e3, wall * 1e3, device.STREAM_FORMATS[st["stream_format"]], st["stream_bytes"] / 1e6), flush=True)
    sys.exit(0)
with device.SpmvEngine(impl) as eng:
    for k in range(3):
        print("---- load %d" % k, file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        eng.load_matrix(cp)
        wall = time.perf_counter() - t0
        st = eng.stats()
        print("%-16s load %d: %.1f ms (python wall %.1f ms), %s re-tile, %s, image %.1f MB, CPSR %.1f MB" % (
            name, k, st["load_seconds"] * 1e3, wall * 1e3, "gpu" if st["retiled_on_gpu"] else "host",
            device.STREAM_FORMATS[st["stream_format"]], st["stream_bytes"] / 1e6, st["cpsr_bytes"] / 1e6), flush=True)
