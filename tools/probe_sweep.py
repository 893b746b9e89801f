"""SWEEP against the planner's row-block choice (OWNER24) on hyper-sparse power-law matrices: where should the planner switch?
python tools/probe_sweep.py [impl ...]   -- square matrices of 1.0 / 1.6 / 2.4 M rows at mean position gaps 25 K ... 200 K (rows x cols / nnz),
whole step (kernel + combine pass, plain back-to-back launches) per format, loaded straight from CSR."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device

impls = sys.argv[1:] or ["fixed", "float_stall"]
sizes = [int(s) for s in os.environ.get("SIZES", "1000000,1600000,2400000").split(",")]
gaps = [int(g) for g in os.environ.get("GAPS", "25000,35000,50000,70000,100000,150000,200000").split(",")]


def step_us(eng, runs=200):
    for _ in range(300):
        eng.run()
    best = 1e9
    for _ in range(3):
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(runs):
            eng.run()
        eng.sync()
        best = min(best, (time.perf_counter() - t0) / runs)
    return best * 1e6


for n in sizes:
    for gap in gaps:
        nnz = n * n // gap
        if nnz > 130e6 or nnz < 4e6:
            continue
        csr = host.CSRMatrix.generate("powerlaw", n, n, a=float(nnz), b=0.4, c=1.0, seed=gap % 1000)
        for impl_name in impls:
            impl = host.impl_id(impl_name)
            out = []
            for fmt in ("auto", "sweep"):
                os.environ.pop("HISPARSE_STREAM_FORMAT", None)
                if fmt != "auto":
                    os.environ["HISPARSE_STREAM_FORMAT"] = fmt
                with device.SpmvEngine(impl) as eng:
                    eng.load_matrix_csr(csr)
                    x = np.random.default_rng(0).uniform(0, 2, eng.num_cols).astype(np.float32)
                    eng.load_vector(host.pack_vector(impl, x))
                    st = eng.stats()
                    t = step_us(eng)
                    out.append("%s %7.1f us (%s, %d slices, %.2f B/nnz, load %.0f ms)" % (fmt, t, device.STREAM_FORMATS[st["stream_format"]], st["col_slices"],
                                                                                 st["stream_bytes"] / max(1, st["nnz"]), st["load_seconds"] * 1e3))
            os.environ.pop("HISPARSE_STREAM_FORMAT", None)
            print("%8d^2 gap %6d nnz %6.1f M (%.1f per row) %-11s: %s" % (n, gap, csr.nnz / 1e6, csr.nnz / n, impl_name, " | ".join(out)), flush=True)
