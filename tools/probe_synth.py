"""PAIRS against DELTA across densities (where should the planner switch?):
python tools/probe_synth.py [impl]   -- power-law matrices (beta 0.3) at mean position gaps 8 ... 4096, kernel us per format"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, device

impl = host.impl_id(sys.argv[1] if len(sys.argv) > 1 else "fixed")
shapes = [(40000, 40000), (400000, 100000)]
for rows, cols in shapes:
    for gap in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
        nnz = rows * cols // gap
        if os.environ.get("ONLY") and os.environ["ONLY"] != "%d,%d,%d" % (rows, cols, gap):
            continue
        if nnz > 150e6 or nnz < 2e6:
            continue
        csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=float(nnz), b=0.3, c=1.0, seed=gap)
        cp = host.format_matrix(csr, impl, skip_empty_rows=True)
        x = host.pack_vector(impl, np.random.default_rng(0).uniform(0, 2, cp.num_cols).astype(np.float32))
        out = []
        for fmt, runs in (("auto", None), ("pairs", None), ("delta", "0"), ("delta", "1")):
            os.environ["HISPARSE_STREAM_FORMAT"] = fmt
            if fmt == "auto":
                os.environ.pop("HISPARSE_STREAM_FORMAT")
            os.environ.pop("HISPARSE_ROW_RUNS", None)
            if runs is not None:
                os.environ["HISPARSE_ROW_RUNS"] = runs
            with device.SpmvEngine(impl) as eng:
                eng.load_matrix(cp)
                eng.load_vector(x)
                st = eng.stats()
                best = min(eng.time_runs(5, 30)[1] / 30 for _ in range(3))
                out.append("%s%s %6.1f (%s %.2f B)" % (fmt, "" if runs is None else " sums=" + runs, best * 1e3, device.STREAM_FORMATS[st["stream_format"]],
                                                         st["stream_bytes"] / max(1, st["nnz"])))
        print("%6d x %6d gap %4d nnz %5.1f M: %s" % (rows, cols, gap, cp.nnz / 1e6, " | ".join(out)), flush=True)
