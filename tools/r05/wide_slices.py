"""round 5: a wide matrix of a few million non-zeros (60 000 x 300 000, 37 sub-tiles) under the slice counts the planner may now take"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hisparse_amd import host, device
for rows, cols, nnz, seed in ((60000, 300000, 3.0e6, 9), (120000, 300000, 8.0e6, 10), (200000, 600000, 2.0e7, 11)):
    csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=nnz, b=0.3, c=1.0, seed=seed)
    x = host.pack_vector(0, np.random.default_rng(1).uniform(0, 2, (cols + 7) // 8 * 8).astype(np.float32))
    for tag, env in (("auto", {}), ("pow2", {"HISPARSE_POW2_SLICES": "1"}), ("cs4", {"HISPARSE_COL_SLICES": "4"}), ("cs5", {"HISPARSE_COL_SLICES": "5"}), ("cs7", {"HISPARSE_COL_SLICES": "7"}), ("cs8", {"HISPARSE_COL_SLICES": "8"})):
        for k in ("HISPARSE_POW2_SLICES", "HISPARSE_COL_SLICES"):
            os.environ.pop(k, None)
        os.environ.update(env)
        os.environ["HISPARSE_LIGHT"] = "0"
        with device.SpmvEngine(0) as eng:
            eng.load_matrix_csr(csr)
            eng.load_vector(x)
            st = eng.stats()
            for _ in range(300):
                eng.run()
            eng.sync()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                eng.run_batch(300)
                eng.sync()
                best = min(best, (time.perf_counter() - t0) / 300)
        print(f"{rows}x{cols} nnz {int(nnz)} {tag:5s} {device.STREAM_FORMATS[st['stream_format']]:6s} slices {st['col_slices']} blocks {st['num_blocks']} step {best * 1e6:7.2f} us", flush=True)
