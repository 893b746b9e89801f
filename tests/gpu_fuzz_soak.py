"""Long seeded fuzz on one GPU: random shapes / densities / bank sizes / numeric modes / stream formats, every result compared
with the oracle; prints the parameters of any failing case.  usage: [FUZZ_PROFILE=dense|large|structured] python tests/gpu_fuzz_soak.py [cases] [seed]"""
import os, sys, time
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hisparse_amd import device, host
from oracle import oracle as orc
import cases

def run(n_cases=200, seed=1, profile=None, verbose=True):
    """(failing cases, cases re-tiled on the GPU).  profile: None | \"dense\" | \"large\" (FUZZ_PROFILE)."""
    rng = np.random.default_rng(seed)
    saved = {k: os.environ.get(k) for k in ("HISPARSE_STREAM_FORMAT", "HISPARSE_ROW_RUNS", "HISPARSE_COL_SLICES")}
    fails = 0
    on_gpu = 0
    t0 = time.time()
    for case in range(n_cases):
        impl = int(rng.integers(0, 3))
        if profile == "dense":      # few long rows: heavy same-accumulator traffic
            rows = int(rng.integers(1, 2000)); cols = int(rng.integers(64, 5000))
            density = float(rng.choice([0.1, 0.3, 0.6]))
        elif profile == "structured":   # round 6: banded / block-diagonal / hub rows / few long rows over many columns -- the plans the tile census, the hub-row
            rows = int(rng.integers(2000, 90000)); cols = int(rng.integers(2000, 90000))      # rule and the tiny-unit rule pick, under random banks (several row / column
            density = float(rng.choice([0.0003, 0.002, 0.01]))                                # partitions), slices and formats, device image against the host builder's
        elif profile == "large":    # several row blocks per workgroup, column slices, bridges
            rows = int(rng.integers(20000, 120000)); cols = int(rng.integers(20000, 120000))
            density = float(rng.choice([0.00002, 0.0001, 0.0005]))
        else:
            rows = int(rng.integers(1, 6000)); cols = int(rng.integers(1, 6000))
            density = float(rng.choice([0.0005, 0.003, 0.02, 0.15]))
        vb = int(rng.choice([1, 2, 16, 64, 4096])); ob = int(rng.choice([1, 2, 8, 64])) * (8 if impl == 2 else 1)
        skip = bool(rng.integers(0, 2))
        fmt = str(rng.choice(["pairs", "delta", "bitmap", "owner", "owner24", "sweep", "auto", "auto"]))      # auto: the planner's own choice of format (and, with slices "", of the tile plan)
        if os.environ.get("HISPARSE_STREAM_FORMAT_ONLY"):      # a whole run in one format (tools/history/r04/long_soak2.sh: sweep)
            fmt = os.environ["HISPARSE_STREAM_FORMAT_ONLY"]
        if fmt.startswith("owner") and density > 0.003:
            fmt = "pairs"      # OWNER sums in fp32 like csim; forced onto long rows its rounding (not a defect) would trip the float64 re-check
        runs = str(rng.choice(["", "0", "1"])); slices = str(rng.choice(["", "", "2", "3", "4", "5"]))
        if fmt == "auto": os.environ.pop("HISPARSE_STREAM_FORMAT", None)
        else: os.environ["HISPARSE_STREAM_FORMAT"] = fmt
        for k, v in (("HISPARSE_ROW_RUNS", runs), ("HISPARSE_COL_SLICES", slices)):
            if v: os.environ[k] = v
            else: os.environ.pop(k, None)
        seed = int(rng.integers(0, 1 << 30))
        if profile == "large":      # the C++ generator: scipy.sparse.random takes minutes at this size
            vb, ob = host.default_banks(impl)
            csr = host.CSRMatrix.generate("powerlaw", rows, cols, a=max(1.0, rows * cols * density), b=float(rng.choice([0.0, 0.35, 0.7])),
                                          c=1.0 if impl == 0 else 2.0, seed=seed)
            ip, ix, dv = csr.arrays()
            m = sp.csr_matrix((dv, ix.astype(np.int64), ip.astype(np.int64)), shape=(rows, cols))
            cp = host.format_matrix(csr, impl, vb_bank=vb, ob_bank=ob, skip_empty_rows=skip)
        elif profile == "structured":
            srng = np.random.default_rng(seed)
            kind = str(srng.choice(["banded", "blockdiag", "hubs", "wide"]))
            per_row = max(1, int(density * cols))
            if kind == "wide":
                rows = int(srng.integers(8, 600)); cols = int(srng.integers(200000, 1500000)); per_row = int(srng.integers(50, 800))
            r = np.repeat(np.arange(rows, dtype=np.int64), per_row)
            if kind == "banded":
                half = int(srng.integers(8, max(9, cols // 8)))
                c = np.clip(r * cols // max(1, rows) + srng.integers(-half, half + 1, r.size), 0, cols - 1)
            elif kind == "blockdiag":
                blk = int(srng.choice([32, 64, 256, 1024]))
                c = np.minimum((r * cols // max(1, rows)) // blk * blk + srng.integers(0, blk, r.size), cols - 1)
            else:
                c = srng.integers(0, cols, r.size)
            if kind == "hubs":
                hub = np.repeat(srng.choice(rows, int(srng.integers(1, 6)), replace=False), int(srng.integers(2000, min(cols, 40000))))
                r = np.concatenate([r, hub]); c = np.concatenate([c, srng.integers(0, cols, hub.size)])
            key = np.unique(r * cols + c)
            vals = (srng.uniform(0.0, 1.0, key.size) if impl == 0 else srng.normal(size=key.size) * 0.5).astype(np.float32)
            m = sp.csr_matrix((vals, (key // cols, key % cols)), shape=(rows, cols))
            m.sort_indices()
            if kind == "wide":
                vb, ob = host.default_banks(impl)
            else:      # (bank sizes that keep the partition grid of a 90 K-column matrix in the thousands)
                vb = int(srng.choice([16, 64, 4096])); ob = int(srng.choice([2, 8, 64])) * (8 if impl == 2 else 1)
            _, cp = cases.formatted(m, impl, vb, ob, skip)
        else:
            m = cases.random_csr(rows, cols, density, seed, impl)
            _, cp = cases.formatted(m, impl, vb, ob, skip)
        xw = host.pack_vector(impl, cases.random_x(cp.num_cols, seed, impl))
        want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
        try:
            eng = device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank)
            eng.load_matrix(cp)
        except device.DeviceError as e:
            if "fewer" in str(e) or "plan" in str(e):
                continue
            raise
        eng.load_vector(xw)
        bad_runs = []
        exact = None
        for r in range(4):
            eng.run()
            got = eng.read_result()
            ok = np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
            if not ok and impl != 0:
                # csim's fp32 running sum is itself off by more than 1e-4 on long rows with cancellation: when the GPU (double sums)
                # agrees with the exact float64 product, the disagreement is the oracle's rounding, not a device error
                exact = np.zeros(cp.num_rows)
                exact[:rows] = m.astype(np.float64) @ xw.view(np.float32)[:cols].astype(np.float64)
                ok = np.allclose(got.view(np.float32).astype(np.float64), exact, rtol=1e-4, atol=1e-4)
            if not ok:
                bad = np.nonzero(got != want)[0] if impl == 0 else np.nonzero(~np.isclose(got.view(np.float32), want.view(np.float32), rtol=1e-4, atol=1e-4))[0]
                bad_runs.append((r, len(bad), bad[:6].tolist()))
        # consecutive launches without anything in between: column-sliced plans carry the combine pass of one step into the next step's
        # kernel (round 5, hs_api.cpp: enqueue / flush_combine) -- every burst length ends on either set of partial vectors
        for burst in (2, 3):
            for _ in range(burst):
                eng.run()
            got = eng.read_result()
            ok = np.array_equal(got, want) if impl == 0 else (cases.float_close(got, want) or
                                                              (exact is not None and np.allclose(got.view(np.float32).astype(np.float64), exact, rtol=1e-4, atol=1e-4)))
            if not ok:
                bad_runs.append((f"burst of {burst}", int((got != want).sum()), []))
        # the reference's literal launch sequence, one row partition at a time, must give the same vector
        for j in range(cp.num_row_partitions):
            eng.run_partition(j, cp.part_len(j))
        got = eng.read_result()
        ok = np.array_equal(got, want) if impl == 0 else (cases.float_close(got, want) or
                                                          (exact is not None and np.allclose(got.view(np.float32).astype(np.float64), exact, rtol=1e-4, atol=1e-4)))
        if not ok:
            bad_runs.append(("partitions", int((got != want).sum()), []))
        st = eng.stats()
        if st["retiled_on_gpu"]:
            # the image the device built against the host builder's, byte for byte
            on_gpu += 1
            dev = eng.read_tiles()
            ref = device.build_tiles(cp, impl, cp.ob_bank, cp.vb_bank, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                                     st["num_compute_units"])
            for part in ("image", "blocks", "units"):
                if dev[part].tobytes() != ref[part].tobytes():
                    bad_runs.append(("gpu re-tile: " + part + " differs from the host builder's", 0, []))
        eng.close()
        if bad_runs:
            fails += 1
            print(f"FAIL case {case}: impl {impl} {rows}x{cols} density {density} vb {vb} ob {ob} skip {skip} fmt {fmt} runs '{runs}' slices '{slices}' seed {seed} "
                  f"blocks {st['num_blocks']} units {st['num_units']} cs {st['col_slices']} ring {st['ring_buffers']}: {bad_runs}", flush=True)
    for k, v in saved.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    if verbose:
        print(f"{n_cases} cases ({on_gpu} re-tiled on the GPU and compared with the host builder), {fails} failing, {time.time() - t0:.0f} s")
    return fails, on_gpu



if __name__ == "__main__":
    f, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 1, os.environ.get("FUZZ_PROFILE"))
    sys.exit(1 if f else 0)
