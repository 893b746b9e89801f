"""bench_extras.py -- the sweeps behind `bench_details.json` (everything bench.py measures beside its headline line).

bench.py prints ONE short JSON line (the headline configuration, its roofline and CPU baseline).  The other measurements of a default
run live here and are written to `bench_details.json` (+ one row per matrix on stderr, bench.summary_rows):
  * per_config                 the other single-GPU configurations of BASELINE.json (+ the R-MAT stand-in), each with a MALL-cold leg
  * bm_list / bm_list_float    the reference's whole sweep (sw/bm.sh:3-35), fixed point and the float modes of the paper's Table 7
  * strong_scaling_prediction  every row slab of the 2-/4-/8-way split timed on this one GPU (no collective involved or predicted)
Each configuration goes through bench.measure_single, i.e. is checked against the oracle before it is timed.
"""
import os
import time

import bench as B


def spmm_probe(np, host, eng, impl, packets, rng, xw, k=8, reps=100):
    """Dense-row (BITMAP) images: hs_spmm_device with k columns of X resident in HBM -- the fused kernel streams the matrix once per 4
    columns (spmm_bitmap.hip) -- next to k SpMVs; column 0 is checked against the SpMV kernel's own answer (bit for bit)."""
    import ctypes as C
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipFree.argtypes = [C.c_void_p]
    X = rng.normal(size=(k, packets.num_cols)).astype(np.float32) if impl else rng.uniform(0.0, 2.0, (k, packets.num_cols)).astype(np.float32)
    Xw = np.stack([host.pack_vector(impl, X[j]) for j in range(k)])
    xd, yd = C.c_void_p(), C.c_void_p()
    if rt.hipMalloc(C.byref(xd), Xw.nbytes) or rt.hipMalloc(C.byref(yd), k * packets.num_rows * 4) or rt.hipMemcpy(xd, Xw.ctypes.data, Xw.nbytes, 1):
        return None
    try:
        for _ in range(20):
            eng.spmm_device(xd.value, packets.num_cols, yd.value, packets.num_rows, k)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.spmm_device(xd.value, packets.num_cols, yd.value, packets.num_rows, k)
        eng.sync()
        us = (time.perf_counter() - t0) / reps * 1e6
        y0 = np.empty(packets.num_rows, dtype=np.uint32)
        rt.hipMemcpy(y0.ctypes.data, yd, y0.nbytes, 2)
        eng.load_vector(Xw[0])
        eng.run()
        same = bool(np.array_equal(y0, eng.read_result()))
        eng.load_vector(xw)            # the context's own vector as the caller left it
    finally:
        rt.hipFree(xd)
        rt.hipFree(yd)
    return {"k": k, "us_per_spmm": round(us, 2), "us_per_column": round(us / k, 2), "column_0_equals_spmv_bit_for_bit": same,
            "note": "hs_spmm_device, X and Y resident; float BITMAP images: 5-16 columns per pass through the matrix engine (spmm_mfma.hip, sums in another "
                    "order than the SpMV kernel: tolerance parity per column), fixed point: 4 columns per pass (spmm_bitmap.hip, bit for bit); reference: no SpMM"}


def mall_cold(np, datasets, device, host, first, steps, warmup, rank):
    """The same configuration ROUND-ROBIN over several different matrices of the same shape (other seeds), enough of them that their
    stream images add up to >= 640 MB -- two and a half times the 256 MiB Infinity Cache -- so that nothing of an image can be left in
    it when its turn comes again (ogbl-ppa: 4 x 291 MB; transformer-50: 16 x 37 MB; mouse_gene: 4 x 186 MB; ogbn-products: 2 x 877 MB).
    All contexts launch on one stream; the result is a whole-job number (kernel + combine pass, launch gaps included), next to the
    same loop over ONE image.  Images so small that 24 of them stay under 300 MB cannot be cooled this way (reported as such)."""
    cfg = first["cfg"]
    image = max(1, first["eng"].stats()["stream_bytes"])
    count = max(2, min(24, -(-640_000_000 // image)))
    if count * image < 300_000_000 or cfg.kind not in ("powerlaw", "bernoulli", "rmat"):
        return {"images": 1, "frac_whole_job_round_robin": None,
                "note": f"a {image/1e6:.1f} MB image: 24 of them would still fit the 256 MiB Infinity Cache; it lives in the caches by nature"}
    engines, nnzs = [first["eng"]], [first["nnz"]]
    stream = first["eng"].get_stream()
    for k in range(1, count):
        csr = host.CSRMatrix.generate(cfg.kind, cfg.rows, cfg.cols, a=cfg.a, b=cfg.b, c=cfg.c, seed=cfg.seed + 1000 * k)
        eng = device.SpmvEngine(first["impl"])
        eng.load_matrix_csr(csr)                 # (byte for byte the image the CPSR path builds: checked for the first matrix in measure_single)
        eng.load_vector(first["xw"])
        eng.set_stream(stream)
        engines.append(eng)
        nnzs.append(eng.stats()["nnz"])
        del csr
    image_mb = sum(e.stats()["stream_bytes"] for e in engines) / 1e6

    def timed(order):
        for _ in range(max(1, B.SPIN_UP_STEPS // len(order))):
            for e in order:
                e.run()
        first["eng"].sync()
        for i in range(warmup):
            order[i % len(order)].run()
        first["eng"].sync()
        t0 = time.perf_counter()
        for i in range(steps):
            order[i % len(order)].run()
        first["eng"].sync()
        return (time.perf_counter() - t0) / steps

    t_rr = timed(engines)                 # the images in turn
    t_one = timed(engines[:1])            # the same loop over one image (warm Infinity Cache), for comparison on equal terms
    mean_nnz = sum(nnzs[i % count] for i in range(steps)) / steps
    for e in engines[1:]:
        e.set_stream(None)
        e.close()
    B.log(rank, f"{cfg.name}: round-robin over {count} images ({image_mb:.0f} MB): {t_rr*1e6:.1f} us per SpMV; one image: {t_one*1e6:.1f} us")
    return {"images": count, "image_megabytes_total": round(image_mb, 1), "ms_per_step_round_robin": round(t_rr * 1e3, 5),
            "ms_per_step_one_image_same_loop": round(t_one * 1e3, 5),
            "frac_whole_job_round_robin": round(8.0 * mean_nnz / t_rr / 1e9 / B.HBM_PEAK_GBS, 4),
            "frac_whole_job_one_image": round(8.0 * first["nnz"] / t_one / 1e9 / B.HBM_PEAK_GBS, 4)}


def quote_hbm_fraction(res):
    """Which whole-job number may be called "fraction of the HBM roofline": an image below 256 MiB lives in the Infinity Cache between the
    launches of a loop over ONE matrix, so for those the MALL-cold (round-robin over > 256 MiB of images) figure is the HBM number and the
    warm one is a cache number; larger images: the warm loop (they lose 0-2 points cold)."""
    r = res["roofline"]
    small = r["streamed_bytes_per_launch"] < 256 * 2 ** 20
    cold = r.get("frac_mall_cold")
    if small and cold is not None:
        res["hbm_roofline_fraction_quoted"] = cold
        res["hbm_roofline_fraction_quoted_from"] = "whole job, MALL-cold round-robin (the image fits the 256 MiB Infinity Cache: the warm loop is a cache number)"
    else:
        res["hbm_roofline_fraction_quoted"] = res["frac_whole_step"]
        res["hbm_roofline_fraction_quoted_from"] = "whole job, one image" + (" (image below 256 MiB and no cold leg measured: an Infinity-Cache number)" if small else "")


def predict_scaling(np, datasets, device, host, sharding, name, steps, rank, ways=(2, 4, 8)):
    """Strong-scaling evidence that ONE GPU can give: for N in 2, 4, 8 every row slab of the N-way split
    (sharding.split_rows_by_nnz, exactly what rank r of `bench.py --gpus N` loads) is timed on this GPU; the slowest slab bounds the
    N-GPU compute-only step, so  efficiency(N) = t(unsplit) / (N x max slab time).  No collective is involved or predicted."""
    cfg, full = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    granule = 128 * (8 if impl == host.IMPL_FLOAT_STALL else 1)
    indptr, indices, data = full.arrays()
    rng = np.random.default_rng(2024)
    cols8 = (full.num_cols + 7) // 8 * 8
    x = rng.uniform(0.0, 2.0, cols8).astype(np.float32) if impl == host.IMPL_FIXED else rng.normal(size=cols8).astype(np.float32)
    xw = host.pack_vector(impl, x)

    def step_us(csr):
        """(plain: K back-to-back steps enqueued from the library's C loop between two HIP events, best of 3; graph: the same K steps
        replayed from ONE captured hipGraph -- hs_run_batch with batch_graph = 1 -- between two host synchronisations, best of 3)"""
        with device.SpmvEngine(impl) as eng:
            eng.load_matrix_csr(csr)
            eng.load_vector(xw)
            st = eng.stats()
            for _ in range(B.SPIN_UP_STEPS // 2):
                eng.run()
            eng.sync()
            best = 1e9
            for _ in range(3):
                region_ms, _ = eng.time_runs(5, steps, kernel=False)
                best = min(best, region_ms / steps)
            graph = None
            try:
                eng.set_option("batch_graph", "1")
                eng.run_batch(steps)                 # capture + instantiate, untimed
                eng.sync()
                graph = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    eng.run_batch(steps)
                    eng.sync()
                    graph = min(graph, (time.perf_counter() - t0) / steps * 1e3)
            except Exception as e:      # noqa: BLE001 -- the graph leg is an extra measurement
                B.log(rank, f"graph replay skipped: {e}")
            return best * 1e3, (graph * 1e3 if graph is not None else None), st

    t_whole, t_whole_graph, st_whole = step_us(full)
    out = {"workload": f"{name}, {B.IMPL_NAMES[impl]} IMPL", "nnz": int(full.nnz), "unsplit_us": round(t_whole, 2),
           "unsplit_us_graph": round(t_whole_graph, 2) if t_whole_graph else None,
           "unsplit_plan": f"{device.STREAM_FORMATS[st_whole['stream_format']]}, {st_whole['col_slices']} slices, {st_whole['num_blocks']} blocks", "splits": []}
    for n in ways:
        bounds = sharding.split_rows_by_nnz(indptr, n, granule)
        slabs = []
        for r in range(n):
            lo, hi = bounds[r], bounds[r + 1]
            if hi == lo:
                continue
            ip, ix, dv = sharding.slab_arrays(indptr, indices, data, lo, hi)
            t, tg, st = step_us(host.CSRMatrix.from_arrays(hi - lo, full.num_cols, ip, ix, dv))
            slabs.append({"rank": r, "rows": int(hi - lo), "nnz": int(ip[-1]), "us": round(t, 2), "us_graph": round(tg, 2) if tg else None,
                          "plan": f"{device.STREAM_FORMATS[st['stream_format']]}, {st['col_slices']} slices, {st['num_blocks']} blocks"})
        worst = max(s["us"] for s in slabs)
        split = {}
        if t_whole_graph and all(s["us_graph"] for s in slabs):      # every step replayed from a captured hipGraph: the host's enqueue rate is out of the picture
            worst_graph = max(s["us_graph"] for s in slabs)
            split = {"max_slab_us_graph": worst_graph, "predicted_compute_only_efficiency_graph": round(min(t_whole, t_whole_graph) / (n * worst_graph), 4)}
        out["splits"].append({"n_gpus": n, "max_slab_us": worst, "mean_slab_us": round(sum(s["us"] for s in slabs) / len(slabs), 2),
                              "predicted_compute_only_efficiency": round(t_whole / (n * worst), 4), **split,
                              "roofline_us_per_slab": round(8.0 * full.nnz / n / (B.HBM_PEAK_GBS * 1e9) * 1e6, 2), "slabs": slabs})
        B.log(rank, f"{name} split {n} ways: slowest slab {worst:.1f} us against {t_whole:.1f} us unsplit -> predicted compute-only efficiency {t_whole / (n * worst) * 100:.0f} %")
    return out



def live_traffic(name, impl_name, kernel, rank, launches=30, timeout=150):
    """HBM bytes per launch of the dominant kernel, MEASURED in this run: two rocprofv3 counter passes (their own runs, --kernel-trace
    only beside --pmc, as MI355X_MICROARCH.md prescribes) of tools/traffic_probe.py -- the same configuration, loaded the same way, a few
    launches -- FETCH_SIZE x 1024 x 2 (gfx950 counts the 128-byte requests of a wide streaming read at 64 bytes) + WRITE_SIZE x 1024,
    medians over the launches.  (bytes, provenance) or (None, why): the caller then falls back to the committed copy."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import sys
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="hisparse_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    medians = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            sub = os.path.join(out, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", sub, "-o", counter.lower(), "--", sys.executable,
                   os.path.join(B.ROOT, "tools", "traffic_probe.py"), name, impl_name, str(launches)]
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd="/tmp")
            hits = glob.glob(os.path.join(sub, "**", "*.db"), recursive=True)
            if p.returncode != 0 or not hits:
                return None, f"rocprofv3 --pmc {counter} failed (exit {p.returncode})"
            vals = sorted(v for (v,) in sqlite3.connect(hits[0]).execute(
                "select value from counters_collection where counter_name = ? and kernel_name like ?", (counter, f"%{kernel}%")))
            if len(vals) < launches // 2:
                return None, f"rocprofv3 --pmc {counter}: {len(vals)} samples of {kernel}"
            medians[counter] = vals[len(vals) // 2]
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error) as e:
        return None, f"{type(e).__name__}: {e}"[:160]
    finally:
        shutil.rmtree(out, ignore_errors=True)
    total = medians["FETCH_SIZE"] * 1024.0 * 2.0 + medians["WRITE_SIZE"] * 1024.0
    B.log(rank, f"{name}: HBM traffic per launch measured in this run: FETCH_SIZE {medians['FETCH_SIZE']:.0f} KiB x 2 + WRITE_SIZE {medians['WRITE_SIZE']:.0f} KiB = {total/1e6:.1f} MB")
    return total, (f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over tools/traffic_probe.py, medians of {launches} launches; "
                   "FETCH_SIZE KiB x 1024 x 2 (gfx950 half-count of wide streaming reads) + WRITE_SIZE KiB x 1024")


def live_kernel_trace(name, impl_name, kernel, rank, launches=400, timeout=150):
    """The dominant kernel's launch duration as rocprofv3 sees it, IN this run: one `rocprofv3 --kernel-trace --stats` pass (no counters) over
    tools/traffic_probe.py in batch mode -- the same configuration, `launches` back-to-back launches from hs_run_batch.  Returns
    ({"avg_us": average over ALL its dispatches in that process -- the figure `--stats` prints --, "steady_median_us": median of the second
    half, "calls": n, "combine_avg_us": the slice-combine kernel's average or None}, how) or (None, why)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import sys
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="hisparse_ktrace_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--stats", "-d", out, "-o", "stats", "--", sys.executable, os.path.join(B.ROOT, "tools", "traffic_probe.py"),
               name, impl_name, str(launches), "batch"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        hits = glob.glob(os.path.join(out, "**", "*.db"), recursive=True)
        if p.returncode != 0 or not hits:
            return None, f"rocprofv3 --kernel-trace --stats failed (exit {p.returncode})"
        d = sqlite3.connect(hits[0])
        durs = [(e - s) / 1000.0 for s, e in d.execute("select start, end from kernels where name like ? order by start", (f"%{kernel}%",))]
        if len(durs) < launches // 2:
            return None, f"rocprofv3 --kernel-trace: {len(durs)} dispatches of {kernel}"
        comb = [(e - s) / 1000.0 for s, e in d.execute("select start, end from kernels where name like '%combine_slices_kernel%'")]
        steady = sorted(durs[len(durs) // 2:])
        res = {"avg_us": round(sum(durs) / len(durs), 3), "steady_median_us": round(steady[len(steady) // 2], 3), "calls": len(durs),
               "combine_avg_us": round(sum(comb) / len(comb), 3) if comb else None}
    except (subprocess.TimeoutExpired, OSError, sqlite3.Error) as e:
        return None, f"{type(e).__name__}: {e}"[:160]
    finally:
        shutil.rmtree(out, ignore_errors=True)
    B.log(rank, f"{name}: rocprofv3 --kernel-trace --stats in this run: {kernel} average {res['avg_us']:.2f} us over {res['calls']} dispatches, steady median {res['steady_median_us']:.2f} us"
                + (f"; combine_slices_kernel {res['combine_avg_us']:.2f} us" if res["combine_avg_us"] else ""))
    return res, f"rocprofv3 --kernel-trace --stats over tools/traffic_probe.py ({launches} back-to-back launches in 4 batches), this run"


def bm_entry(name, paper_gops, res, impl="fixed"):
    """one line of the reference's sweep (sw/bm.sh) next to the paper's U280 figure for the same matrix and numeric mode (Table 3: fixed
    point; Table 7: float_pob = "PB", float_stall = "RI")"""
    r = res["roofline"]
    row = {"matrix": name, "impl": impl, "nnz": res["nnz"], "partitions": res["partitions"], "stream_format": res["stream_format"], "col_slices": res["col_slices"],
           "ms_per_step": res["ms_per_step"], "ms_per_step_long_run": res.get("ms_per_step_long_run"), "long_run_steps": res.get("long_run_steps"),
           "ms_per_step_synchronous": res["ms_per_step_synchronous"], "value": res["value"], "unit": "GB/s", "gops": res["gops"],
           "frac_whole_step": res["frac_whole_step"], "frac": r["frac"], "frac_event_pairs": r["frac_event_pairs"], "frac_mall_cold": r.get("frac_mall_cold"),
           "kernel_ms": r["kernel_ms"], "frac_steady": r.get("frac_steady"), "kernel_ms_steady": r.get("kernel_ms_steady"), "launches_per_step": r.get("launches_per_step"),
           "streamed_bytes_per_launch": r["streamed_bytes_per_launch"],
           "image_fits_infinity_cache": r["streamed_bytes_per_launch"] < 256 * 2 ** 20, "parity_vs_oracle": res["parity_vs_oracle"],
           "paper_gops_u280": paper_gops, "paper_table": "Table 3" if impl == "fixed" else "Table 7",
           "gops_vs_paper": round(res["gops"] / paper_gops, 1) if paper_gops else None}
    if "float_error" in res:
        row["float_error"] = res["float_error"]
    return row


def run_suite(np, datasets, device, host, sharding, steps, warmup, rank):
    """(details dict, {(matrix, impl): result}) of a default N = 1 run, measured BEFORE the headline configuration."""
    per_config, bm_rows, measured = [], {}, {}
    # the three other single-GPU configurations of BASELINE.json + the second ogbl-ppa stand-in (symmetric R-MAT),
    # each also round-robin over enough images to be Infinity-Cache-cold
    for name in ("transformer_50", "ogbn_products", "mouse_gene", "ogbl_ppa_rmat"):
        res, ctx = B.measure_single(np, datasets, device, host, name, steps, warmup, rank=rank)
        cold = mall_cold(np, datasets, device, host, ctx, steps, warmup, rank)
        res["roofline"]["frac_mall_cold"] = cold["frac_whole_job_round_robin"]
        res["mall_cold"] = cold
        quote_hbm_fraction(res)
        ctx["eng"].close()
        del ctx
        per_config.append(res)
        if name == "mouse_gene":
            bm_rows[name] = res
        measured[(name, res["impl"])] = res
    # the rest of the reference's sweep (sw/bm.sh:3-17), in the numeric mode of the paper's Table 3
    for name, _ in datasets.BM_LIST:
        if name in ("ogbl_ppa", "mouse_gene"):
            continue
        res, ctx = B.measure_single(np, datasets, device, host, name, steps, warmup, impl_override="fixed", rank=rank, with_spmm=False)
        ctx["eng"].close()
        del ctx
        bm_rows[name] = res
        measured[(name, "fixed")] = res
    # ... and the float modes of the sweep on the matrices the paper quotes in all three (Table 7; sw/bm.sh:19-35)
    float_rows = []
    for name, fx, pb, ri in datasets.BM_FLOAT:
        for impl_name, paper in (("float_pob", pb), ("float_stall", ri)):
            res = measured.get((name, impl_name))
            if res is None:
                res, ctx = B.measure_single(np, datasets, device, host, name, steps, warmup, impl_override=impl_name, rank=rank, with_spmm=False)
                ctx["eng"].close()
                del ctx
                measured[(name, impl_name)] = res
            row = bm_entry(name, paper, res, impl_name)
            fixed = measured.get((name, "fixed"))
            if fixed is not None:
                row["points_vs_fixed_point"] = round((res["frac_whole_step"] - fixed["frac_whole_step"]) * 100, 1)
            float_rows.append(row)
    scaling = [predict_scaling(np, datasets, device, host, sharding, "mouse_gene", steps, rank)]
    # the larger graphs, where row slabs are still 15-25 us of streaming (8-way split only: 1 + 8 loads each)
    scaling += [predict_scaling(np, datasets, device, host, sharding, name, steps, rank, ways=(8,)) for name in ("hollywood", "ogbn_products")]
    return {"per_config": per_config, "bm_rows": bm_rows, "bm_list_float": float_rows, "strong_scaling_prediction": scaling}, measured
