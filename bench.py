#!/usr/bin/env python3
"""bench.py — SpMV throughput of the MI355X hot path on the reference's benchmark matrices.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--npz FILE]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
          or simply `python bench.py --gpus N`: without a launcher's WORLD_SIZE in the environment the script starts its N ranks itself
          (the same torch.distributed.run command line, master on 127.0.0.1) and fails loudly -- a JSON error line, exit code 2 --
          when the machine shows fewer than N GPUs.  It never prints an `n_gpus: 1` line for a `--gpus N > 1` request.

A "step" is one full SpMV (every row partition) y = A x through the drop-in C-ABI, with the matrix (re-tiled at load
time), x and y already resident in HBM.  Metric (BASELINE.json): the reference's "data throughput" of
sw/benchmark.cpp:312-346 — 8 bytes per non-zero per SpMV — in decimal GB/s, plus GOPS (2 flops per non-zero), the GiB-based
number the reference prints, and the fraction of the 8 TB/s HBM roofline.

N = 1 (default): the headline line is BASELINE.json configs[1] — ogbl-ppa (seeded stand-in, hisparse_amd/datasets.py),
fixed-point IMPL, default banks — and it is the LAST line printed.  In front of it the same process measures the other
single-GPU configurations (transformer-50 / float_pob, ogbn-products / float_stall, mouse_gene / fixed: each with its
parity check against the oracle) and embeds them as `per_config`, and re-measures the headline matrix ROUND-ROBIN over
four different ogbl-ppa-sized matrices (1.1 GB of images, more than the 256 MiB Infinity Cache holds) as
`roofline.frac_mall_cold`.  `--config NAME` measures only that configuration; `--quick` skips the extras.

N > 1 (BASELINE.json configs[4]): mouse_gene, ONE matrix split into N row slabs by non-zero count (`--scaling strong`;
hisparse_amd/sharding.py), every rank formats and loads its slab and holds all of x.  Timed = `value`: K slab SpMVs and ONE final
all-gather of the y slabs over RCCL (BASELINE.json's north_star: "a final RCCL gather over xGMI"; the reference too runs its NUM_RUNS
launches and collects y once, sw/benchmark.cpp:318-346).  Timed alongside and reported in the same line: the same K SpMVs with y left
sharded (`compute_only`), with an all-gather after EVERY SpMV as an iterative caller needs it (`exchange_every_step`: the gather of step
k overlaps the SpMV of step k+1, double-buffered), and with that per-step gather done by peer stores instead of a collective
(`exchange_push`).  `--scaling weak` gives every rank a slab the size of the whole N = 1 matrix instead; `--gather step|off` makes one
of the other patterns the `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
IMPL_NAMES = ["fixed", "float_pob", "float_stall"]
SPIN_UP_STEPS = 1000    # untimed, at least: the clocks have dropped during the CPU legs (oracle, formatting)


def spin_up(eng, batch=200, max_batches=40):
    """Untimed steps until the step time stops improving: after seconds of CPU work the GPU's clocks come up over tens of milliseconds,
    and a short timed region (the driver's run times 20 steps = 1 ms) would otherwise read a few per cent slow.  Returns the steps run."""
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
        if flat >= 3 and n >= SPIN_UP_STEPS:
            break
    return n


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def ensure_built():
    need = [os.path.join(ROOT, "hisparse_amd", "lib", n) for n in ("libhisparse_host.so", "libhisparse_hip.so")]
    need.append(os.path.join(ROOT, "oracle", "liboracle.so"))
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


def sources_sha16():
    """sha256[:16] over the kernel and tiler sources: profiles/hbm_traffic.json records it, so a counter result taken on another build shows."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "hisparse_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".cpp", ".h")):
            with open(os.path.join(csrc, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def read_traffic(config, stream_bytes, impl=None):
    """(HBM bytes per launch, provenance) of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json:
    FETCH_SIZE x 2 + WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes).  The counters need their own rocprofv3 passes, so this
    is a COPY of a committed measurement, not something measured in this run: the provenance says which round, git commit and source
    hash it was taken at and whether the sources have changed since.  (None, reason) when there is no entry for this configuration or the
    entry was taken with a different stream image (the kernel or the tiler changed since)."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None, {"source": "profiles/hbm_traffic.json missing"}
    e = None
    for key in (config, f"{config}:{IMPL_NAMES[impl]}" if impl is not None else None):      # an entry counts for the numeric mode it was profiled in
        c = t.get(key) if key and isinstance(t.get(key), dict) else None
        if c and (impl is None or c.get("impl") is None or int(c.get("impl")) == int(impl)):
            e = c
            break
    if not e:
        return None, {"source": "profiles/hbm_traffic.json has no entry for this configuration and numeric mode"}
    prov = {"source": "profiles/hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, copied, not measured in this run)",
            "round": e.get("round"), "git_head": e.get("git_head"), "csrc_sha16": e.get("csrc_sha16"),
            "sources_unchanged_since": e.get("csrc_sha16") == sources_sha16(), "profiled_stream_bytes": e.get("stream_bytes")}
    ref = e.get("stream_bytes")
    if ref and abs(ref - stream_bytes) > 0.02 * stream_bytes:
        prov["source"] += "; STALE: profiled on another stream image"
        return None, prov
    # the committed rocprofv3 --kernel-trace --stats pass of the same command (tools/profile_cfg.sh): the dominant kernel's average
    # duration and the fraction it prices, next to this run's HIP-event figure
    prov["kernel_us_rocprof"] = e.get("kernel_avg_us")
    prov["frac_rocprof"] = round(e["roofline_frac_rocprof"], 4) if e.get("roofline_frac_rocprof") else None
    prov["step_us_in_the_profiled_process"] = e.get("step_us_wall_best")
    return e.get("hbm_bytes_per_launch"), prov


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


def oracle_check(np, host, impl, packets, xw, y_gpu, seconds, exact=None):
    """(parity string, y of the oracle, seconds per oracle SpMV, repetitions, float error report or None) — oracle/cpu_ref.c, one
    thread, the same channel buffers.  exact: the float64 product (float modes), for the error of both against the exact result."""
    from oracle import oracle as orc
    chans = [packets.channel_ptr(c)[0] for c in range(16)]
    args = (impl, chans, xw, packets.num_rows, packets.num_cols, packets.num_row_partitions, packets.num_col_partitions, packets.ob_bank, packets.vb_bank)
    t0 = time.perf_counter()
    y_cpu = orc.spmv(*args)
    t_one = time.perf_counter() - t0
    reps = max(1, min(20, int(seconds / max(t_one, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps - 1):
        orc.spmv(*args)
    t_cpu = (time.perf_counter() - t0 + t_one) / reps
    report = None
    if impl == host.IMPL_FIXED:
        parity = "bit-exact" if np.array_equal(y_gpu, y_cpu) else "MISMATCH"
    else:
        a, b = y_gpu.view(np.float32).astype(np.float64), y_cpu.view(np.float32).astype(np.float64)
        err = np.abs(a - b)
        rel_ok = bool((err <= 1e-4 * np.maximum(1.0, np.abs(b))).all())     # 1e-4 * max(1, |y_csim|) -- north_star's "1e-4 rel-err"
        abs_ok = bool((err <= 1e-4).all())                                  # csim's own verify: absolute 1e-4 (spmv_csim/csim.cpp:160-175)
        report = {"max_abs_err_vs_csim": float(err.max()), "max_abs_y": float(np.abs(b).max()),
                  "meets_relative_1e-4_times_max(1,|y|)": rel_ok, "meets_csim_absolute_1e-4": abs_ok, "rows_over_csim_absolute_1e-4": int((err > 1e-4).sum())}
        if exact is not None:
            n = exact.size
            report["max_abs_err_gpu_vs_float64"] = float(np.abs(a[:n] - exact).max())
            report["max_abs_err_csim_vs_float64"] = float(np.abs(b[:n] - exact).max())
        which = "relative 1e-4*max(1,|y|): met; csim's absolute 1e-4: " + ("met" if abs_ok else f"not met on {report['rows_over_csim_absolute_1e-4']} rows")
        parity = f"within tolerance ({which}; max abs err {err.max():.2e} at max |y| {np.abs(b).max():.1f})" if rel_ok else "MISMATCH"
    return parity, y_cpu, t_cpu, reps, report


PARITY_PINS = ("oracle/cpu_ref.c is pinned by the reference's own vectors where it holds any (formatter goldens of unit_tests/test_io.cpp, csim's integer "
               "known answers); AP_RND / AP_SAT / float->fixed follow the documented ap_ufixed<32,8,AP_RND,AP_SAT> semantics and are pinned only by "
               "this repository's three independent restatements -- the reference holds no vector for them")


def measure_single(np, datasets, device, host, name, steps, warmup, device_id=0, npz=None, impl_override=None, cpu_seconds=0.0, rank=0, with_spmm=True):
    """One configuration on one GPU: load, parity against the oracle, K timed steps, HIP-event kernel time."""
    t0 = time.perf_counter()
    cfg, csr = datasets.load(name, path=npz)
    impl = host.impl_id(impl_override or cfg.impl)
    true_rows = csr.num_rows
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    packets = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
    t_fmt = time.perf_counter() - t0
    nnz = packets.nnz
    rng = np.random.default_rng(2024)
    # x: uniform [0, 2) for fixed point (the reference uses rand() % 2; random values keep the clocks honest), N(0,1) for float
    x = rng.uniform(0.0, 2.0, packets.num_cols).astype(np.float32) if impl == host.IMPL_FIXED else rng.normal(size=packets.num_cols).astype(np.float32)
    xw = host.pack_vector(impl, x)
    eng = device.SpmvEngine(impl, device_id=device_id)
    eng.load_matrix(packets)
    eng.load_vector(xw)
    stats = eng.stats()
    log(rank, f"{name}: {packets.num_rows}x{packets.num_cols}, nnz {nnz}, partitions {packets.num_row_partitions}x{packets.num_col_partitions}, "
              f"generate {t_gen:.2f}s format {t_fmt:.2f}s device-load {stats['load_seconds']:.2f}s, CPSR {stats['cpsr_bytes']/1e6:.0f} MB -> "
              f"{device.STREAM_FORMATS[stats['stream_format']]} stream {stats['stream_bytes']/1e6:.0f} MB")
    eng.run()
    y_gpu = eng.read_result()
    exact = None
    if impl != host.IMPL_FIXED:      # float modes: the exact (float64) product, to place the GPU's and csim's rounding errors side by side
        import scipy.sparse as sp
        ip, ix, dv = csr.arrays()
        exact = sp.csr_matrix((dv.astype(np.float64), ix.astype(np.int64), ip.astype(np.int64)), shape=(csr.num_rows, csr.num_cols)) @ x[:csr.num_cols].astype(np.float64)
        del ip, ix, dv
    parity, y_cpu, t_cpu, reps, float_error = oracle_check(np, host, impl, packets, xw, y_gpu, cpu_seconds, exact)
    log(rank, f"{name}: oracle {t_cpu*1e3:.1f} ms per SpMV on 1 core; GPU result {parity}")
    if parity == "MISMATCH":
        print(json.dumps({"error": "GPU result does not match the oracle", "config": name}))
        sys.exit(1)
    spun = spin_up(eng)
    for _ in range(warmup):
        eng.run()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run()
    eng.sync()
    elapsed = time.perf_counter() - t0
    # the reference's own step: queue.finish() after every launch (sw/benchmark.cpp:331-337), so its spmv_time_ms is a LATENCY -- the same
    # K SpMVs with a host synchronisation after each one, beside the pipelined figure above (which is `value`)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run()
        eng.sync()
    elapsed_sync = time.perf_counter() - t0
    # Two HIP-event measurements, both reported; WHICH one prices the roofline is fixed by the plan, not by which came out smaller:
    #   * one kernel per step (no slice-combine pass): two events around the K back-to-back launches / K -- an event pair around every
    #     launch adds ~2 us to each, a fifth of a 14 us kernel (rocprofv3 agrees with the region figure);
    #   * column-sliced plans (SpMV kernel + combine kernel per step): the event pair around every SpMV launch.
    _, ev_kernel_ms = eng.time_runs(0, steps)
    region_ms, _ = eng.time_runs(0, steps, kernel=False)
    kernel_ms_pairs, step_ms_region = ev_kernel_ms / steps, region_ms / steps
    if stats["col_slices"] == 1:
        kernel_ms, kernel_ms_how = step_ms_region, "two HIP events around the K back-to-back launches / K (one kernel per step)"
    else:
        kernel_ms, kernel_ms_how = kernel_ms_pairs, "HIP event pair around every SpMV launch (the step has a second, slice-combine kernel)"
    ms = elapsed / steps * 1e3
    value = 8.0 * nnz / (elapsed / steps) / 1e9
    achieved = 8.0 * nnz / (kernel_ms * 1e-3) / 1e9
    # the same matrix loaded straight from CSR (hs_load_matrix_csr: pad + convert + re-tile on the device, no csr2cpsr): same image,
    # so the same y -- checked -- and the pre-processing cost of a caller that does not need the CPSR buffers for anything else
    with device.SpmvEngine(impl, device_id=device_id) as eng2:
        eng2.load_matrix_csr(csr)
        st2 = eng2.stats()
        eng2.load_vector(xw)
        eng2.run()
        y2 = eng2.read_result()
    same = np.array_equal(y2, y_gpu) if impl == host.IMPL_FIXED else np.allclose(y2.view(np.float32), y_gpu.view(np.float32), rtol=1e-5, atol=1e-5)
    if not same or st2["stream_bytes"] != stats["stream_bytes"]:
        print(json.dumps({"error": "the CSR load path gives a different image or result than the CPSR path", "config": name}))
        sys.exit(1)
    spmm = spmm_probe(np, host, eng, impl, packets, rng, xw) if device.STREAM_FORMATS[stats["stream_format"]] == "bitmap" and with_spmm else None
    traffic, traffic_from = read_traffic(name, stats["stream_bytes"], impl)
    res = {
        "workload": f"{name}, {IMPL_NAMES[impl]} IMPL, v={packets.vb_bank} o={packets.ob_bank}",
        "rows": true_rows, "cols": packets.num_cols, "nnz": int(nnz), "partitions": f"{packets.num_row_partitions}x{packets.num_col_partitions}",
        "stream_format": device.STREAM_FORMATS[stats["stream_format"]] + (" (light kernel)" if stats.get("light_kernel") else ""), "col_slices": stats["col_slices"],
        "ms_per_step": round(ms, 5), "value": round(value, 2), "unit": "GB/s", "gops": round(2.0 * nnz / (elapsed / steps) / 1e9, 2),
        "ms_per_step_synchronous": round(elapsed_sync / steps * 1e3, 5),
        "value_synchronous": round(8.0 * nnz / (elapsed_sync / steps) / 1e9, 2),
        "spin_up_steps": spun,
        "gibps_reference_formula": round(8.0 * nnz / 2 ** 30 / (elapsed / steps), 2),
        "hbm_roofline_fraction_whole_job": round(value / HBM_PEAK_GBS, 4),
        "roofline": {"bound": "hbm", "kernel": "spmv_bitmap_kernel" if stats["stream_format"] == 2 else "spmv_sweep_kernel" if stats["stream_format"] == 6 else "spmv_light_kernel" if stats.get("light_kernel") else "spmv_rowblock_kernel",
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms": round(kernel_ms, 5), "kernel_ms_from": kernel_ms_how, "kernel_ms_event_pairs": round(kernel_ms_pairs, 5),
                     "step_ms_two_events_around_K_launches": round(step_ms_region, 5),
                     "frac_event_pairs": round(8.0 * nnz / (kernel_ms_pairs * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(8 * nnz),
                     "streamed_bytes_per_launch": int(stats["stream_bytes"]), "traffic": traffic, "traffic_provenance": traffic_from,
                     "frac_rocprof": traffic_from.get("frac_rocprof"), "kernel_us_rocprof": traffic_from.get("kernel_us_rocprof")},
        "parity_vs_oracle": parity,
        "preprocess_s": {"format_csr2cpsr": round(t_fmt, 3), "device_load_retile": round(stats["load_seconds"], 3),
                         "device_load_from_csr_instead": round(st2["load_seconds"], 3)},
    }
    if float_error:
        res["float_error"] = float_error
    if spmm:
        res["spmm_extension"] = spmm
    return res, dict(eng=eng, packets=packets, csr=csr, x=x, xw=xw, impl=impl, nnz=nnz, y_cpu=y_cpu, t_cpu=t_cpu, reps=reps, cfg=cfg)


def cpu_baseline_for(np, host, ctx, rank):
    """The reported CPU baseline: the oracle (csim-equivalent restatement) on one core + two context numbers."""
    from oracle import oracle as orc
    packets, impl, nnz, xw, x, csr, y_cpu = ctx["packets"], ctx["impl"], ctx["nnz"], ctx["xw"], ctx["x"], ctx["csr"], ctx["y_cpu"]
    t_cpu, reps = ctx["t_cpu"], ctx["reps"]
    chans = [packets.channel_ptr(c)[0] for c in range(16)]
    base = {"value": round(8.0 * nnz / t_cpu / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{reps} full SpMV(s) of the same matrix through oracle/cpu_ref.c (csim-equivalent restatement, 1 thread), {t_cpu*1e3:.1f} ms each",
            "gops": round(2.0 * nnz / t_cpu / 1e9, 4), "host_cpus": orc.usable_cores()}
    try:   # context only: the same restatement with one host thread per cluster (the 16 clusters are independent)
        threads = min(16, orc.usable_cores())
        t0 = time.perf_counter()
        y_par = orc.spmv_per_channel_threads(impl, chans, xw, packets.num_rows, packets.num_cols, packets.num_row_partitions,
                                             packets.num_col_partitions, packets.ob_bank, packets.vb_bank, threads=threads)
        t_par = time.perf_counter() - t0
        if np.array_equal(y_par, y_cpu):
            base["cpsr_one_thread_per_cluster"] = {"value": round(8.0 * nnz / t_par / 1e9, 3), "unit": "GB/s", "cores": threads,
                                                   "gops": round(2.0 * nnz / t_par / 1e9, 3), "sample": f"1 SpMV, {t_par*1e3:.1f} ms"}
    except Exception as e:
        log(rank, f"per-cluster-thread baseline skipped: {e}")
    try:   # context only: plain float32 CSR loop (compute_ref, csim.cpp:143-158) with OpenMP over the host cores
        ip, ix, dv = csr.arrays()
        xf = np.ascontiguousarray(x, dtype=np.float32)
        yref = np.zeros(packets.num_rows, dtype=np.float32)
        best = None
        for threads in sorted({min(16, orc.usable_cores()), min(64, orc.usable_cores()), orc.usable_cores()}):   # cgroup quotas make "all" a bad guess
            orc.compute_ref_parallel(packets.num_rows, ip, ix, dv, xf, out=yref, threads=threads)
            t0 = time.perf_counter()
            for _ in range(5):
                orc.compute_ref_parallel(packets.num_rows, ip, ix, dv, xf, out=yref, threads=threads)
            t_omp = (time.perf_counter() - t0) / 5
            if best is None or t_omp < best[0]:
                best = (t_omp, threads)
        t_omp, threads = best
        base["csr_openmp_best_thread_count"] = {"value": round(8.0 * nnz / t_omp / 1e9, 3), "unit": "GB/s", "cores": threads,
                                                "gops": round(2.0 * nnz / t_omp / 1e9, 3),
                                                "sample": f"5 float32 CSR SpMVs, {t_omp*1e3:.2f} ms each (best of 16 / 64 / all host threads)"}
    except Exception as e:  # the context number must never break the bench line
        log(rank, f"csr_openmp baseline skipped: {e}")
    return base


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
        for _ in range(max(1, SPIN_UP_STEPS // len(order))):
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
    log(rank, f"{cfg.name}: round-robin over {count} images ({image_mb:.0f} MB): {t_rr*1e6:.1f} us per SpMV; one image: {t_one*1e6:.1f} us")
    return {"images": count, "image_megabytes_total": round(image_mb, 1), "ms_per_step_round_robin": round(t_rr * 1e3, 5),
            "ms_per_step_one_image_same_loop": round(t_one * 1e3, 5),
            "frac_whole_job_round_robin": round(8.0 * mean_nnz / t_rr / 1e9 / HBM_PEAK_GBS, 4),
            "frac_whole_job_one_image": round(8.0 * first["nnz"] / t_one / 1e9 / HBM_PEAK_GBS, 4)}


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
        res["hbm_roofline_fraction_quoted"] = res["hbm_roofline_fraction_whole_job"]
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
        with device.SpmvEngine(impl) as eng:
            eng.load_matrix_csr(csr)
            eng.load_vector(xw)
            st = eng.stats()
            for _ in range(SPIN_UP_STEPS // 2):
                eng.run()
            eng.sync()
            best = 1e9
            for _ in range(3):
                region_ms, _ = eng.time_runs(5, steps, kernel=False)
                best = min(best, region_ms / steps)
            return best * 1e3, st

    t_whole, st_whole = step_us(full)
    out = {"workload": f"{name}, {IMPL_NAMES[impl]} IMPL", "nnz": int(full.nnz), "unsplit_us": round(t_whole, 2),
           "unsplit_plan": f"{device.STREAM_FORMATS[st_whole['stream_format']]}, {st_whole['col_slices']} slices, {st_whole['num_blocks']} blocks", "splits": []}
    for n in ways:
        bounds = sharding.split_rows_by_nnz(indptr, n, granule)
        slabs = []
        for r in range(n):
            lo, hi = bounds[r], bounds[r + 1]
            if hi == lo:
                continue
            ip, ix, dv = sharding.slab_arrays(indptr, indices, data, lo, hi)
            t, st = step_us(host.CSRMatrix.from_arrays(hi - lo, full.num_cols, ip, ix, dv))
            slabs.append({"rank": r, "rows": int(hi - lo), "nnz": int(ip[-1]), "us": round(t, 2),
                          "plan": f"{device.STREAM_FORMATS[st['stream_format']]}, {st['col_slices']} slices, {st['num_blocks']} blocks"})
        worst = max(s["us"] for s in slabs)
        out["splits"].append({"n_gpus": n, "max_slab_us": worst, "mean_slab_us": round(sum(s["us"] for s in slabs) / len(slabs), 2),
                              "predicted_compute_only_efficiency": round(t_whole / (n * worst), 4),
                              "roofline_us_per_slab": round(8.0 * full.nnz / n / (HBM_PEAK_GBS * 1e9) * 1e6, 2), "slabs": slabs})
        log(rank, f"{name} split {n} ways: slowest slab {worst:.1f} us against {t_whole:.1f} us unsplit -> predicted compute-only efficiency {t_whole / (n * worst) * 100:.0f} %")
    return out


def fail(message, **extra):
    """A request this run cannot honour: ONE JSON line with `error` (so a driver that parses the last line sees it) and exit code 2."""
    print(json.dumps({"error": message, **extra}), flush=True)
    sys.exit(2)


def visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <the same arguments>` -- exactly what the driver would type -- and pass its
    output and exit code through.  With the RCCL backend every rank needs its own GPU: fewer visible GPUs than N is an error, not a
    reason to measure something smaller."""
    import socket
    import subprocess
    if args.backend == "nccl":
        have = visible_gpus()
        if have < args.gpus:
            fail(f"--gpus {args.gpus} asked for, {have} GPU(s) visible on this machine: nothing measured", n_gpus_requested=args.gpus, gpus_visible=have)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(0, "no launcher in the environment: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        fail(f"the {args.gpus}-rank run exited with code {rc}", n_gpus_requested=args.gpus)
    sys.exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default=None, help="N = 1: measure only this configuration ('bm': the reference's whole sweep, sw/bm.sh, in fixed point); N > 1: the matrix to shard (default mouse_gene)")
    ap.add_argument("--npz", default=None, help="real dataset file instead of the seeded stand-in")
    ap.add_argument("--impl", default=None, help="override the config's numeric mode")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong", help="N > 1: split ONE matrix (default) or one matrix-sized slab per rank")
    ap.add_argument("--gather", choices=["step", "final", "off"], default="final",
                    help="N > 1: what `value` times beside the K SpMVs: one all-gather of the y slabs at the end (default), one after every SpMV "
                         "(overlapped with the next), or none; the other patterns are reported alongside")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: nccl = RCCL over xGMI, one GPU per rank (the measurement).  gloo = the launcher / sharding self-test on host "
                         "memory: needs HISPARSE_HIP_LIB=<libhisparse_cpu.so> (the separate host-thread build of the C-ABI); never a measurement")
    ap.add_argument("--quick", action="store_true", help="N = 1: headline only (no per-config runs, no round-robin leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed path even for one GPU")
    ap.add_argument("--predict-scaling", action="store_true",
                    help="N = 1: time every row slab of the 2-, 4- and 8-way split of --config (default mouse_gene) on this GPU and print the predicted compute-only scaling")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus < 1:
        fail(f"--gpus {args.gpus}: need at least one")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)             # no launcher: start the N ranks ourselves (never falls through to the 1-GPU line)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:                   # a line labelled n_gpus = N must have been measured by N ranks
        if rank == 0:
            fail(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to measure one and label it the other")
        sys.exit(2)
    dist_mode = world > 1 or args.force_dist
    if rank == 0:
        ensure_built()
    if dist_mode:
        return main_distributed(args, rank, local_rank, world)

    import numpy as np
    from hisparse_amd import datasets, device, host

    from hisparse_amd import sharding

    def bm_entry(name, paper_gops, res, impl="fixed"):
        """one line of the reference's sweep (sw/bm.sh) next to the paper's U280 figure for the same matrix and numeric mode (Table 3: fixed
        point; Table 7: float_pob = "PB", float_stall = "RI")"""
        row = {"matrix": name, "impl": impl, "nnz": res["nnz"], "partitions": res["partitions"], "stream_format": res["stream_format"], "col_slices": res["col_slices"],
               "ms_per_step": res["ms_per_step"], "ms_per_step_synchronous": res["ms_per_step_synchronous"], "value": res["value"], "unit": "GB/s", "gops": res["gops"],
               "hbm_roofline_fraction_whole_job": res["hbm_roofline_fraction_whole_job"], "frac": res["roofline"]["frac"], "frac_kernel": res["roofline"]["frac"],
               "frac_rocprof": res["roofline"].get("frac_rocprof"), "frac_mall_cold": res["roofline"].get("frac_mall_cold"),
               "kernel_ms": res["roofline"]["kernel_ms"], "streamed_bytes_per_launch": res["roofline"]["streamed_bytes_per_launch"],
               "image_fits_infinity_cache": res["roofline"]["streamed_bytes_per_launch"] < 256 * 2 ** 20,
               "parity_vs_oracle": res["parity_vs_oracle"],
               "paper_gops_u280": paper_gops, "paper_table": "Table 3" if impl == "fixed" else "Table 7", "gops_vs_paper": round(res["gops"] / paper_gops, 1) if paper_gops else None}
        if impl == "fixed":
            row["paper_table3_gops_u280_fixed"] = paper_gops
        if "float_error" in res:
            row["float_error"] = res["float_error"]
        return row

    def float_sweep(already):
        """The matrices the paper quotes in all three numeric modes (Table 7) in float_pob (o = 1024: 8 x the row partitions) and float_stall
        (F = 8), each checked against the oracle at full size inside measure_single; `already`: (name, impl) -> res measured above."""
        rows = []
        for name, fx, pb, ri in datasets.BM_FLOAT:
            for impl_name, paper in (("float_pob", pb), ("float_stall", ri)):
                res = already.get((name, impl_name))
                if res is None:
                    res, ctx = measure_single(np, datasets, device, host, name, sub_steps, sub_warm, impl_override=impl_name, rank=rank, with_spmm=False)
                    ctx["eng"].close()
                    del ctx
                fixed = already.get((name, "fixed"))
                row = bm_entry(name, paper, res, impl_name)
                if fixed is not None:      # each within 5 points of the fixed-point figure, or a named cause
                    row["fixed_point_fraction_whole_job"] = fixed["hbm_roofline_fraction_whole_job"]
                    row["points_vs_fixed_point"] = round((res["hbm_roofline_fraction_whole_job"] - fixed["hbm_roofline_fraction_whole_job"]) * 100, 1)
                rows.append(row)
                log(rank, f"bm {name}/{impl_name}: {res['ms_per_step']*1e3:.1f} us per SpMV = {res['gops']:.0f} GOPS, {res['hbm_roofline_fraction_whole_job']*100:.1f} % of the HBM roofline "
                          f"(paper {paper} GOPS); {res['parity_vs_oracle']}")
        return rows

    sub_steps = max(20, min(args.steps, 200))
    sub_warm = min(args.warmup, 20)
    if args.predict_scaling:      # only the strong-scaling prediction (bench.py --predict-scaling [--config mouse_gene])
        print(json.dumps({"predict_scaling": predict_scaling(np, datasets, device, host, sharding, args.config or "mouse_gene", sub_steps, rank)}), flush=True)
        return
    if args.config == "bm":       # only the reference's sweep (sw/bm.sh runs it in the mode of its bitstream: --impl), the whole list as the last line
        sweep_impl = args.impl or "fixed"
        paper7 = {n: {"fixed": fx, "float_pob": pb, "float_stall": ri} for n, fx, pb, ri in datasets.BM_FLOAT}
        rows = []
        for name, paper in datasets.BM_LIST:
            res, ctx = measure_single(np, datasets, device, host, name, sub_steps, sub_warm, impl_override=sweep_impl, rank=rank, with_spmm=False)
            ctx["eng"].close()
            del ctx
            quoted = paper if sweep_impl == "fixed" else paper7.get(name, {}).get(sweep_impl)
            rows.append(bm_entry(name, quoted, res, sweep_impl))
            log(rank, f"bm {name}/{sweep_impl}: {res['ms_per_step']*1e3:.1f} us per SpMV = {res['gops']:.0f} GOPS ({res['hbm_roofline_fraction_whole_job']*100:.1f} % of the HBM roofline), paper {quoted}")
        print(json.dumps({"metric": f"SpMV GBPS / GOPS per matrix of sw/bm.sh, {sweep_impl} IMPL, 1 x MI355X", "bm_list": rows}), flush=True)
        return

    headline = args.config or "ogbl_ppa"
    per_config, bm_rows, scaling, scaling_more, float_rows = [], {}, None, [], []
    if not args.config and not args.quick:
        # the three other single-GPU configurations of BASELINE.json + the second ogbl-ppa stand-in (symmetric R-MAT),
        # each also round-robin over enough images to be Infinity-Cache-cold
        measured = {}      # (matrix, numeric mode) -> result, for the float sweep below
        for name in ("transformer_50", "ogbn_products", "mouse_gene", "ogbl_ppa_rmat"):
            res, ctx = measure_single(np, datasets, device, host, name, sub_steps, sub_warm, cpu_seconds=0.0, rank=rank)
            cold = mall_cold(np, datasets, device, host, ctx, sub_steps, sub_warm, rank)
            res["roofline"]["frac_mall_cold"] = cold["frac_whole_job_round_robin"]
            res["roofline"]["mall_cold"] = cold
            quote_hbm_fraction(res)
            ctx["eng"].close()
            del ctx
            per_config.append(res)
            if name == "mouse_gene":
                bm_rows[name] = res
            measured[(name, res["workload"].split(", ")[1].split(" ")[0])] = res
            log(rank, f"{name}: {res['ms_per_step']*1e3:.1f} us per SpMV, kernel {res['roofline']['kernel_ms']*1e3:.1f} us = {res['roofline']['frac']*100:.1f} % of the HBM roofline")
        # the rest of the reference's sweep (sw/bm.sh:3-17), in the numeric mode of the paper's Table 3
        for name, _ in datasets.BM_LIST:
            if name in ("ogbl_ppa", "mouse_gene"):
                continue
            res, ctx = measure_single(np, datasets, device, host, name, sub_steps, sub_warm, impl_override="fixed", rank=rank, with_spmm=False)
            ctx["eng"].close()
            del ctx
            bm_rows[name] = res
            measured[(name, "fixed")] = res
            log(rank, f"bm {name}/fixed: {res['ms_per_step']*1e3:.1f} us per SpMV = {res['gops']:.0f} GOPS, {res['hbm_roofline_fraction_whole_job']*100:.1f} % of the HBM roofline")
        # ... and the float modes of the sweep on the matrices the paper quotes in all three (Table 7; sw/bm.sh:19-35)
        float_rows = float_sweep(measured)
        scaling = predict_scaling(np, datasets, device, host, sharding, "mouse_gene", sub_steps, rank)
        # the larger graphs, where row slabs are still 15-25 us of streaming (8-way split only: 1 + 8 loads each)
        scaling_more = [predict_scaling(np, datasets, device, host, sharding, name, sub_steps, rank, ways=(8,)) for name in ("hollywood", "ogbn_products")]
    res, ctx = measure_single(np, datasets, device, host, headline, args.steps, args.warmup, npz=args.npz, impl_override=args.impl,
                              cpu_seconds=0.0 if args.no_cpu_baseline else args.cpu_seconds, rank=rank)
    cpu_baseline = None if args.no_cpu_baseline else cpu_baseline_for(np, host, ctx, rank)
    if not args.quick and not args.npz:
        cold = mall_cold(np, datasets, device, host, ctx, args.steps, args.warmup, rank)
        res["roofline"]["frac_mall_cold"] = cold["frac_whole_job_round_robin"]
        res["roofline"]["mall_cold"] = cold
        quote_hbm_fraction(res)
    ctx["eng"].close()
    impl = ctx["impl"]
    bm_rows[headline] = res
    out = {
        "metric": "SpMV GBPS (8 B per non-zero per SpMV, sw/benchmark.cpp:312-346; GOPS and % of the HBM roofline alongside)",
        "value": res["value"], "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "spin_up_steps": res.get("spin_up_steps", SPIN_UP_STEPS),
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32 Q8.24 fixed point (u64 accumulate)" if impl == host.IMPL_FIXED else "f32",
        "data": "synthetic" if not args.npz else "file",
        "config": {"workload": res["workload"], "rows": res["rows"], "cols": res["cols"], "nnz_per_gpu": res["nnz"], "nnz_total": res["nnz"],
                   "partitions": res["partitions"], "stream_format": res["stream_format"], "parallelism": "row-slab x1"},
        "gops": res["gops"], "gibps_reference_formula": res["gibps_reference_formula"],
        "hbm_roofline_fraction_whole_job": res["hbm_roofline_fraction_whole_job"],
        "hbm_roofline_fraction_quoted": res.get("hbm_roofline_fraction_quoted", res["hbm_roofline_fraction_whole_job"]),
        "hbm_roofline_fraction_quoted_from": res.get("hbm_roofline_fraction_quoted_from", "whole job, one image"),
        "ms_per_step_synchronous": res["ms_per_step_synchronous"], "value_synchronous": res["value_synchronous"],
        "roofline": res["roofline"], "cpu_baseline": cpu_baseline, "parity_vs_oracle": res["parity_vs_oracle"], "parity_pins": PARITY_PINS,
        "preprocess_s": res["preprocess_s"],
    }
    if "float_error" in res:
        out["float_error"] = res["float_error"]
    if "spmm_extension" in res:
        out["spmm_extension"] = res["spmm_extension"]
    if per_config:
        out["per_config"] = per_config
    if len(bm_rows) == len(datasets.BM_LIST):
        out["bm_list"] = [bm_entry(name, paper, bm_rows[name]) for name, paper in datasets.BM_LIST]
    if float_rows:
        out["bm_list_float"] = float_rows
    if scaling:
        out["strong_scaling_prediction"] = scaling
    if scaling_more:
        out["strong_scaling_prediction_more"] = scaling_more
    print(json.dumps(out), flush=True)


def main_distributed(args, rank, local_rank, world):
    import numpy as np
    import torch
    import torch.distributed as dist
    from hisparse_amd import datasets, device, host, sharding

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    on_gpu = args.backend == "nccl"
    if on_gpu:
        have = visible_gpus()
        if local_rank >= have:
            if rank == 0:
                fail(f"{world} ranks over RCCL need {world} GPUs, {have} visible", n_gpus_requested=world, gpus_visible=have)
            sys.exit(2)
        torch.cuda.set_device(local_rank)
    elif os.path.basename(device._LIB_PATH) == "libhisparse_hip.so":
        if rank == 0:
            fail("--backend gloo is the launcher / sharding self-test on host memory: point HISPARSE_HIP_LIB at libhisparse_cpu.so "
                 "(the HIP library has no host path and writes y to device memory)")
        sys.exit(2)
    dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    dist.barrier()
    n_gpus = world
    name = args.config or "mouse_gene"
    dev = f"cuda:{local_rank}" if on_gpu else "cpu"
    spin_up_steps = SPIN_UP_STEPS if on_gpu else 0
    cuda_sync = torch.cuda.synchronize if on_gpu else (lambda: None)

    # ---- workload: this rank's row slab ---------------------------------------------------------------------------------
    t0 = time.perf_counter()
    cfg = datasets.CONFIGS[name]
    impl = host.impl_id(args.impl or cfg.impl)
    granule = 128 * (8 if impl == host.IMPL_FLOAT_STALL else 1)
    full_rows = None
    if args.scaling == "strong":
        _, full = datasets.load(name, path=args.npz)
        indptr, indices, data = full.arrays()
        full_rows = full.num_rows
        bounds = sharding.split_rows_by_nnz(indptr, n_gpus, granule)
        lo, hi = bounds[rank], bounds[rank + 1]
        if hi == lo:
            print(json.dumps({"error": f"the matrix has fewer than {n_gpus} x {granule} rows: rank {rank} has no slab"}))
            sys.exit(1)
        ip, ix, dv = sharding.slab_arrays(indptr, indices, data, lo, hi)
        csr = host.CSRMatrix.from_arrays(hi - lo, full.num_cols, ip, ix, dv)
        whole = full if rank == 0 else None        # rank 0 also times the unsplit matrix on its one GPU (the curve's N = 1 point)
        del full, indptr, indices, data
    else:   # weak: rank r owns slab r of a matrix that is n_gpus slabs tall; same generator, different seed per slab
        c = cfg
        csr = host.CSRMatrix.generate(c.kind, c.rows, c.cols, a=c.a, b=c.b, c=c.c, seed=c.seed + 1000 * rank) if not args.npz \
            else host.load_csr_matrix_from_float_npz(args.npz)
    true_rows = csr.num_rows
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    packets = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
    t_fmt = time.perf_counter() - t0
    nnz = packets.nnz
    rng = np.random.default_rng(2024)
    x = rng.uniform(0.0, 2.0, packets.num_cols).astype(np.float32) if impl == host.IMPL_FIXED else rng.normal(size=packets.num_cols).astype(np.float32)
    xw = host.pack_vector(impl, x)
    if args.scaling != "strong":
        whole = None
    eng = device.SpmvEngine(impl, device_id=local_rank if on_gpu else 0)
    eng.load_matrix(packets)
    eng.load_vector(xw)
    stats = eng.stats()
    log(rank, f"{name} ({args.scaling}): slab {packets.num_rows}x{packets.num_cols}, nnz {nnz}, generate {t_gen:.2f}s format {t_fmt:.2f}s "
              f"device-load {stats['load_seconds']:.2f}s, stream {stats['stream_bytes']/1e6:.0f} MB")

    # ---- y slab inside an all-gather buffer; one explicit stream for the kernels and the point RCCL synchronises against -------
    rows_all = [None] * world
    dist.all_gather_object(rows_all, packets.num_rows)
    chunk = max(rows_all)
    y_chunks = [torch.zeros(chunk, dtype=torch.int32, device=dev) for _ in range(2)]
    gathered = [torch.zeros(chunk * world, dtype=torch.int32, device=dev) for _ in range(2)]
    if on_gpu:
        main_stream = torch.cuda.Stream(device=dev)     # the legacy default stream has handle 0 = "the library's private stream"
        torch.cuda.set_stream(main_stream)
        torch.cuda.synchronize()
        eng.set_stream(main_stream.cuda_stream)
    pending = [None, None]
    step_no = [0]

    def run_into(y_tensor):
        """one slab SpMV whose result lands in y_tensor: the kernels write straight into it (hs_bind_device_result); the host-memory
        self-test (--backend gloo, libhisparse_cpu.so has no binding hooks) copies the library's own y instead"""
        if on_gpu:
            eng.bind_device_result(y_tensor.data_ptr())
            eng.run()
        else:
            eng.run()
            y_tensor[:packets.num_rows] = torch.from_numpy(eng.read_result().view(np.int32))

    def step(gather):
        if gather != "step":
            run_into(y_chunks[0])
            return
        cur = step_no[0] & 1
        step_no[0] += 1
        if pending[cur] is not None:
            pending[cur].wait()              # the gather that read this slab two steps ago (stream-level wait, no host sync)
            pending[cur] = None
        run_into(y_chunks[cur])
        pending[cur] = dist.all_gather_into_tensor(gathered[cur], y_chunks[cur], async_op=True)

    def sync():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None
        cuda_sync()
        eng.sync()

    def timed(gather, steps):
        for _ in range(spin_up_steps):
            step(gather)
        sync()
        for _ in range(args.warmup):
            step(gather)
        sync()
        dist.barrier()
        cuda_sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(gather)
        if gather == "final":
            dist.all_gather_into_tensor(gathered[0], y_chunks[0])
        sync()
        cuda_sync()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- correctness of what is about to be timed: this rank's slab against the oracle, and the gathered buffer ------------------
    run_into(y_chunks[0])
    dist.all_gather_into_tensor(gathered[0], y_chunks[0])
    sync()
    y_gpu = y_chunks[0][:packets.num_rows].cpu().numpy().view(np.uint32)
    mine = gathered[0][rank * chunk: rank * chunk + packets.num_rows].cpu().numpy().view(np.uint32)
    if not np.array_equal(mine, y_gpu):
        print(json.dumps({"error": "all-gathered y differs from the local slab", "rank": rank}))
        sys.exit(1)
    parity, _, t_cpu, _, _ = oracle_check(np, host, impl, packets, xw, y_gpu, 0.0)
    flag = torch.tensor([1.0 if parity == "MISMATCH" else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if flag.item() > 0:
        if rank == 0:
            print(json.dumps({"error": "a rank's slab does not match the oracle", "config": name}))
        sys.exit(1)

    # ---- timing: the exchange pattern asked for = `value`; the same SpMVs without any exchange alongside -------------------------
    elapsed = timed(args.gather, args.steps)
    compute_elapsed = timed("off", args.steps) if args.gather != "off" else elapsed
    step_elapsed = elapsed if args.gather == "step" else timed("step", args.steps)      # an all-gather after every SpMV (iterative callers)
    # ---- the same K steps with the gather done by PEER STORES instead of a collective (hs_push_result; hisparse_amd/peer_gather.py): every
    #      rank's kernels write y into its slot of its own gather buffer and one small kernel pushes the slab into every peer's buffer over
    #      xGMI.  Reported beside the RCCL figure; a failure here (IPC not available) is reported, not fatal.
    push = None
    if on_gpu:
        try:
            from hisparse_amd import peer_gather
            pg = peer_gather.PeerGather(dist, rank, world, chunk, device_id=local_rank)
            push_no = [0]

            def push_step(_gather):
                b = push_no[0] & 1
                push_no[0] += 1
                eng.bind_device_result(pg.my_slot(b))
                eng.run()
                eng.push_result(pg.targets(b), packets.num_rows)

            saved_step = step
            step = push_step
            try:
                push_elapsed = timed("push", args.steps)
            finally:
                step = saved_step
            sync()
            dist.barrier()
            mine_rows = [None] * world
            dist.all_gather_object(mine_rows, packets.num_rows)
            ok = True
            for b in (0, 1):      # both buffers against what RCCL gathered before the timing (same x, same matrix: the same y)
                got = pg.read(b)
                ref = gathered[0].cpu().numpy().view(np.uint32).reshape(world, chunk)
                for r in range(world):
                    ok = ok and bool(np.array_equal(got[r, :mine_rows[r]], ref[r, :mine_rows[r]]))
            flag2 = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=dev)
            dist.all_reduce(flag2, op=dist.ReduceOp.MAX)
            dist.barrier()
            eng.bind_device_result(y_chunks[0].data_ptr())
            pg.close()
            push = {"ms_per_step": round(push_elapsed / args.steps * 1e3, 5), "ms_per_step_added": round((push_elapsed - compute_elapsed) / args.steps * 1e3, 5),
                    "gathered_equals_rccl_gather_on_every_rank": flag2.item() == 0.0,
                    "note": "hs_push_result: plain 16-byte stores into the peers' gather buffers (hipIpcOpenMemHandle), stream-ordered behind the SpMV; "
                            "the ranks synchronise once around the K steps"}
        except Exception as e:      # noqa: BLE001 -- the push path is an extra measurement
            push = {"error": f"{type(e).__name__}: {e}"}
            log(rank, f"peer-store gather skipped: {e}")
    tot = torch.tensor([float(nnz)], dtype=torch.float64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    total_nnz = float(tot.item())
    if on_gpu:
        eng.set_stream(None)
    _, ev_kernel_ms = eng.time_runs(0, args.steps)
    kernel_ms = ev_kernel_ms / args.steps
    dist.barrier()
    # strong scaling: the SAME matrix, unsplit, on rank 0's GPU alone -- the N = 1 point the N-GPU numbers of this workload belong to
    # (bench.py --gpus 1 without a launcher measures the ogbl-ppa headline instead)
    one_gpu = None
    if whole is not None:
        with device.SpmvEngine(impl, device_id=local_rank if on_gpu else 0) as eng1:
            eng1.load_matrix_csr(whole)
            eng1.load_vector(xw)
            for _ in range(spin_up_steps + args.warmup):
                eng1.run()
            eng1.sync()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                eng1.run()
            eng1.sync()
            one = (time.perf_counter() - t1) / args.steps
            one_gpu = {"n_gpus": 1, "ms_per_step": round(one * 1e3, 5), "value": round(8.0 * whole.nnz / one / 1e9, 2), "unit": "GB/s",
                       "note": "the same matrix unsplit on rank 0's GPU, y in HBM (no exchange); measured while the other ranks wait"}
        del whole
    dist.barrier()

    out = None
    if rank == 0:
        per_step = elapsed / args.steps
        value = 8.0 * total_nnz / per_step / 1e9
        achieved = 8.0 * nnz / (kernel_ms * 1e-3) / 1e9
        gather_text = {"step": " + all_gather(y) over RCCL every step (overlapped with the next SpMV)", "final": " + one final all_gather(y) over RCCL", "off": ""}[args.gather]
        out = {
            "metric": "SpMV GBPS (8 B per non-zero per SpMV, sw/benchmark.cpp:312-346; GOPS and % of the HBM roofline alongside)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "spin_up_steps": spin_up_steps,
            "ms_per_step": round(per_step * 1e3, 5), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "backend": "nccl (RCCL)" if on_gpu else f"gloo on host memory with {os.path.basename(device._LIB_PATH)}: launcher self-test, NOT a measurement",
            "dtype": "u32 Q8.24 fixed point (u64 accumulate)" if impl == host.IMPL_FIXED else "f32",
            "data": "synthetic" if not args.npz else "file",
            "config": {"workload": f"{name}, {IMPL_NAMES[impl]} IMPL, v={packets.vb_bank} o={packets.ob_bank}", "rows": full_rows or true_rows * n_gpus,
                       "cols": packets.num_cols, "nnz_per_gpu": int(nnz), "nnz_total": int(total_nnz), "slab_rows_rank0": true_rows,
                       "parallelism": f"row-slab x{n_gpus}, balanced by non-zeros" + gather_text},
            "gops": round(2.0 * total_nnz / per_step / 1e9, 2),
            "gibps_reference_formula": round(8.0 * total_nnz / 2 ** 30 / per_step, 2),
            "hbm_roofline_fraction_whole_job": round(value / (HBM_PEAK_GBS * n_gpus), 4),
            "compute_only": {"ms_per_step": round(compute_elapsed / args.steps * 1e3, 5), "value": round(8.0 * total_nnz / (compute_elapsed / args.steps) / 1e9, 2),
                             "unit": "GB/s", "hbm_roofline_fraction": round(8.0 * total_nnz / (compute_elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * n_gpus), 4),
                             "note": "the same K slab SpMVs with y left sharded in HBM, like the reference leaves it (sw/benchmark.cpp:318-338)"},
            "same_workload_on_one_gpu": one_gpu,
            "exchange": {"pattern": args.gather, "bytes_per_rank_per_gather": int(chunk) * 4,
                         "ms_per_step_added": round((elapsed - compute_elapsed) / args.steps * 1e3, 5)},
            "exchange_every_step": {"ms_per_step": round(step_elapsed / args.steps * 1e3, 5),
                                    "ms_per_step_added": round((step_elapsed - compute_elapsed) / args.steps * 1e3, 5),
                                    "value": round(8.0 * total_nnz / (step_elapsed / args.steps) / 1e9, 2), "unit": "GB/s",
                                    "note": "all_gather_into_tensor(y) over RCCL after EVERY SpMV, the gather of step k overlapping the SpMV of step k + 1"},
            "exchange_push": push,
            "roofline": {"bound": "hbm", "kernel": "spmv_bitmap_kernel" if stats["stream_format"] == 2 else "spmv_sweep_kernel" if stats["stream_format"] == 6 else "spmv_rowblock_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_bytes_per_launch": int(8 * nnz), "streamed_bytes_per_launch": int(stats["stream_bytes"]), "traffic": None,
                         "note": "rank 0's slab"},
            "cpu_baseline": None if args.no_cpu_baseline else {"value": round(8.0 * nnz / t_cpu / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                                                               "sample": f"rank 0's slab, 1 SpMV through oracle/cpu_ref.c (1 thread), {t_cpu*1e3:.1f} ms"},
            "parity_vs_oracle": parity + " (every rank's slab)",
            "preprocess_s": {"format_csr2cpsr": round(t_fmt, 3), "device_load_retile": round(stats["load_seconds"], 3)},
        }
    eng.close()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner to the C stdout buffer; flush it first so that the JSON line is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
