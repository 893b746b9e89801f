#!/usr/bin/env python3
"""bench.py -- SpMV throughput of the MI355X hot path on the reference's benchmark matrices.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME] [--npz FILE] [--quick]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
          or simply `python bench.py --gpus N`: without a launcher's WORLD_SIZE in the environment the script starts its N ranks itself
          and fails loudly (a JSON error line, exit code 2) when the machine shows fewer than N GPUs.

A "step" is one full SpMV (every row partition) y = A x through the drop-in C-ABI, with the matrix (re-tiled at load time), x and y
already resident in HBM; the K timed steps are enqueued by one hs_run_batch(K) call (the reference's NUM_RUNS loop as a unit) and
followed by one synchronisation.  Metric (BASELINE.json): the reference's "data throughput" of sw/benchmark.cpp:312-346 -- 8 bytes per non-zero
per SpMV -- in decimal GB/s, with GOPS (2 flops per non-zero) and the fraction of the 8 TB/s HBM roofline.

OUTPUT.  The LAST line of stdout is one JSON object of < 4 KB (the driver keeps an 8 KB tail of stdout and of stderr): the headline
configuration -- BASELINE.json configs[1], ogbl-ppa (seeded stand-in, hisparse_amd/datasets.py), fixed point, default banks -- with
`roofline` and `cpu_baseline`.  The three fractions are named for what they divide:
    roofline.frac               8 nnz / (average duration of the SpMV kernel) / 8 TB/s -- hs_time_kernel: ONE HIP event pair around K
                                back-to-back launches of the kernel alone; the figure rocprofv3 --stats gives (profiles/)
    roofline.frac_whole_step    8 nnz / (wall time per step: kernel + slice-combine pass + launch gaps) / 8 TB/s   (= value / 8000)
    roofline.frac_event_pairs   the kernel inside whole steps, a HIP event pair around EVERY launch (each pair adds ~3 us)
`roofline.traffic` = HBM bytes per launch from two short rocprofv3 counter passes run BY this script (bench_extras.live_traffic; the
committed copy of profiles/hbm_traffic.json beside it and as the fallback).
Everything else a default run measures (the other BASELINE configurations, the reference's whole sweep sw/bm.sh in all numeric modes,
MALL-cold round-robin legs, the strong-scaling prediction: bench_extras.py) goes to `bench_details.json` next to this file and, one row
per matrix, to stderr.  `--config NAME` measures only that configuration; `--quick` skips the extras.

N > 1 (BASELINE.json configs[4]): mouse_gene, ONE matrix split into N row slabs by non-zero count (`--scaling strong`;
hisparse_amd/sharding.py), every rank formats and loads its slab and holds all of x.  Timed = `value`: K slab SpMVs and ONE final
all-gather of the y slabs over RCCL (`gather: final`, north_star's "a final RCCL gather over xGMI"; the reference too runs its NUM_RUNS
launches and collects y once, sw/benchmark.cpp:318-346).  In the same line: the same K SpMVs with y left sharded (`compute_only`),
with an all-gather after EVERY SpMV (`exchange_every_step`) and with that gather done by peer stores (`exchange_push`).
`--backend gloo --share-gpu` is the DRY RUN of that path on one GPU: N processes on GPU 0, the HIP engine, host-staged gloo collectives.
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
LINE_LIMIT = 4000       # bytes of the final JSON line (the driver's tail holds 8000)
DETAILS_FILE = os.path.join(ROOT, "bench_details.json")


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




def kernel_name(stats):
    return ("spmv_bitmap_kernel" if stats["stream_format"] == 2 else "spmv_sweep_kernel" if stats["stream_format"] == 6
            else "spmv_light_kernel" if stats.get("light_kernel") else "spmv_rowblock_kernel")


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
    fmt = device.STREAM_FORMATS[stats["stream_format"]] + (" (light kernel)" if stats.get("light_kernel") else "")
    eng.run()
    y_gpu = eng.read_result()
    exact = None
    if impl != host.IMPL_FIXED:      # float modes: the exact (float64) product, to place the GPU's and csim's rounding errors side by side
        import scipy.sparse as sp
        ip, ix, dv = csr.arrays()
        exact = sp.csr_matrix((dv.astype(np.float64), ix.astype(np.int64), ip.astype(np.int64)), shape=(csr.num_rows, csr.num_cols)) @ x[:csr.num_cols].astype(np.float64)
        del ip, ix, dv
    parity, y_cpu, t_cpu, reps, float_error = oracle_check(np, host, impl, packets, xw, y_gpu, cpu_seconds, exact)
    log(rank, f"{name}/{IMPL_NAMES[impl]}: {packets.num_rows}x{packets.num_cols} nnz {nnz} parts {packets.num_row_partitions}x{packets.num_col_partitions}; "
              f"format {t_fmt:.2f}s load {stats['load_seconds']:.2f}s; CPSR {stats['cpsr_bytes']/1e6:.0f} MB -> {fmt} {stats['stream_bytes']/1e6:.0f} MB; oracle {t_cpu*1e3:.0f} ms: {short_parity(parity)}")
    if parity == "MISMATCH":
        print(json.dumps({"error": "GPU result does not match the oracle", "config": name}))
        sys.exit(1)
    spun = spin_up(eng)
    for _ in range(warmup):
        eng.run()
    eng.sync()
    # the timed region: K steps = the reference's NUM_RUNS loop, enqueued by ONE hs_run_batch call (the library's C loop: a Python loop
    # over hs_run adds a ctypes call per step, which a 6 us step of a small matrix feels), one synchronisation at the end
    t0 = time.perf_counter()
    eng.run_batch(steps)
    eng.sync()
    elapsed = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(steps):                                   # the same K steps as K hs_run calls from Python, beside it
        eng.run()
    eng.sync()
    elapsed_python = time.perf_counter() - t0
    # the reference's own step: queue.finish() after every launch (sw/benchmark.cpp:331-337), so its spmv_time_ms is a LATENCY -- the same
    # K SpMVs with a host synchronisation after each one, beside the pipelined figure above (which is `value`)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run()
        eng.sync()
    elapsed_sync = time.perf_counter() - t0
    # the same K steps replayed from ONE captured hipGraph (hs_run_batch, batch_graph = 1): the step with the host's enqueue rate out of
    # the picture -- reported beside the plain loop, never `value`
    ms_graph = None
    try:
        eng.set_option("batch_graph", "1")
        eng.run_batch(steps)
        eng.sync()
        t0 = time.perf_counter()
        eng.run_batch(steps)
        eng.sync()
        ms_graph = (time.perf_counter() - t0) / steps * 1e3
        eng.set_option("batch_graph", None)
    except Exception as e:      # noqa: BLE001
        log(rank, f"{name}: graph replay skipped: {e}")
    # HIP events, on the stream the kernels are launched on.  kernel_ms (prices roofline.frac): hs_time_kernel -- ONE event pair around
    # K back-to-back launches of the SpMV kernel alone, / K = the average launch duration rocprofv3 --stats reports for it (plus the
    # sub-microsecond dispatch gap).  Beside it: the kernel inside whole steps with an event pair around every launch (each pair adds
    # ~3 us), and whole steps between two events.
    kernel_ms = eng.time_kernel(min(warmup, 20), steps) / steps
    _, ev_kernel_ms = eng.time_runs(0, steps)
    region_ms, _ = eng.time_runs(0, steps, kernel=False)
    kernel_ms_pairs, step_ms_region = ev_kernel_ms / steps, region_ms / steps
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
    spmm = None
    if with_spmm and device.STREAM_FORMATS[stats["stream_format"]] == "bitmap":
        import bench_extras
        spmm = bench_extras.spmm_probe(np, host, eng, impl, packets, rng, xw)
    traffic, traffic_from = read_traffic(name, stats["stream_bytes"], impl)
    res = {
        "matrix": name, "impl": IMPL_NAMES[impl],
        "workload": f"{name}, {IMPL_NAMES[impl]} IMPL, v={packets.vb_bank} o={packets.ob_bank}",
        "rows": true_rows, "cols": packets.num_cols, "nnz": int(nnz), "partitions": f"{packets.num_row_partitions}x{packets.num_col_partitions}",
        "stream_format": fmt, "col_slices": stats["col_slices"],
        "ms_per_step": round(ms, 5), "value": round(value, 2), "unit": "GB/s", "gops": round(2.0 * nnz / (elapsed / steps) / 1e9, 2),
        "ms_per_step_synchronous": round(elapsed_sync / steps * 1e3, 5),
        "ms_per_step_python_loop": round(elapsed_python / steps * 1e3, 5),
        "ms_per_step_graph_replay": round(ms_graph, 5) if ms_graph else None,
        "spin_up_steps": spun,
        "gibps_reference_formula": round(8.0 * nnz / 2 ** 30 / (elapsed / steps), 2),
        "frac_whole_step": round(value / HBM_PEAK_GBS, 4),
        "roofline": {"bound": "hbm", "kernel": kernel_name(stats),
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms": round(kernel_ms, 5),
                     "kernel_ms_from": "hs_time_kernel: one HIP event pair around K back-to-back launches of the kernel alone, / K",
                     "frac_whole_step": round(value / HBM_PEAK_GBS, 4),
                     "frac_event_pairs": round(8.0 * nnz / (kernel_ms_pairs * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "kernel_ms_event_pairs": round(kernel_ms_pairs, 5), "step_ms_two_events_around_K_steps": round(step_ms_region, 5),
                     "algorithmic_bytes_per_launch": int(8 * nnz), "streamed_bytes_per_launch": int(stats["stream_bytes"]),
                     "traffic": traffic, "traffic_from": traffic_from},
        "parity_vs_oracle": parity,
        "preprocess_s": {"format_csr2cpsr": round(t_fmt, 3), "device_load_retile": round(stats["load_seconds"], 3),
                         "device_load_from_csr_instead": round(st2["load_seconds"], 3)},
    }
    if float_error:
        res["float_error"] = float_error
    if spmm:
        res["spmm_extension"] = spmm
    log(rank, f"{name}/{IMPL_NAMES[impl]}: step {ms*1e3:.1f} us = {value:.0f} GB/s = {value/HBM_PEAK_GBS*100:.1f} % whole step; kernel {kernel_ms*1e3:.1f} us = "
              f"{achieved/HBM_PEAK_GBS*100:.1f} %; event pairs {kernel_ms_pairs*1e3:.1f} us" + (f"; graph replay {ms_graph*1e3:.1f} us per step" if ms_graph else ""))
    return res, dict(eng=eng, packets=packets, csr=csr, x=x, xw=xw, impl=impl, nnz=nnz, y_cpu=y_cpu, t_cpu=t_cpu, reps=reps, cfg=cfg)


def short_parity(parity):
    """the parity verdict in a few words (stderr rows): bit-exact | tol-ok | tol-ok, csim abs 1e-4 over on N rows | MISMATCH"""
    if parity.startswith("bit-exact") or parity == "MISMATCH":
        return parity
    if "not met on" in parity:
        return "tol-ok (csim abs 1e-4: over on " + parity.split("not met on ")[1].split(";")[0] + ")"
    return "tol-ok"


def cpu_baseline_for(np, host, ctx, rank):
    """(compact baseline for the line, the longer record for the details file): the oracle (csim-equivalent restatement) on one core, and
    as context the same with one thread per cluster and a plain float32 CSR loop on the host cores."""
    from oracle import oracle as orc
    packets, impl, nnz, xw, x, csr, y_cpu = ctx["packets"], ctx["impl"], ctx["nnz"], ctx["xw"], ctx["x"], ctx["csr"], ctx["y_cpu"]
    t_cpu, reps = ctx["t_cpu"], ctx["reps"]
    chans = [packets.channel_ptr(c)[0] for c in range(16)]
    base = {"value": round(8.0 * nnz / t_cpu / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{reps} full SpMV(s) of the same matrix through oracle/cpu_ref.c (csim-equivalent restatement, 1 thread), {t_cpu*1e3:.1f} ms each",
            "gops": round(2.0 * nnz / t_cpu / 1e9, 4), "host_cpus": orc.usable_cores()}
    more = {}
    try:   # context only: the same restatement with one host thread per cluster (the 16 clusters are independent)
        threads = min(16, orc.usable_cores())
        t0 = time.perf_counter()
        y_par = orc.spmv_per_channel_threads(impl, chans, xw, packets.num_rows, packets.num_cols, packets.num_row_partitions,
                                             packets.num_col_partitions, packets.ob_bank, packets.vb_bank, threads=threads)
        t_par = time.perf_counter() - t0
        if np.array_equal(y_par, y_cpu):
            more["cpsr_one_thread_per_cluster"] = {"value": round(8.0 * nnz / t_par / 1e9, 3), "unit": "GB/s", "cores": threads,
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
        more["csr_openmp_best_thread_count"] = {"value": round(8.0 * nnz / t_omp / 1e9, 3), "unit": "GB/s", "cores": threads,
                                                "gops": round(2.0 * nnz / t_omp / 1e9, 3),
                                                "sample": f"5 float32 CSR SpMVs, {t_omp*1e3:.2f} ms each (best of 16 / 64 / all host threads)"}
        base["csr_openmp"] = {"value": more["csr_openmp_best_thread_count"]["value"], "cores": threads}
    except Exception as e:  # the context number must never break the bench line
        log(rank, f"csr_openmp baseline skipped: {e}")
    return base, dict(base, **more)


def summary_row(res):
    """one stderr row per measured (matrix, numeric mode): what the driver's 8 KB stderr tail should still hold"""
    r = res["roofline"]
    cold = r.get("frac_mall_cold")
    return (f"{res['matrix']}/{res['impl']}".ljust(28) + f"{res['stream_format'][:18]:<19}{res['col_slices']:>2} {res['ms_per_step']*1e3:8.1f} {res['value']:7.0f} "
            f"{res['gops']:6.0f} {res['frac_whole_step']*100:6.1f} {r['frac']*100:6.1f} " + (f"{cold*100:6.1f}" if cold is not None else "     -") + "  " + short_parity(res["parity_vs_oracle"]))


SUMMARY_HEAD = "matrix/impl".ljust(28) + "format".ljust(19) + "sl  us/step    GB/s   GOPS  %step %kernl  %cold  parity"


def emit(out, details, rows, scaling=None):
    """details -> bench_details.json; one row per matrix + the scaling prediction -> stderr; the compact line -> stdout, LAST."""
    try:
        with open(DETAILS_FILE, "w") as f:
            json.dump(details, f, indent=1)
        out["details"] = os.path.basename(DETAILS_FILE)
        gp = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(gp):      # a gpurun call merges only this directory back
            with open(os.path.join(gp, "bench_details.json"), "w") as f:
                json.dump(details, f, indent=1)
    except OSError as e:
        log(0, f"details file not written: {e}")
    if rows:
        log(0, "summary (% of the 8 TB/s HBM roofline: whole step / kernel alone / whole step MALL-cold)")
        log(0, SUMMARY_HEAD)
        for r in rows:
            log(0, r)
    for s in scaling or []:
        log(0, f"strong-scaling prediction {s['workload']}: unsplit {s['unsplit_us']} us; " + "; ".join(
            f"{sp['n_gpus']} GPUs: slowest slab {sp['max_slab_us']} us -> {sp['predicted_compute_only_efficiency']*100:.0f} %"
            + (f" (graph replay {sp['max_slab_us_graph']} us -> {sp['predicted_compute_only_efficiency_graph']*100:.0f} %)" if "max_slab_us_graph" in sp else "")
            for sp in s["splits"]))
    line = json.dumps(out)
    if len(line) > LINE_LIMIT:      # never again a line the driver's tail cannot hold: drop the optional keys, longest first
        for key in sorted((k for k in out if k not in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                                        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle")),
                          key=lambda k: -len(json.dumps(out[k]))):
            del out[key]
            line = json.dumps(out)
            if len(line) <= LINE_LIMIT:
                break
    sys.stderr.flush()
    print(line, flush=True)


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
                    help="N > 1: nccl = RCCL over xGMI, one GPU per rank (the measurement).  gloo = self-tests, never a measurement: with "
                         "--share-gpu the DRY RUN of the N-rank path on GPU 0 (HIP engine, host-staged collectives); without it the launcher / "
                         "sharding self-test on host memory, which needs HISPARSE_HIP_LIB=<libhisparse_cpu.so>")
    ap.add_argument("--share-gpu", action="store_true", help="--backend gloo: all N ranks open GPU 0 with the HIP library (dry run of the multi-GPU path on one GPU)")
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
    from hisparse_amd import datasets, device, host, sharding
    import bench_extras

    sub_steps = max(20, min(args.steps, 200))
    sub_warm = min(args.warmup, 20)
    if args.predict_scaling:      # only the strong-scaling prediction (bench.py --predict-scaling [--config mouse_gene])
        pred = bench_extras.predict_scaling(np, datasets, device, host, sharding, args.config or "mouse_gene", sub_steps, rank)
        emit({"predict_scaling": {"workload": pred["workload"], "unsplit_us": pred["unsplit_us"],
                                  "splits": [{k: v for k, v in sp.items() if k != "slabs"} for sp in pred["splits"]]}}, {"predict_scaling": pred}, [], [pred])
        return
    if args.config == "bm":       # only the reference's sweep (sw/bm.sh runs it in the mode of its bitstream: --impl)
        sweep_impl = args.impl or "fixed"
        paper7 = {n: {"fixed": fx, "float_pob": pb, "float_stall": ri} for n, fx, pb, ri in datasets.BM_FLOAT}
        rows, table = [], []
        for name, paper in datasets.BM_LIST:
            res, ctx = measure_single(np, datasets, device, host, name, sub_steps, sub_warm, impl_override=sweep_impl, rank=rank, with_spmm=False)
            ctx["eng"].close()
            del ctx
            rows.append(bench_extras.bm_entry(name, paper if sweep_impl == "fixed" else paper7.get(name, {}).get(sweep_impl), res, sweep_impl))
            table.append(summary_row(res))
        emit({"metric": f"SpMV GBPS / GOPS per matrix of sw/bm.sh, {sweep_impl} IMPL, 1 x MI355X",
              "bm_list": [{k: r[k] for k in ("matrix", "ms_per_step", "value", "gops", "frac_whole_step", "frac")} for r in rows]}, {"bm_list": rows}, table)
        return

    headline = args.config or "ogbl_ppa"
    details, table, scaling, bm_rows = {}, [], None, {}
    if not args.config and not args.quick:
        suite, measured = bench_extras.run_suite(np, datasets, device, host, sharding, sub_steps, sub_warm, rank)
        bm_rows = suite.pop("bm_rows")
        details.update(suite)
        scaling = suite["strong_scaling_prediction"]
        table = [summary_row(r) for r in measured.values()]
    res, ctx = measure_single(np, datasets, device, host, headline, args.steps, args.warmup, npz=args.npz, impl_override=args.impl,
                              cpu_seconds=0.0 if args.no_cpu_baseline else args.cpu_seconds, rank=rank)
    cpu_line, cpu_full = (None, None) if args.no_cpu_baseline else cpu_baseline_for(np, host, ctx, rank)
    if not args.quick and not args.npz:
        cold = bench_extras.mall_cold(np, datasets, device, host, ctx, args.steps, args.warmup, rank)
        res["roofline"]["frac_mall_cold"] = cold["frac_whole_job_round_robin"]
        res["mall_cold"] = cold
        bench_extras.quote_hbm_fraction(res)
    ctx["eng"].close()
    impl = ctx["impl"]
    table.append(summary_row(res))
    r = res["roofline"]
    tf = r["traffic_from"]
    roofline = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "kernel_ms_from", "frac_whole_step", "frac_event_pairs",
                                  "algorithmic_bytes_per_launch", "streamed_bytes_per_launch", "traffic")}
    roofline["frac_mall_cold"] = r.get("frac_mall_cold")
    roofline["traffic_from"] = (f"{tf.get('source', '')}: round {tf.get('round')}, commit {tf.get('git_head')}, kernel sources unchanged since: {tf.get('sources_unchanged_since')}; "
                                f"rocprofv3 kernel avg then {tf.get('kernel_us_rocprof')} us")[:400]
    if not args.quick and not args.npz:      # the counters of THIS run (two short rocprofv3 passes); the committed copy stays as the fallback and beside it
        live, how = bench_extras.live_traffic(headline, IMPL_NAMES[impl], r["kernel"], rank)
        if live is not None:
            roofline["traffic_committed_copy"] = roofline["traffic"]
            roofline["traffic"], roofline["traffic_from"] = round(live, 1), how
        else:
            log(rank, f"{headline}: live counter passes not available ({how}): roofline.traffic is the committed copy")
            roofline["traffic_from"] = (roofline["traffic_from"] + f" [live passes: {how}]")[:480]
    out = {
        "metric": "SpMV GBPS (8 B per non-zero per SpMV, sw/benchmark.cpp:312-346)",
        "value": res["value"], "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32 (Q8.24 fixed point, u64 row sums)" if impl == host.IMPL_FIXED else "f32",
        "data": "synthetic" if not args.npz else "file",
        "config": {"workload": res["workload"], "rows": res["rows"], "cols": res["cols"], "nnz": res["nnz"],
                   "partitions": res["partitions"], "stream_format": res["stream_format"], "col_slices": res["col_slices"], "parallelism": "row-slab x1"},
        "gops": res["gops"], "gibps_reference_formula": res["gibps_reference_formula"], "ms_per_step_synchronous": res["ms_per_step_synchronous"],
        "ms_per_step_graph_replay": res["ms_per_step_graph_replay"],
        "roofline": roofline, "cpu_baseline": cpu_line, "parity_vs_oracle": res["parity_vs_oracle"],
    }
    if "float_error" in res:
        out["float_error"] = {k: res["float_error"][k] for k in ("max_abs_err_vs_csim", "max_abs_y", "rows_over_csim_absolute_1e-4")}
    details["headline"] = dict(res, cpu_baseline=cpu_full, parity_pins=PARITY_PINS)
    if len(bm_rows) + 1 == len(datasets.BM_LIST):
        bm_rows[headline] = res
        details["bm_list"] = [bench_extras.bm_entry(name, paper, bm_rows[name]) for name, paper in datasets.BM_LIST]
    emit(out, details, table, scaling)


def main_distributed(args, rank, local_rank, world):
    import numpy as np
    import torch
    import torch.distributed as dist
    from hisparse_amd import datasets, device, host, sharding

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    rccl = args.backend == "nccl"
    dry = args.backend == "gloo" and args.share_gpu      # N processes on GPU 0: everything of the N-rank path except RCCL itself
    on_gpu = rccl or dry
    gpu_id = local_rank if rccl else 0
    if args.share_gpu and rccl:
        if rank == 0:
            fail("--share-gpu is the dry run over gloo (RCCL cannot run several ranks on one GPU): add --backend gloo")
        sys.exit(2)
    if on_gpu:
        have = visible_gpus()
        if gpu_id >= have:
            if rank == 0:
                fail(f"{world} ranks over RCCL need {world} GPUs, {have} visible" if rccl else "--share-gpu needs one GPU, none visible", n_gpus_requested=world, gpus_visible=have)
            sys.exit(2)
        torch.cuda.set_device(gpu_id)
        if os.path.basename(device._LIB_PATH) == "libhisparse_cpu.so":
            if rank == 0:
                fail("a GPU run with HISPARSE_HIP_LIB pointing at the host-thread library")
            sys.exit(2)
    elif os.path.basename(device._LIB_PATH) == "libhisparse_hip.so":
        if rank == 0:
            fail("--backend gloo without --share-gpu is the launcher / sharding self-test on host memory: point HISPARSE_HIP_LIB at "
                 "libhisparse_cpu.so (the HIP library has no host path and writes y to device memory)")
        sys.exit(2)
    dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    dist.barrier()
    n_gpus = world
    name = args.config or "mouse_gene"
    dev = f"cuda:{gpu_id}" if on_gpu else "cpu"
    ctl = dev if rccl else "cpu"                          # where the control-plane reductions (timings, flags) live
    spin_up_steps = SPIN_UP_STEPS if rccl else (100 if dry else 0)
    cuda_sync = torch.cuda.synchronize if on_gpu else (lambda: None)

    # ---- workload: this rank's row slab ---------------------------------------------------------------------------------
    t0 = time.perf_counter()
    cfg = datasets.CONFIGS[name]
    impl = host.impl_id(args.impl or cfg.impl)
    granule = 128 * (8 if impl == host.IMPL_FLOAT_STALL else 1)
    full_rows, whole = None, None
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
    eng = device.SpmvEngine(impl, device_id=gpu_id)
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
    if dry:      # host staging buffers of the gloo collective
        stage_in = torch.zeros(chunk, dtype=torch.int32).pin_memory()
        stage_out = torch.zeros(chunk * world, dtype=torch.int32).pin_memory()
    if on_gpu:
        main_stream = torch.cuda.Stream(device=dev)     # the legacy default stream has handle 0 = "the library's private stream"
        torch.cuda.set_stream(main_stream)
        torch.cuda.synchronize()
        eng.set_stream(main_stream.cuda_stream)
    pending = [None, None]
    step_no = [0]

    def all_gather(dst, src, async_op=False):
        """the y exchange: RCCL on device memory; in the dry run the same call pattern staged through host memory over gloo (synchronous)"""
        if not dry:
            return dist.all_gather_into_tensor(dst, src, async_op=async_op)
        stage_in.copy_(src, non_blocking=True)
        main_stream.synchronize()
        dist.all_gather_into_tensor(stage_out, stage_in)
        dst.copy_(stage_out, non_blocking=True)
        return None

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
        pending[cur] = all_gather(gathered[cur], y_chunks[cur], async_op=True)

    def sync():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None
        cuda_sync()
        eng.sync()

    def run_steps(gather, n):
        """n slab SpMVs: with an exchange after every step, one hs_run at a time (each must be complete in stream order before its gather);
        otherwise as ONE batch (hs_run_batch: the reference's NUM_RUNS loop as a unit -- enqueued from the library's C loop, the steps of a
        column-sliced slab carrying each other's combine pass, the last one settled before the call returns)"""
        if gather in ("final", "off") and on_gpu and step is plain_step:
            if n:
                eng.bind_device_result(y_chunks[0].data_ptr())
                eng.run_batch(n)
            return
        for _ in range(n):
            step(gather)

    plain_step = step

    def timed(gather, steps):
        run_steps(gather, spin_up_steps)
        sync()
        run_steps(gather, args.warmup)
        sync()
        dist.barrier()
        cuda_sync()
        t0 = time.perf_counter()
        run_steps(gather, steps)
        if gather == "final":
            all_gather(gathered[0], y_chunks[0])
        sync()
        cuda_sync()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- correctness of what is about to be timed: this rank's slab against the oracle, and the gathered buffer ------------------
    run_into(y_chunks[0])
    all_gather(gathered[0], y_chunks[0])
    sync()
    y_gpu = y_chunks[0][:packets.num_rows].cpu().numpy().view(np.uint32)
    everyone = gathered[0].cpu().numpy().view(np.uint32).reshape(world, chunk)
    if not np.array_equal(everyone[rank, :packets.num_rows], y_gpu):
        print(json.dumps({"error": "all-gathered y differs from the local slab", "rank": rank}))
        sys.exit(1)
    parity, _, t_cpu, _, _ = oracle_check(np, host, impl, packets, xw, y_gpu, 0.0)
    # ... and every OTHER rank's slot of the gathered buffer against that rank's own checksum (the layout the consumer of the gather sees)
    sums = [None] * world
    dist.all_gather_object(sums, int(y_gpu.astype(np.uint64).sum()))
    layout_ok = all(int(everyone[r, :rows_all[r]].astype(np.uint64).sum()) == sums[r] for r in range(world))
    flag = torch.tensor([1.0 if parity == "MISMATCH" or not layout_ok else 0.0], dtype=torch.float64, device=ctl)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if flag.item() > 0:
        if rank == 0:
            print(json.dumps({"error": "a rank's slab does not match the oracle, or the gathered layout is wrong", "config": name}))
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
            pg = peer_gather.PeerGather(dist, rank, world, chunk, device_id=gpu_id)
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
            ok = True
            for b in (0, 1):      # both buffers against what the collective gathered before the timing (same x, same matrix: the same y)
                got = pg.read(b)
                for r in range(world):
                    ok = ok and bool(np.array_equal(got[r, :rows_all[r]], everyone[r, :rows_all[r]]))
            flag2 = torch.tensor([0.0 if ok else 1.0], dtype=torch.float64, device=ctl)
            dist.all_reduce(flag2, op=dist.ReduceOp.MAX)
            dist.barrier()
            eng.bind_device_result(y_chunks[0].data_ptr())
            pg.close()
            push = {"ms_per_step": round(push_elapsed / args.steps * 1e3, 5), "ms_per_step_added": round((push_elapsed - compute_elapsed) / args.steps * 1e3, 5),
                    "equals_collective_on_every_rank": flag2.item() == 0.0}
        except Exception as e:      # noqa: BLE001 -- the push path is an extra measurement
            push = {"error": f"{type(e).__name__}: {e}"[:200]}
            log(rank, f"peer-store gather skipped: {e}")
    tot = torch.tensor([float(nnz)], dtype=torch.float64, device=ctl)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    total_nnz = float(tot.item())
    if on_gpu:
        eng.set_stream(None)
    kernel_ms = eng.time_kernel(min(args.warmup, 20), args.steps) / args.steps
    dist.barrier()
    # strong scaling: the SAME matrix, unsplit, on rank 0's GPU alone -- the N = 1 point the N-GPU numbers of this workload belong to
    # (bench.py --gpus 1 without a launcher measures the ogbl-ppa headline instead)
    one_gpu = None
    if whole is not None:
        with device.SpmvEngine(impl, device_id=gpu_id) as eng1:
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
            one_gpu = {"n_gpus": 1, "ms_per_step": round(one * 1e3, 5), "value": round(8.0 * whole.nnz / one / 1e9, 2), "unit": "GB/s"}
        del whole
    dist.barrier()

    out = None
    if rank == 0:
        per_step = elapsed / args.steps
        value = 8.0 * total_nnz / per_step / 1e9
        achieved = 8.0 * nnz / (kernel_ms * 1e-3) / 1e9
        gather_text = {"step": " + all_gather(y) every step (overlapped with the next SpMV)", "final": " + one final all_gather(y)", "off": ""}[args.gather]
        backend = ("nccl (RCCL)" if rccl else f"gloo, {world} processes sharing GPU 0, HIP engine, host-staged collectives: DRY RUN of the N-rank path, NOT a measurement" if dry
                   else f"gloo on host memory with {os.path.basename(device._LIB_PATH)}: launcher self-test, NOT a measurement")
        out = {
            "metric": "SpMV GBPS (8 B per non-zero per SpMV, sw/benchmark.cpp:312-346)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(per_step * 1e3, 5), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "backend": backend, "gather": args.gather,
            "dtype": "u32 (Q8.24 fixed point, u64 row sums)" if impl == host.IMPL_FIXED else "f32",
            "data": "synthetic" if not args.npz else "file",
            "config": {"workload": f"{name}, {IMPL_NAMES[impl]} IMPL, v={packets.vb_bank} o={packets.ob_bank}", "rows": full_rows or true_rows * n_gpus,
                       "cols": packets.num_cols, "nnz_per_gpu": int(nnz), "nnz_total": int(total_nnz), "slab_rows_rank0": true_rows,
                       "parallelism": f"row-slab x{n_gpus}, balanced by non-zeros" + gather_text},
            "gops": round(2.0 * total_nnz / per_step / 1e9, 2),
            "frac_whole_step": round(value / (HBM_PEAK_GBS * n_gpus), 4),
            "compute_only": {"ms_per_step": round(compute_elapsed / args.steps * 1e3, 5), "value": round(8.0 * total_nnz / (compute_elapsed / args.steps) / 1e9, 2),
                             "frac_whole_step": round(8.0 * total_nnz / (compute_elapsed / args.steps) / 1e9 / (HBM_PEAK_GBS * n_gpus), 4)},
            "same_workload_on_one_gpu": one_gpu,
            "exchange": {"bytes_per_rank_per_gather": int(chunk) * 4, "ms_per_step_added": round((elapsed - compute_elapsed) / args.steps * 1e3, 5)},
            "exchange_every_step": {"ms_per_step": round(step_elapsed / args.steps * 1e3, 5),
                                    "value": round(8.0 * total_nnz / (step_elapsed / args.steps) / 1e9, 2)},
            "exchange_push": push,
            "roofline": {"bound": "hbm", "kernel": kernel_name(stats), "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "kernel_ms": round(kernel_ms, 5),
                         "kernel_ms_from": "hs_time_kernel on rank 0's slab: one HIP event pair around K back-to-back launches of the kernel alone, / K",
                         "algorithmic_bytes_per_launch": int(8 * nnz), "streamed_bytes_per_launch": int(stats["stream_bytes"]), "traffic": None},
            "cpu_baseline": None if args.no_cpu_baseline else {"value": round(8.0 * nnz / t_cpu / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                                                               "sample": f"rank 0's slab, 1 SpMV through oracle/cpu_ref.c (1 thread), {t_cpu*1e3:.1f} ms"},
            "parity_vs_oracle": parity + " (every rank's slab; gathered layout checked on every rank)",
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
        details = dict(out, preprocess_s={"format_csr2cpsr": round(t_fmt, 3), "device_load_retile": round(stats["load_seconds"], 3)}, slab_rows=rows_all)
        emit(out, {"distributed": details}, [])


if __name__ == "__main__":
    main()
