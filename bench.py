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
    roofline.frac               8 nnz / (AVERAGE duration of the SpMV kernel over all launches) / 8 TB/s -- hs_time_kernel: ONE HIP event pair
                                around K back-to-back launches of the kernel alone, no warm-up launches, regions entered from an idle stream:
                                the average `rocprofv3 --kernel-trace --stats` prints (profiles/; roofline.rocprofv3_live = the same from a
                                pass run BY this script)
    roofline.frac_steady        the same with warm-up launches in front (the steady state; what rounds 4-5 printed as `frac`)
    roofline.launches_per_step  2 on column-sliced plans (kernel + combine_slices_kernel): `frac` prices the first launch alone
    roofline.frac_whole_step    8 nnz / (wall time per step: kernel + slice-combine pass + launch gaps) / 8 TB/s   (= value / 8000)
    roofline.frac_event_pairs   the kernel inside whole steps, a HIP event pair around EVERY launch (each pair adds ~3 us)
`roofline.traffic` = HBM bytes per launch from two short rocprofv3 counter passes run BY this script (bench_extras.live_traffic; the
committed copy of profiles/hbm_traffic.json beside it and as the fallback).
Everything else a default run measures (the other BASELINE configurations, the reference's whole sweep sw/bm.sh in all numeric modes,
MALL-cold round-robin legs, the strong-scaling prediction: bench_extras.py) goes to `bench_details.json` next to this file and, one row
per matrix, to stderr.  `--config NAME` measures only that configuration; `--quick` skips the extras.

N > 1, nothing named (the driver's series): `value` = the N = 1 workload carried to N GPUs the way the path shards -- WEAK scaling: every rank owns
an ogbl-ppa-sized row slab (own seed) of a matrix N slabs tall and all of x (hisparse_amd/sharding.py), no collective inside the SpMV, K slab
SpMVs and ONE final all-gather of the y slabs over RCCL (`gather: final`, north_star's "a final RCCL gather over xGMI"; the reference too runs
its NUM_RUNS launches and collects y once, sw/benchmark.cpp:318-346); per-GPU work is fixed, so value(N) / (N x value(1)) reads as scaling
efficiency.  In the same line under `baseline_config_4`: BASELINE.json configs[4] -- mouse_gene, ONE matrix split into N row slabs by non-zero
count (strong), per-rank plans and the one-GPU prediction of its slabs beside the measured ones.  `--config NAME` / `--scale-matrix NAME`
(+ `--scaling strong|weak`) measures that one workload instead.  Beside `value`: the same K SpMVs with y left sharded (`compute_only`),
with an all-gather after EVERY SpMV (`exchange_every_step`) and with that gather done by peer stores (`exchange_push`).
`--backend gloo --share-gpu` is the DRY RUN of that path on one GPU: N processes on GPU 0, the HIP engine, host-staged gloo collectives.
The N-rank leg lives in bench_dist.py, the sweeps behind the details file in bench_extras.py.
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
    # K steps of a small matrix are over in 0.1-0.3 ms, of which the final synchronisation and the first launch's latency are 20-70 us:
    # where K steps took under 2 ms, the same loop over enough steps for ~5 ms is reported BESIDE it (never `value`)
    long_steps, ms_long = 0, None
    if elapsed < 2e-3:
        long_steps = int(min(5000, max(steps * 2, 5e-3 / (elapsed / steps))))
        t0 = time.perf_counter()
        eng.run_batch(long_steps)
        eng.sync()
        ms_long = (time.perf_counter() - t0) / long_steps * 1e3
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
    kernel_ms_steady = eng.time_kernel(min(warmup, 20), steps) / steps
    # ... and the figure that prices roofline.frac (SURVEY 8(d) says AVERAGE launch duration, which is what `rocprofv3 --stats`
    # prints -- over EVERY dispatch, the first ones after an idle stream included -- not the steady state): the same event pair around K
    # launches with NO warm-up launches in front, each region entered from an idle, synchronised stream, averaged over several regions
    regions = max(5, min(25, 500 // max(steps, 1)))
    region_ms = []
    for _ in range(regions):
        eng.sync()
        region_ms.append(eng.time_kernel(0, steps))
    # (a region that took more than 1.5 x the median one is a hiccup of the box -- one such region of 25 put mouse_gene's "kernel alone" 20 % above its
    #  whole step in a run of round 6 -- and is left out; `regions_dropped` says how many)
    median_region = sorted(region_ms)[len(region_ms) // 2]
    kept = [t for t in region_ms if t <= 1.5 * median_region]
    kernel_ms = max(sum(kept) / (len(kept) * steps), kernel_ms_steady)      # (never better than the steady state it contains)
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
        "ms_per_step_long_run": round(ms_long, 5) if ms_long else None, "long_run_steps": long_steps or None,
        "spin_up_steps": spun,
        "gibps_reference_formula": round(8.0 * nnz / 2 ** 30 / (elapsed / steps), 2),
        "frac_whole_step": round(value / HBM_PEAK_GBS, 4),
        "roofline": {"bound": "hbm", "kernel": kernel_name(stats),
                     "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "kernel_ms": round(kernel_ms, 5),
                     "kernel_ms_from": f"hs_time_kernel: one HIP event pair around K back-to-back launches of the kernel alone, NO warm-up launches, each of {regions} regions entered from an idle stream; average over all {len(kept) * steps} launches" + (f" ({regions - len(kept)} region(s) of more than 1.5 x the median left out)" if len(kept) < regions else ""),
                     "frac_steady": round(8.0 * nnz / (kernel_ms_steady * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernel_ms_steady": round(kernel_ms_steady, 5),
                     # one SpMV = this many launches: a column-sliced plan is kernel + combine_slices_kernel; `frac` is the FIRST one alone,
                     # frac_whole_step the whole SpMV
                     "launches_per_step": 2 if stats["col_slices"] > 1 else 1,
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
    long_run = f"  [{res['ms_per_step_long_run']*1e3:.1f} us/step over {res['long_run_steps']} steps]" if res.get("ms_per_step_long_run") else ""
    return (f"{res['matrix']}/{res['impl']}".ljust(28) + f"{res['stream_format'][:18]:<19}{res['col_slices']:>2} {res['ms_per_step']*1e3:8.1f} {res['value']:7.0f} "
            f"{res['gops']:6.0f} {res['frac_whole_step']*100:6.1f} {r['frac']*100:6.1f} " + (f"{cold*100:6.1f}" if cold is not None else "     -") + "  " + short_parity(res["parity_vs_oracle"]) + long_run)


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
                                                        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity_vs_oracle", "baseline_config_4", "compute_only",
                                                        "same_workload_on_one_gpu")),
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
    ap.add_argument("--scale-matrix", default=None, choices=["mouse_gene", "hollywood", "ogbn_products", "ogbl_ppa", "pokec", "gplus"],
                    help="N > 1: the matrix to shard, by name (same as --config; default mouse_gene = BASELINE.json configs[4]).  hollywood and "
                         "ogbn_products are the ones a curve can look right on: their 1/8 slabs are still 25-32 us of streaming (one-GPU prediction: "
                         "70 % / 79 % compute-only at 8 ways), where a 1/8 slab of mouse_gene is 3.6 us of bytes against a 3 us launch floor")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: strong = ONE matrix split N ways by non-zeros; weak = one matrix-sized row slab per rank (the matrix is N slabs tall).  Default: with "
                         "a matrix named (--config / --scale-matrix) strong; with nothing named the driver's series -- ogbl_ppa weak as `value` (the N = 1 line's "
                         "workload, per-GPU work fixed) AND mouse_gene strong (BASELINE.json configs[4]) in the same line under `baseline_config_4`")
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
        import bench_dist
        return bench_dist.main_distributed(args, rank, local_rank, world)

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
    roofline = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "kernel_ms", "kernel_ms_from", "frac_steady", "kernel_ms_steady", "launches_per_step",
                                  "frac_whole_step", "frac_event_pairs", "algorithmic_bytes_per_launch", "streamed_bytes_per_launch", "traffic")}
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
        # ... and the launch duration as `rocprofv3 --kernel-trace --stats` sees it, in this run: the average over all dispatches must agree with `frac`'s kernel_ms
        trace, how_trace = bench_extras.live_kernel_trace(headline, IMPL_NAMES[impl], r["kernel"], rank)
        if trace is not None:
            roofline["rocprofv3_live"] = {"kernel_avg_us": trace["avg_us"], "kernel_steady_median_us": trace["steady_median_us"], "dispatches": trace["calls"],
                                          "combine_avg_us": trace["combine_avg_us"], "frac": round(8.0 * res["nnz"] / (trace["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
            # Two clocks for one kernel: the HIP-event figure is the launch-to-launch PERIOD of back-to-back launches (a launch's ramp overlaps its
            # predecessor's tail), rocprofv3's is every dispatch's own begin-to-end duration and is what `--stats` averages.  `frac` is priced with the
            # LONGER of the two, so that it can never read better than the committed rocprofv3 summaries ("whose average must agree").
            if trace["avg_us"] * 1e-3 > roofline["kernel_ms"]:
                roofline["kernel_ms_hip_events"], roofline["frac_hip_events"] = roofline["kernel_ms"], roofline["frac"]
                roofline["kernel_ms"] = round(trace["avg_us"] * 1e-3, 5)
                roofline["achieved"] = round(8.0 * res["nnz"] / (trace["avg_us"] * 1e-6) / 1e9, 2)
                roofline["frac"] = round(roofline["achieved"] / HBM_PEAK_GBS, 4)
                roofline["kernel_ms_from"] = (f"rocprofv3 --kernel-trace --stats run by this script: average over all {trace['calls']} dispatches (longer than the HIP-event period "
                                              f"of back-to-back launches, kernel_ms_hip_events: a launch's ramp overlaps its predecessor's tail)")
        else:
            log(rank, f"{headline}: live rocprofv3 --stats pass not available ({how_trace})")
        res["roofline"].update({k: roofline[k] for k in ("achieved", "frac", "kernel_ms", "kernel_ms_from", "rocprofv3_live", "kernel_ms_hip_events", "frac_hip_events") if k in roofline})      # the details file says what the line says
    out = {
        "metric": "SpMV GBPS (8 B per non-zero per SpMV, sw/benchmark.cpp:312-346)",
        "value": res["value"], "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,      # (N = 1 of the weak series bench_dist.py continues: one ogbl-ppa-sized slab per GPU)
        "dtype": "u32 (Q8.24 fixed point, u64 row sums)" if impl == host.IMPL_FIXED else "f32",
        "data": "synthetic" if not args.npz else "file",
        "config": {"workload": res["workload"], "rows": res["rows"], "cols": res["cols"], "nnz": res["nnz"],
                   "partitions": res["partitions"], "stream_format": res["stream_format"], "col_slices": res["col_slices"], "parallelism": "row-slab x1"},
        "gops": res["gops"], "gibps_reference_formula": res["gibps_reference_formula"], "ms_per_step_synchronous": res["ms_per_step_synchronous"],
        "ms_per_step_graph_replay": res["ms_per_step_graph_replay"],
        "spin_up_steps": res["spin_up_steps"],      # untimed steps in front of the W warm-up steps (clocks back up after the CPU legs): `warmup` alone understates what precedes the timed region
        "roofline": roofline, "cpu_baseline": cpu_line, "parity_vs_oracle": res["parity_vs_oracle"],
    }
    if "float_error" in res:
        out["float_error"] = {k: res["float_error"][k] for k in ("max_abs_err_vs_csim", "max_abs_y", "rows_over_csim_absolute_1e-4")}
    details["headline"] = dict(res, cpu_baseline=cpu_full, parity_pins=PARITY_PINS)
    if len(bm_rows) + 1 == len(datasets.BM_LIST):
        bm_rows[headline] = res
        details["bm_list"] = [bench_extras.bm_entry(name, paper, bm_rows[name]) for name, paper in datasets.BM_LIST]
    emit(out, details, table, scaling)


if __name__ == "__main__":
    main()
