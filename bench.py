#!/usr/bin/env python3
"""bench.py — SpMV throughput of the MI355X hot path on the reference's headline workload.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config ogbl_ppa] [--npz FILE]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one full SpMV (every row partition) y = A x through the drop-in C-ABI, with the matrix
(re-tiled at load time), x and y already resident in HBM.  Metric (BASELINE.json): the reference's
"data throughput" of sw/benchmark.cpp:312-346 — 8 bytes per non-zero per SpMV — in decimal GB/s, plus
GOPS (2 flops per non-zero) and the GiB-based number the reference prints, and the fraction of the
8 TB/s HBM roofline.

N = 1 workload = BASELINE.json configs[1]: ogbl-ppa (seeded stand-in, see hisparse_amd/datasets.py),
fixed-point IMPL, default banks.  N > 1: every rank owns one row slab; `--scaling weak` (default) gives
every rank a slab the size of the whole N = 1 matrix (the global matrix is N slabs tall), `--scaling
strong` splits the one matrix by non-zero count.  The ranks' y slabs are all-gathered over RCCL each
step (the only exchange of the path; `--no-gather` leaves y sharded like the reference leaves it in HBM).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md: 8.0 TB/s spec)


def log(rank, *a):
    if rank == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def ensure_built():
    need = [os.path.join(ROOT, "hisparse_amd", "lib", n) for n in ("libhisparse_host.so", "libhisparse_hip.so")]
    need.append(os.path.join(ROOT, "oracle", "liboracle.so"))
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


def read_traffic(kernel_launch_bytes):
    """HBM bytes per launch from a committed rocprofv3 --pmc summary (profiles/hbm_traffic.json), if present."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="ogbl_ppa")
    ap.add_argument("--npz", default=None, help="real dataset file instead of the seeded stand-in")
    ap.add_argument("--impl", default=None, help="override the config's numeric mode")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--gather", choices=["final", "step", "off"], default="final",
                    help="N > 1: all-gather the y slabs once after the timed SpMVs (default), after every SpMV (overlapped), or never")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true", help="use the torch.distributed path even for one GPU")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist_mode = world > 1 or args.force_dist
    if args.gpus != world and world > 1:
        log(rank, f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    n_gpus = world

    if rank == 0:
        ensure_built()
    torch = dist = None
    if dist_mode:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)
        dist.barrier()

    import numpy as np
    from hisparse_amd import datasets, device, host, sharding

    # ---- workload ------------------------------------------------------------------------------------
    t0 = time.perf_counter()
    cfg = datasets.CONFIGS[args.config]
    impl = host.impl_id(args.impl or cfg.impl)
    granule = 128 * (8 if impl == host.IMPL_FLOAT_STALL else 1)
    if n_gpus > 1 and args.scaling == "strong":
        _, full = datasets.load(args.config, path=args.npz)
        indptr, indices, data = full.arrays()
        bounds = sharding.split_rows_by_nnz(indptr, n_gpus, granule)
        lo, hi = bounds[rank], bounds[rank + 1]
        ip, ix, dv = sharding.slab_arrays(indptr, indices, data, lo, hi)
        csr = host.CSRMatrix.from_arrays(hi - lo, full.num_cols, ip, ix, dv)
        del full, indptr, indices, data
    elif n_gpus > 1:
        # weak: rank r owns slab r of a matrix that is n_gpus slabs tall; same generator, different seed per slab
        c = cfg
        csr = host.CSRMatrix.generate(c.kind, c.rows, c.cols, a=c.a, b=c.b, c=c.c, seed=c.seed + 1000 * rank) if not args.npz \
            else host.load_csr_matrix_from_float_npz(args.npz)
    else:
        _, csr = datasets.load(args.config, path=args.npz)
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

    eng = device.SpmvEngine(impl, device_id=local_rank)
    eng.load_matrix(packets)
    eng.load_vector(xw)
    stats = eng.stats()
    log(rank, f"{args.config}: {packets.num_rows}x{packets.num_cols}, nnz {nnz}, partitions {packets.num_row_partitions}x{packets.num_col_partitions}, "
              f"generate {t_gen:.2f}s format {t_fmt:.2f}s device-load {stats['load_seconds']:.2f}s, CPSR {stats['cpsr_bytes']/1e6:.0f} MB -> stream {stats['stream_bytes']/1e6:.0f} MB")

    # ---- distributed plumbing: y slab inside an all-gather buffer -------------------------------------
    # `final`: one all-gather of the y slabs after the K timed SpMVs (inside the timed region) -- the path itself has no
    # exchange step: rows are independent and the reference leaves y in device memory.  `step`: gather after every SpMV
    # (iterative-solver pattern, y_k feeds x_k+1), double-buffered so that the gather of step k overlaps the SpMV of k+1.
    gather = args.gather if dist_mode else "off"
    y_chunks = gathered = None
    pending = [None, None]
    if dist_mode:
        rows_all = [None] * world
        dist.all_gather_object(rows_all, packets.num_rows)
        chunk = max(rows_all)
        dev = f"cuda:{local_rank}"
        y_chunks = [torch.zeros(chunk, dtype=torch.int32, device=dev) for _ in range(2)]
        gathered = [torch.zeros(chunk * world, dtype=torch.int32, device=dev) for _ in range(2)]
        # one explicit (non-default) stream for both the SpMV kernels and the point RCCL synchronises against: the legacy
        # default stream has handle 0, which hs_set_stream reads as "use the library's private stream"
        main_stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(main_stream)
        torch.cuda.synchronize()
        eng.set_stream(main_stream.cuda_stream)
        eng.bind_device_result(y_chunks[0].data_ptr())
    step_no = [0]

    def step():
        if gather != "step":
            eng.run()
            return
        cur = step_no[0] & 1
        step_no[0] += 1
        if pending[cur] is not None:
            pending[cur].wait()              # the gather that read this slab two steps ago (stream-level wait, no host sync)
            pending[cur] = None
        eng.bind_device_result(y_chunks[cur].data_ptr())
        eng.run()
        pending[cur] = dist.all_gather_into_tensor(gathered[cur], y_chunks[cur], async_op=True)

    def final_gather():
        if gather == "final":
            dist.all_gather_into_tensor(gathered[0], y_chunks[0])

    def sync():
        if dist_mode:
            for i in (0, 1):
                if pending[i] is not None:
                    pending[i].wait()
                    pending[i] = None
            torch.cuda.synchronize()
        eng.sync()

    # ---- correctness of what is about to be timed (and the CPU baseline) --------------------------------
    step()
    final_gather()
    sync()
    y_gpu = eng.read_result() if not dist_mode else y_chunks[0][:packets.num_rows].cpu().numpy().view(np.uint32)
    if gather != "off":   # the gathered buffer must hold this rank's slab at its offset (kernel -> RCCL ordering on the shared stream)
        mine = gathered[0][rank * y_chunks[0].numel(): rank * y_chunks[0].numel() + packets.num_rows].cpu().numpy().view(np.uint32)
        if not np.array_equal(mine, y_gpu):
            print(json.dumps({"error": "all-gathered y differs from the local slab", "rank": rank}))
            sys.exit(1)
    cpu_baseline = None
    parity = "unchecked"
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        chans = [packets.channel_ptr(c)[0] for c in range(16)]
        t0 = time.perf_counter()
        y_cpu = orc.spmv(impl, chans, xw, packets.num_rows, packets.num_cols, packets.num_row_partitions, packets.num_col_partitions,
                         packets.ob_bank, packets.vb_bank)
        t_one = time.perf_counter() - t0
        reps = max(1, min(20, int(args.cpu_seconds / max(t_one, 1e-6))))
        t0 = time.perf_counter()
        for _ in range(reps - 1):
            orc.spmv(impl, chans, xw, packets.num_rows, packets.num_cols, packets.num_row_partitions, packets.num_col_partitions,
                     packets.ob_bank, packets.vb_bank)
        t_cpu = (time.perf_counter() - t0 + t_one) / reps
        if impl == host.IMPL_FIXED:
            parity = "bit-exact" if np.array_equal(y_gpu, y_cpu) else "MISMATCH"
        else:
            parity = "within 1e-4" if np.allclose(y_gpu.view(np.float32), y_cpu.view(np.float32), rtol=1e-4, atol=1e-4) else "MISMATCH"
        cpu_baseline = {"value": round(8.0 * nnz / t_cpu / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                        "sample": f"{reps} full SpMV(s) of the same matrix through oracle/cpu_ref.c (csim-equivalent restatement, 1 thread), {t_cpu*1e3:.1f} ms each",
                        "gops": round(2.0 * nnz / t_cpu / 1e9, 4), "host_cpus": orc.usable_cores()}
        # context only: the same restatement with one host thread per cluster (the 16 clusters are independent)
        try:
            threads = min(16, orc.usable_cores())
            t0 = time.perf_counter()
            y_par = orc.spmv_per_channel_threads(impl, chans, xw, packets.num_rows, packets.num_cols, packets.num_row_partitions,
                                                 packets.num_col_partitions, packets.ob_bank, packets.vb_bank, threads=threads)
            t_par = time.perf_counter() - t0
            if np.array_equal(y_par, y_cpu):
                cpu_baseline["cpsr_one_thread_per_cluster"] = {"value": round(8.0 * nnz / t_par / 1e9, 3), "unit": "GB/s", "cores": threads,
                                                               "gops": round(2.0 * nnz / t_par / 1e9, 3), "sample": f"1 SpMV, {t_par*1e3:.1f} ms"}
        except Exception as e:
            log(rank, f"per-cluster-thread baseline skipped: {e}")
        # context only: plain float32 CSR loop (compute_ref, csim.cpp:143-158) with OpenMP over every host core
        try:
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
            cpu_baseline["csr_openmp_best_thread_count"] = {"value": round(8.0 * nnz / t_omp / 1e9, 3), "unit": "GB/s", "cores": threads,
                                                            "gops": round(2.0 * nnz / t_omp / 1e9, 3),
                                                            "sample": f"5 float32 CSR SpMVs, {t_omp*1e3:.2f} ms each (best of 16 / 64 / all host threads)"}
            del ip, ix, dv
        except Exception as e:  # the context number must never break the bench line
            log(rank, f"csr_openmp baseline skipped: {e}")
        log(rank, f"oracle: {t_cpu*1e3:.1f} ms per SpMV on 1 core; GPU result {parity}")
        if parity == "MISMATCH":
            print(json.dumps({"error": "GPU result does not match the oracle", "config": args.config}))
            sys.exit(1)

    # ---- timing -----------------------------------------------------------------------------------------
    # The GPU sat idle for seconds during the CPU baseline above and has dropped its clocks; the requested W warm-up steps
    # (a few ms at most) are not enough to bring them back.  A fixed, untimed spin-up first (reported as spin_up_steps).
    SPIN_UP_STEPS = 300
    for _ in range(SPIN_UP_STEPS):
        step()
    sync()
    for _ in range(args.warmup):
        step()
    sync()
    if dist_mode:
        dist.barrier()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    final_gather()
    sync()
    if dist_mode:
        torch.cuda.synchronize()
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_mode:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([float(nnz)], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_nnz = float(tot.item())
    else:
        total_nnz = float(nnz)

    # the kernel alone, HIP events on the launch stream, same K launches (rank 0's slab)
    if dist_mode:
        eng.set_stream(None)
    ev_total_ms, ev_kernel_ms = eng.time_runs(0, args.steps)
    kernel_ms = ev_kernel_ms / args.steps
    if dist_mode:
        dist.barrier()

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = 8.0 * total_nnz / (elapsed / args.steps) / 1e9
        achieved = 8.0 * nnz / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "SpMV GBPS (8 B per non-zero per SpMV, sw/benchmark.cpp:312-346; GOPS and % of the HBM roofline alongside)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "spin_up_steps": SPIN_UP_STEPS,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": args.scaling if n_gpus > 1 else "weak",
            "vs_baseline": None, "dtype": "u32 Q8.24 fixed point (u64 accumulate)" if impl == host.IMPL_FIXED else "f32",
            "data": "synthetic" if not args.npz else "file",
            "config": {"workload": f"{args.config}, {['fixed', 'float_pob', 'float_stall'][impl]} IMPL, v={packets.vb_bank} o={packets.ob_bank}",
                       "rows": true_rows, "cols": packets.num_cols, "nnz_per_gpu": int(nnz), "nnz_total": int(total_nnz),
                       "partitions": f"{packets.num_row_partitions}x{packets.num_col_partitions}",
                       "parallelism": f"row-slab x{n_gpus}" + ({"final": " + one final all_gather(y) over RCCL", "step": " + all_gather(y) over RCCL every step (overlapped)", "off": ""}[gather] if n_gpus > 1 else "")},
            "gops": round(2.0 * total_nnz / (elapsed / args.steps) / 1e9, 2),
            "gibps_reference_formula": round(8.0 * total_nnz / 2 ** 30 / (elapsed / args.steps), 2),
            "hbm_roofline_fraction_whole_job": round(value / (HBM_PEAK_GBS * n_gpus), 4),
            "roofline": {"bound": "hbm", "kernel": "spmv_rowblock_kernel", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "kernel_ms": round(kernel_ms, 5),
                         "algorithmic_bytes_per_launch": int(8 * nnz), "streamed_bytes_per_launch": int(stats["stream_bytes"]),
                         "traffic": read_traffic(8 * nnz)},
            "cpu_baseline": cpu_baseline,
            "parity_vs_oracle": parity,
            "preprocess_s": {"format_csr2cpsr": round(t_fmt, 3), "device_load_retile": round(stats["load_seconds"], 3)},
        }
    eng.close()
    if dist_mode:
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
