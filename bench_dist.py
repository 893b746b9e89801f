"""bench_dist.py -- the N-rank leg of bench.py (`bench.py --gpus N`, one process per GPU; BASELINE.json configs[4]).

ONE matrix (default mouse_gene) split into N row slabs by non-zero count (`--scaling strong`; hisparse_amd/sharding.py), every rank formats
and loads its slab and holds all of x; no collective inside the SpMV.  Timed = `value`: K slab SpMVs and ONE final all-gather of the y slabs
over RCCL (`gather: final`); in the same line: `compute_only`, `exchange_every_step` (overlapped, double-buffered), `exchange_push` (peer
stores over IPC handles).  Every rank's slab is checked against the oracle, and the gathered layout against every rank's checksum, before
anything is timed.  `--backend gloo --share-gpu`: the DRY RUN of all of this on one GPU (N processes on GPU 0, the HIP engine, host-staged
collectives); `--backend gloo` alone: the launcher / sharding self-test on host memory with libhisparse_cpu.so.  Never measurements.
"""
import json
import os
import sys
import time

import bench as B


def one_gpu_prediction(name, impl_name, n):
    """What ONE GPU predicted for this split (bench.py's default run / --predict-scaling: every slab of the N-way split timed on one GPU,
    bench_extras.predict_scaling), so that the first real N-GPU run can be read against it slab by slab.  Looked up in the details file a
    1-GPU run of this checkout left behind, then in the committed copies under profiles/ (newest round first); None when nobody predicted it."""
    import glob
    paths = [B.DETAILS_FILE] + sorted(glob.glob(os.path.join(B.ROOT, "profiles", "r*_bench_details.json")), reverse=True)
    for path in paths:
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for pred in d.get("strong_scaling_prediction") or []:
            if pred.get("workload") != f"{name}, {impl_name} IMPL":
                continue
            for sp in pred.get("splits", []):
                if sp.get("n_gpus") == n:
                    return {"source": os.path.relpath(path, B.ROOT), "unsplit_us": pred.get("unsplit_us"), "max_slab_us": sp.get("max_slab_us"),
                            "predicted_compute_only_efficiency": sp.get("predicted_compute_only_efficiency"),
                            "slabs": [{k: sl.get(k) for k in ("rank", "rows", "nnz", "us", "plan")} for sl in sp.get("slabs", [])]}
    return None


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
            B.fail("--share-gpu is the dry run over gloo (RCCL cannot run several ranks on one GPU): add --backend gloo")
        sys.exit(2)
    if on_gpu:
        have = B.visible_gpus()
        if gpu_id >= have:
            if rank == 0:
                B.fail(f"{world} ranks over RCCL need {world} GPUs, {have} visible" if rccl else "--share-gpu needs one GPU, none visible", n_gpus_requested=world, gpus_visible=have)
            sys.exit(2)
        torch.cuda.set_device(gpu_id)
        if os.path.basename(device._LIB_PATH) == "libhisparse_cpu.so":
            if rank == 0:
                B.fail("a GPU run with HISPARSE_HIP_LIB pointing at the host-thread library")
            sys.exit(2)
    elif os.path.basename(device._LIB_PATH) == "libhisparse_hip.so":
        if rank == 0:
            B.fail("--backend gloo without --share-gpu is the launcher / sharding self-test on host memory: point HISPARSE_HIP_LIB at "
                 "libhisparse_cpu.so (the HIP library has no host path and writes y to device memory)")
        sys.exit(2)
    dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    dist.barrier()
    n_gpus = world
    dev = f"cuda:{gpu_id}" if on_gpu else "cpu"
    ctl = dev if rccl else "cpu"                          # where the control-plane reductions (timings, flags) live
    spin_up_steps = B.SPIN_UP_STEPS if rccl else (100 if dry else 0)
    cuda_sync = torch.cuda.synchronize if on_gpu else (lambda: None)


    def measure(name, scaling, brief=False):
        """One workload on the N ranks: (the JSON record on rank 0 -- None elsewhere --, what the details file gets beside it).  brief: the K SpMVs + final gather and
        the compute-only leg only (the secondary record of a default run)."""
        # ---- workload: this rank's row slab ---------------------------------------------------------------------------------
        t0 = time.perf_counter()
        cfg = datasets.CONFIGS[name]
        impl = host.impl_id(args.impl or cfg.impl)
        granule = 128 * (8 if impl == host.IMPL_FLOAT_STALL else 1)
        full_rows, whole = None, None
        if scaling == "strong":
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
        B.log(rank, f"{name} ({scaling}): slab {packets.num_rows}x{packets.num_cols}, nnz {nnz}, generate {t_gen:.2f}s format {t_fmt:.2f}s "
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
        parity, _, t_cpu, _, _ = B.oracle_check(np, host, impl, packets, xw, y_gpu, 0.0)
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
        step_elapsed = elapsed if args.gather == "step" else None if brief else timed("step", args.steps)      # an all-gather after every SpMV (iterative callers)
        # ---- the same K steps with the gather done by PEER STORES instead of a collective (hs_push_result; hisparse_amd/peer_gather.py): every
        #      rank's kernels write y into its slot of its own gather buffer and one small kernel pushes the slab into every peer's buffer over
        #      xGMI.  Reported beside the RCCL figure; a failure here (IPC not available) is reported, not fatal.
        push = None
        if on_gpu and not brief:
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
                B.log(rank, f"peer-store gather skipped: {e}")
        tot = torch.tensor([float(nnz)], dtype=torch.float64, device=ctl)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_nnz = float(tot.item())
        # every rank's own slab on its own clock, no barrier around it: K steps as one batch between two local synchronisations -- the quantity
        # the one-GPU prediction (bench_extras.predict_scaling) times per slab, so the two can be compared slab by slab; the barrier-timed
        # `compute_only` above is the slowest of these plus whatever the ranks' skew adds
        sync()
        local_best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            run_steps("off", args.steps)
            sync()
            local_best = min(local_best, (time.perf_counter() - t0) / args.steps)
        if on_gpu:
            eng.set_stream(None)
        kernel_ms = eng.time_kernel(min(args.warmup, 20), args.steps) / args.steps
        plan = {"rank": rank, "rows": int(true_rows), "padded_rows": int(packets.num_rows), "nnz": int(nnz),
                "plan": f"{device.STREAM_FORMATS[stats['stream_format']]}" + (" (light kernel)" if stats.get("light_kernel") else "") +
                        f", {stats['col_slices']} slices, {stats['num_blocks']} blocks",
                "image_mb": round(stats["stream_bytes"] / 1e6, 1), "local_step_us": round(local_best * 1e6, 2), "kernel_us": round(kernel_ms * 1e3, 2)}
        plans = [None] * world
        dist.all_gather_object(plans, plan)
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
        elif scaling == "weak":      # per-GPU work is fixed: the N = 1 point of this workload IS one slab on one GPU -- rank 0's, on its own clock
            one_gpu = {"n_gpus": 1, "ms_per_step": round(local_best * 1e3, 5), "value": round(8.0 * nnz / local_best / 1e9, 2), "unit": "GB/s",
                       "what": "rank 0's slab, K steps as one batch between two local synchronisations inside this run"}
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
                "ms_per_step": round(per_step * 1e3, 5), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
                "backend": backend, "gather": args.gather,
                "dtype": "u32 (Q8.24 fixed point, u64 row sums)" if impl == host.IMPL_FIXED else "f32",
                "data": "synthetic" if not args.npz else "file",
                "config": {"workload": f"{name}, {B.IMPL_NAMES[impl]} IMPL, v={packets.vb_bank} o={packets.ob_bank}", "rows": full_rows or true_rows * n_gpus,
                           "cols": packets.num_cols, "nnz_per_gpu": int(nnz), "nnz_total": int(total_nnz), "slab_rows_rank0": true_rows,
                           "parallelism": f"row-slab x{n_gpus}, balanced by non-zeros" + gather_text},
                "gops": round(2.0 * total_nnz / per_step / 1e9, 2),
                "frac_whole_step": round(value / (B.HBM_PEAK_GBS * n_gpus), 4),
                "compute_only": {"ms_per_step": round(compute_elapsed / args.steps * 1e3, 5), "value": round(8.0 * total_nnz / (compute_elapsed / args.steps) / 1e9, 2),
                                 "frac_whole_step": round(8.0 * total_nnz / (compute_elapsed / args.steps) / 1e9 / (B.HBM_PEAK_GBS * n_gpus), 4)},
                "same_workload_on_one_gpu": one_gpu,
                "exchange": {"bytes_per_rank_per_gather": int(chunk) * 4, "ms_per_step_added": round((elapsed - compute_elapsed) / args.steps * 1e3, 5)},
                "exchange_every_step": None if step_elapsed is None else {"ms_per_step": round(step_elapsed / args.steps * 1e3, 5),
                                                                          "value": round(8.0 * total_nnz / (step_elapsed / args.steps) / 1e9, 2)},
                "exchange_push": push,
                # one entry per rank: [rows, nnz, plan, local step us (own clock, no barrier), kernel us]; beside it what ONE GPU predicted for the same slabs
                "per_rank": [[p["rows"], p["nnz"], p["plan"], p["local_step_us"], p["kernel_us"]] for p in plans],
                "slowest_rank_local_step_us": max(p["local_step_us"] for p in plans),
                "one_gpu_prediction": (lambda q: None if q is None else {"source": q["source"], "unsplit_us": q["unsplit_us"], "max_slab_us": q["max_slab_us"],
                                                                         "slab_us": [sl["us"] for sl in q["slabs"]],
                                                                         "predicted_compute_only_efficiency": q["predicted_compute_only_efficiency"]})(
                    one_gpu_prediction(name, B.IMPL_NAMES[impl], n_gpus)) if scaling == "strong" else None,
                "roofline": {"bound": "hbm", "kernel": B.kernel_name(stats), "achieved": round(achieved, 2),
                             "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / B.HBM_PEAK_GBS, 4), "kernel_ms": round(kernel_ms, 5),
                             "kernel_ms_from": "hs_time_kernel on rank 0's slab: one HIP event pair around K back-to-back launches of the kernel alone, / K",
                             "algorithmic_bytes_per_launch": int(8 * nnz), "streamed_bytes_per_launch": int(stats["stream_bytes"]), "traffic": None},
                "cpu_baseline": None if args.no_cpu_baseline else {"value": round(8.0 * nnz / t_cpu / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                                                                   "sample": f"rank 0's slab, 1 SpMV through oracle/cpu_ref.c (1 thread), {t_cpu*1e3:.1f} ms"},
                "parity_vs_oracle": parity + " (every rank's slab; gathered layout checked on every rank)",
            }
        eng.close()
        dist.barrier()
        extra = dict(preprocess_s={"format_csr2cpsr": round(t_fmt, 3), "device_load_retile": round(stats["load_seconds"], 3)}, slab_rows=rows_all, per_rank_plans=plans,
                     one_gpu_prediction_full=one_gpu_prediction(name, B.IMPL_NAMES[impl], n_gpus) if scaling == "strong" else None)
        return out, extra

    # ---- what this run measures ------------------------------------------------------------------------------------------------------------
    # A matrix named on the command line (--config / --scale-matrix) or an explicit --scaling: that ONE workload (strong unless told otherwise).
    # Nothing named -- the driver's `bench.py --gpus N --steps K --warmup W` -- : the N = 1 line's workload carried to N GPUs the way the path
    # shards, WEAK: every rank owns an ogbl-ppa-sized row slab (own seed) of a matrix N slabs tall, all of x, no collective inside the SpMV, one
    # final RCCL all-gather of the y slabs -- per-GPU work fixed as N grows, so value(N) / (N x value(1)) of the driver's own series reads as
    # scaling efficiency -- and, in the same line, BASELINE.json configs[4]: mouse_gene, ONE matrix split N ways (strong), with the one-GPU
    # prediction of its slabs beside the measured ones.
    named = getattr(args, "scale_matrix", None) or args.config
    if named or args.scaling:
        out, extra = measure(named or "mouse_gene", args.scaling or "strong")
        second = None
    else:
        out, extra = measure("ogbl_ppa", "weak")
        second = measure("mouse_gene", "strong", brief=True)
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
        details = {"distributed": dict(out, **extra)}
        if second is not None:
            o2, e2 = second
            details["distributed_baseline_config_4"] = dict(o2, **e2)
            out["baseline_config_4"] = {k: o2[k] for k in ("value", "unit", "ms_per_step", "scaling", "gather", "frac_whole_step", "compute_only", "same_workload_on_one_gpu",
                                                          "per_rank", "slowest_rank_local_step_us", "one_gpu_prediction", "parity_vs_oracle")}
            out["baseline_config_4"]["workload"] = o2["config"]["workload"] + f", ONE matrix split {n_gpus} ways"
            out["baseline_config_4"]["per_rank"] = [[p[1], p[3]] for p in o2["per_rank"]]      # [non-zeros, local step us] per rank; plans: the details file
        try:      # a 1-GPU run of this checkout left its predictions in the details file: the N-rank record must not erase what it is read against
            with open(B.DETAILS_FILE) as f:
                old = json.load(f)
            details.update({k: old[k] for k in ("strong_scaling_prediction",) if k in old})
        except (OSError, ValueError):
            pass
        B.emit(out, details, [])
