"""bench.py --gpus N must measure N ranks or say why it cannot: without a launcher's WORLD_SIZE it starts the
ranks itself; with fewer GPUs than N (here: none) it prints one JSON error line and exits with code 2 instead of measuring one GPU under
an N-GPU label.  The launcher path itself runs here at world_size 2 over gloo on host memory, with libhisparse_cpu.so -- the separate
host-thread build of the C-ABI -- as the engine; every rank's slab is checked against the oracle inside bench.py before timing."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "hisparse_amd", "lib", "libhisparse_cpu.so")


def _bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stdout + p.stderr


def test_more_gpus_than_the_machine_has_is_an_error_not_a_one_gpu_line():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this machine has two GPUs")
    rc, line, text = _bench(["--gpus", "2", "--quick", "--steps", "2", "--warmup", "1"])
    assert rc == 2, text
    assert line is not None and "error" in line and line.get("n_gpus_requested") == 2
    assert "n_gpus" not in line and "value" not in line


def test_launcher_world_size_must_match_gpus():
    rc, line, text = _bench(["--gpus", "4", "--quick"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2, text
    assert line is not None and "WORLD_SIZE=2" in line["error"]


def test_self_launch_two_ranks_over_gloo():
    rc, line, text = _bench(["--gpus", "2", "--backend", "gloo", "--config", "ppa_small", "--steps", "3", "--warmup", "1"], {"HISPARSE_HIP_LIB": CPU_LIB})
    assert rc == 0, text
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["parity_vs_oracle"].startswith("bit-exact")
    assert "NOT a measurement" in line["backend"]
    assert line["config"]["nnz_total"] > line["config"]["nnz_per_gpu"] > 0
    assert line["same_workload_on_one_gpu"]["n_gpus"] == 1
    # round 6: one entry per rank -- rows, non-zeros, plan, the slab on its own clock, kernel -- to read a real run against the prediction
    assert len(line["per_rank"]) == 2 and sum(p[1] for p in line["per_rank"]) == line["config"]["nnz_total"]
    assert all(p[3] > 0 and p[4] > 0 and isinstance(p[2], str) for p in line["per_rank"])
    assert line["slowest_rank_local_step_us"] == max(p[3] for p in line["per_rank"])
    assert line["one_gpu_prediction"] is None          # nobody predicted ppa_small


def test_scale_matrix_names_the_matrix_to_shard():
    rc, line, text = _bench(["--gpus", "2", "--backend", "gloo", "--scale-matrix", "gplus", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"HISPARSE_HIP_LIB": CPU_LIB}, timeout=900)
    assert rc == 0, text
    assert line["config"]["workload"].startswith("gplus, fixed IMPL") and line["config"]["nnz_total"] > 1.3e7


def test_default_n_rank_run_is_the_weak_series_with_config_4_beside_it():
    """nothing named: `value` = ogbl-ppa-sized slab per rank (weak: the N = 1 line's workload), BASELINE.json configs[4] -- mouse_gene split N ways -- in the same line"""
    rc, line, text = _bench(["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], {"HISPARSE_HIP_LIB": CPU_LIB}, timeout=1200)
    assert rc == 0, text
    assert line["scaling"] == "weak" and line["config"]["workload"].startswith("ogbl_ppa, fixed IMPL") and line["config"]["nnz_total"] > 8.4e7
    assert line["same_workload_on_one_gpu"]["n_gpus"] == 1 and line["parity_vs_oracle"].startswith("bit-exact")
    b4 = line["baseline_config_4"]
    assert b4["scaling"] == "strong" and b4["workload"].startswith("mouse_gene, fixed IMPL") and b4["parity_vs_oracle"].startswith("bit-exact")
    assert len(b4["per_rank"]) == 2 and b4["one_gpu_prediction"]["max_slab_us"] > 0 and b4["same_workload_on_one_gpu"]["n_gpus"] == 1
    last = [l for l in text.splitlines() if l.startswith("{")][-1]
    assert len(last) < 3400          # (N = 8 adds ~100 bytes per further rank: still inside bench.LINE_LIMIT)


def test_gloo_backend_refuses_the_hip_library():
    rc, line, text = _bench(["--gpus", "2", "--backend", "gloo", "--config", "ppa_small", "--steps", "1", "--warmup", "0"])
    assert rc == 2, text
    assert line is not None and "error" in line


# ---- the driver keeps the last 8 000 characters of stdout and of stderr: the final line must fit, whole ----------
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "cpu_baseline", "parity_vs_oracle")


def _tail_holds_the_line(stdout, stderr, line_text):
    assert len(line_text) < 6000, len(line_text)
    assert stdout.rstrip("\n").endswith(line_text)                 # the JSON object is the LAST line of stdout ...
    assert line_text in stdout[-8000:]                              # ... whole inside the driver's stdout tail ...
    assert line_text in (stdout[-8000:] + "\n---- stderr ----\n" + stderr[-8000:])[-16100:]


def test_distributed_line_is_compact_and_complete():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HISPARSE_HIP_LIB"] = CPU_LIB
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "ppa_small", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout + p.stderr
    last = [l for l in p.stdout.splitlines() if l.strip()][-1]
    line = json.loads(last)
    _tail_holds_the_line(p.stdout, p.stderr, last)
    for key in REQUIRED:
        assert key in line, key
    assert line["gather"] == "final"
    for key in ("bound", "kernel", "achieved", "peak", "frac", "kernel_ms", "traffic", "algorithmic_bytes_per_launch"):
        assert key in line["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in line["cpu_baseline"], key


def test_emit_never_prints_a_line_the_tail_cannot_hold(capsys, tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, "DETAILS_FILE", str(tmp_path / "bench_details.json"))
    out = {k: 1 for k in REQUIRED}
    out["roofline"] = {"bound": "hbm", "frac": 0.8}
    out["notes"] = "x" * 20000                      # an optional key that would blow the line up
    out["more"] = {"a": list(range(500))}
    bench.emit(out, {"everything": ["y" * 1000] * 40}, [bench.SUMMARY_HEAD])
    cap = capsys.readouterr()
    last = cap.out.strip().splitlines()[-1]
    line = json.loads(last)
    assert len(last) <= bench.LINE_LIMIT
    assert "notes" not in line and all(k in line for k in REQUIRED)
    assert line["details"] == "bench_details.json"
    assert json.load(open(tmp_path / "bench_details.json"))["everything"][0].startswith("y")


@__import__("pytest").mark.gpu
def test_default_single_gpu_run_fits_the_drivers_tail():
    """the driver's own command line; the whole default run (every configuration, the sweeps) with its stderr rows"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    last = [l for l in p.stdout.splitlines() if l.strip()][-1]
    line = json.loads(last)
    _tail_holds_the_line(p.stdout, p.stderr, last)
    for key in REQUIRED:
        assert key in line, key
    assert line["n_gpus"] == 1 and line["parity_vs_oracle"] == "bit-exact" and "ogbl_ppa" in line["config"]["workload"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and 0.3 < r["frac"] <= 1.0 and r["frac_whole_step"] <= r["frac"] * 1.02
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 0.01 * r["achieved"]
    # round 6: `frac` is the all-launch average (what rocprofv3 --stats prints), the steady state rides beside it, and the line says how many launches a step is
    assert r["frac"] <= r["frac_steady"] <= 1.0 and r["kernel_ms"] >= r["kernel_ms_steady"] and r["launches_per_step"] == (2 if line["config"]["col_slices"] > 1 else 1)
    live = r.get("rocprofv3_live")
    if live:      # `frac` never reads better than the live rocprofv3 --stats average; the HIP-event period of back-to-back launches is within 8 % of it
        assert r["kernel_ms"] * 1e3 >= live["kernel_avg_us"] * 0.999 and r["frac"] <= live["frac"] + 1e-4, (live, r["kernel_ms"])
        assert abs(live["kernel_avg_us"] - r.get("kernel_ms_hip_events", r["kernel_ms"]) * 1e3) < 0.08 * live["kernel_avg_us"], (live, r)
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] == 1
    # one row per matrix of the sweep inside the stderr tail, and the details next to the script
    tail = p.stderr[-8000:]
    for name in ("transformer_95/fixed", "pokec/fixed", "ogbn_products/float_stall", "mouse_gene/fixed", "ogbl_ppa/fixed"):
        assert name in tail, name
    det = json.load(open(os.path.join(ROOT, "bench_details.json")))
    assert len(det["bm_list"]) == 12 and len(det["per_config"]) == 4 and det["strong_scaling_prediction"]


@__import__("pytest").mark.gpu
@__import__("pytest").mark.parametrize("n", [2, 8])
def test_n_rank_dry_run_on_one_gpu(n):
    """main_distributed with N processes sharing GPU 0 -- set_device, the HIP engine, stream binding,
    hs_bind_device_result, slab parity against the oracle, the gathered layout, the peer-store gather over IPC handles, the JSON line --
    everything of `bench.py --gpus N` except RCCL (gloo, host-staged)."""
    args = ["--gpus", str(n), "--backend", "gloo", "--share-gpu", "--config", "mouse_gene", "--steps", "20", "--warmup", "5"]
    rc, line, text = _bench(args, timeout=1200)
    if rc != 0:      # seen ONCE, on a fresh box with the page cache cold (8 processes starting the HIP runtime at once), never reproduced in
        # 30 further runs: keep the evidence, try once more; a second failure fails the test
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, f"dry_run_{n}_first_failure.txt"), "w") as f:
                f.write(text)
        rc, line, text = _bench(args, timeout=1200)
    assert rc == 0, "\n".join(l for l in text.splitlines() if "rank" in l.lower() or "error" in l.lower() or "Traceback" in l or l.startswith("  File"))[-6000:]
    assert line["n_gpus"] == n and "DRY RUN" in line["backend"] and line["gather"] == "final"
    assert line["parity_vs_oracle"].startswith("bit-exact")
    assert line["config"]["nnz_total"] == 28967291 or line["config"]["nnz_total"] > 2.8e7
    assert line["same_workload_on_one_gpu"]["n_gpus"] == 1
    assert line["compute_only"]["ms_per_step"] > 0 and line["exchange_every_step"]["ms_per_step"] > 0
    assert line["roofline"]["kernel_ms"] > 0
    assert len(line["per_rank"]) == n and sum(p[1] for p in line["per_rank"]) == line["config"]["nnz_total"]
    pred = line["one_gpu_prediction"]                  # the committed one-GPU prediction for mouse_gene split n ways rides beside the measured slabs
    assert pred is not None and len(pred["slab_us"]) == n and pred["max_slab_us"] == max(pred["slab_us"])
    push = line["exchange_push"]
    assert push is not None and ("error" in push or push["equals_collective_on_every_rank"]), push
    last = [l for l in text.splitlines() if l.startswith("{")][-1]
    assert len(last) < 6000


@__import__("pytest").mark.gpu
def test_default_series_dry_run_on_one_gpu():
    """the driver's own N-rank command line (nothing named) as a dry run at N = 2: ogbl-ppa-sized slab per rank (weak) as `value`, mouse_gene split
    2 ways (BASELINE.json configs[4]) beside it -- HIP engine, stream binding, result binding, slab parity, gathered layout; only RCCL is left out"""
    rc, line, text = _bench(["--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "20", "--warmup", "5"], timeout=1500)
    assert rc == 0, "\n".join(l for l in text.splitlines() if "rank" in l.lower() or "error" in l.lower() or "Traceback" in l or l.startswith("  File"))[-6000:]
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and "DRY RUN" in line["backend"] and line["parity_vs_oracle"].startswith("bit-exact")
    assert line["config"]["workload"].startswith("ogbl_ppa") and line["config"]["nnz_total"] > 8.4e7
    assert 40 < line["same_workload_on_one_gpu"]["ms_per_step"] * 1e3 < 400          # one ogbl-ppa-sized slab on the one GPU it shares with the other rank
    b4 = line["baseline_config_4"]
    assert b4["scaling"] == "strong" and b4["parity_vs_oracle"].startswith("bit-exact") and len(b4["per_rank"]) == 2
    last = [l for l in text.splitlines() if l.startswith("{")][-1]
    assert len(last) < 4000
