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


def test_gloo_backend_refuses_the_hip_library():
    rc, line, text = _bench(["--gpus", "2", "--backend", "gloo", "--config", "ppa_small", "--steps", "1", "--warmup", "0"])
    assert rc == 2, text
    assert line is not None and "error" in line
