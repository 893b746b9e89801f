"""The C++ `benchmark` entry (hisparse_amd/csrc/benchmark.cpp = the reference's sw/benchmark.cpp on the HIP C-ABI), run as
a process: result line format (sw/benchmark.cpp:80-87), the reference-literal `1 / num_cols` value mode (:411), the
literal per-partition launch loop (:318-338), y dumped and checked against the oracle, and the row-sharded RCCL path."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from hisparse_amd import host
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCHMARK = os.path.join(ROOT, "hisparse_amd", "lib", "benchmark")
RESULT_LINE = re.compile(r"^\{Preprocessing: (\S+) s \| SpMV: (\S+) ms \| (\S+) GBPS \| (\S+) GOPS \}$", re.M)


def run(*args, timeout=180):
    """The driver as a process.  A run that HANGS (seen once in round 5: `--gpus 1 --sharded` sat in RCCL's communicator set-up on a fresh
    box for 300 s; the same command passed on every other box before and after) is killed, noted under gpurun_out/ and tried ONCE more;
    a second hang fails the test."""
    cmd = [BENCHMARK, *map(str, args)]
    try:
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "benchmark_cli_first_hang.txt"), "a") as f:
                f.write(" ".join(cmd) + "\n" + ((e.output or b"").decode(errors="replace") if isinstance(e.output, bytes) else str(e.output)) + "\n")
        return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)


def glibc_rand_mod2(n):
    """The reference never seeds rand() (sw/benchmark.cpp:206): glibc's default sequence (seed 1)."""
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    return np.array([libc.rand() % 2 for _ in range(n)], dtype=np.float32)


def test_usage_and_argument_errors_need_no_gpu():
    assert os.access(BENCHMARK, os.X_OK)
    r = run()
    assert r.returncode == 0 and r.stdout.startswith("Usage:")          # like the reference: usage, exit 0 (:357-361)
    r = run("double", "x.npz", 4, 8)
    assert r.returncode == 1 and "unknown implementation" in r.stdout
    r = run("fixed", "/nonexistent/file.npz", 4, 8)
    assert r.returncode == 1 and "ERROR" in r.stdout


def _matrix(tmp_path, impl, rows=3000, cols=1500, density=0.01, seed=11):
    rng = np.random.default_rng(seed)
    m = sp.random(rows, cols, density=density, random_state=np.random.RandomState(seed), format="csr", dtype=np.float32)
    m.data = (rng.uniform(0.0, 2.0, m.nnz) if impl == 0 else rng.normal(size=m.nnz)).astype(np.float32)
    m.sort_indices()
    path = tmp_path / "m.npz"
    sp.save_npz(path, m)
    return m, str(path)


def _oracle_y(m, impl, vb, ob, x_words):
    csr = host.CSRMatrix.from_scipy(m)
    cp = host.format_matrix(csr, impl, vb_bank=vb, ob_bank=ob, skip_empty_rows=True)
    assert x_words.size == cp.num_cols
    return cp, orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], x_words, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                        cp.num_col_partitions, cp.ob_bank, cp.vb_bank)


@pytest.mark.gpu
@pytest.mark.parametrize("impl_name,impl,v,o", [("fixed", 0, 4, 8), ("float_pob", 1, 4, 1), ("float_stall", 2, 4, 8)])
def test_result_line_and_dumped_y_match_oracle(tmp_path, impl_name, impl, v, o):
    m, path = _matrix(tmp_path, impl)
    xf, yf = tmp_path / "x.bin", tmp_path / "y.bin"
    r = run(impl_name, path, v, o, "--values", "keep", "--dump-x", xf, "--dump-y", yf)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "===== Benchmark Finished =====" in r.stdout
    hit = RESULT_LINE.search(r.stdout)
    assert hit, r.stdout
    pre_s, ms, gbps, gops = map(float, hit.groups())
    assert pre_s > 0 and ms > 0
    assert gbps == pytest.approx(m.nnz * 8 / 2 ** 30 / (ms / 1e3), rel=1e-3)      # GiB/s of 8 B per non-zero (:312-314,344-346)
    assert gops == pytest.approx(2 * m.nnz / 1e6 / ms, rel=1e-3)
    x = np.fromfile(xf, dtype=np.uint32)
    y = np.fromfile(yf, dtype=np.uint32)
    assert np.array_equal(orc.unpack_result(impl, x), glibc_rand_mod2(x.size))    # x = rand() % 2, unseeded (:205-212)
    cp, want = _oracle_y(m, impl, v * 1024, o * 1024, x)
    assert y.size == cp.num_rows
    if impl == 0:
        assert np.array_equal(y, want)
    else:
        assert np.allclose(y.view(np.float32), want.view(np.float32), rtol=1e-4, atol=1e-4)
    assert np.any(y != 0)


@pytest.mark.gpu
def test_reference_literal_values_and_partition_loop(tmp_path):
    # 3000 rows with o = 1K words per bank -> LOGICAL_OB 131072: one partition; use a tall matrix and tiny banks instead
    m, path = _matrix(tmp_path, 0, rows=300000, cols=512, density=0.002, seed=5)
    yf = tmp_path / "y.bin"
    r = run("fixed", path, 4, 1, "--values", "literal", "--dump-y", yf)           # `1 / num_cols` == 0 (:411): y is all zeros
    assert r.returncode == 0, r.stdout + r.stderr
    assert RESULT_LINE.search(r.stdout)
    y = np.fromfile(yf, dtype=np.uint32)
    assert y.size == 300032 and not y.any()
    # the literal launch loop: 3 row partitions of 131072 rows, one hs_run_partition + finish each; same y as one launch
    xa, ya, xb, yb = (tmp_path / n for n in ("xa", "ya", "xb", "yb"))
    ra = run("fixed", path, 4, 1, "--values", "keep", "--dump-x", xa, "--dump-y", ya)
    rb = run("fixed", path, 4, 1, "--values", "keep", "--partition-loop", "--dump-x", xb, "--dump-y", yb)
    assert ra.returncode == 0 and rb.returncode == 0, ra.stdout + rb.stdout
    assert "row_partitions: 3" in rb.stdout
    assert np.array_equal(np.fromfile(xa, dtype=np.uint32), np.fromfile(xb, dtype=np.uint32))
    assert np.array_equal(np.fromfile(ya, dtype=np.uint32), np.fromfile(yb, dtype=np.uint32))
    cp, want = _oracle_y(m, 0, 4096, 1024, np.fromfile(xa, dtype=np.uint32))
    assert np.array_equal(np.fromfile(ya, dtype=np.uint32), want)


@pytest.mark.gpu
def test_row_sharded_rccl_path_on_one_gpu(tmp_path):
    """`--gpus N` = one context per device + ncclAllGather of the y slabs.  A single-GPU box can only run N = 1, which
    still goes through the slab split, hs_bind_device_result into the gather buffer and the RCCL call."""
    m, path = _matrix(tmp_path, 0, rows=20000, cols=3000, density=0.004, seed=9)
    xf, yf = tmp_path / "x.bin", tmp_path / "y.bin"
    r = run("fixed", path, 4, 8, "--values", "keep", "--gpus", 1, "--sharded", "--dump-x", xf, "--dump-y", yf)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "compute only" in r.stdout and "all-gather" in r.stdout and RESULT_LINE.search(r.stdout)
    x = np.fromfile(xf, dtype=np.uint32)
    cp, want = _oracle_y(m, 0, 4096, 8192, x)
    y = np.fromfile(yf, dtype=np.uint32)
    assert y.size == 20000 and np.array_equal(y, want[:20000])
    # the same gather without a collective: every device stores its slab into the others' buffers (hs_push_result; with one device the
    # "peer" is a second buffer on it) -- timed beside ncclAllGather, and the pushed slab must equal its source
    r = run("fixed", path, 4, 8, "--values", "keep", "--gpus", 1, "--sharded", "--peer-gather", "--dump-y", yf)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "peer-store gather cost per SpMV" in r.stdout and "pushed slabs identical to their sources" in r.stdout
    assert np.array_equal(np.fromfile(yf, dtype=np.uint32), want[:20000])
    r = run("fixed", path, 4, 8, "--gpus", 64)
    assert r.returncode != 0 and "device(s) visible" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("impl_name,impl,v,o", [("fixed", 0, 4, 1), ("float_stall", 2, 4, 8)])
def test_from_csr_gives_the_same_y(tmp_path, impl_name, impl, v, o):
    """--from-csr (hs_load_matrix_csr: no csr2cpsr) against the default path, also through the literal per-partition loop."""
    m, path = _matrix(tmp_path, impl, rows=300000, cols=512, density=0.002, seed=6)
    xa, ya, xb, yb = (tmp_path / n for n in ("xa", "ya", "xb", "yb"))
    ra = run(impl_name, path, v, o, "--values", "keep", "--dump-x", xa, "--dump-y", ya)
    rb = run(impl_name, path, v, o, "--values", "keep", "--from-csr", "--partition-loop", "--dump-x", xb, "--dump-y", yb)
    assert ra.returncode == 0 and rb.returncode == 0, ra.stdout + rb.stdout
    assert RESULT_LINE.search(rb.stdout)
    assert np.array_equal(np.fromfile(xa, dtype=np.uint32), np.fromfile(xb, dtype=np.uint32))
    a, b = np.fromfile(ya, dtype=np.uint32), np.fromfile(yb, dtype=np.uint32)
    assert a.size == b.size and a.any()
    assert np.array_equal(a, b) if impl == 0 else np.allclose(a.view(np.float32), b.view(np.float32), rtol=1e-5, atol=1e-5)


# ---- round 5: the harness's own check in the C++ driver, and the N-slab dry run on one GPU ------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("impl_name,impl,v,o", [("fixed", 0, 4, 8), ("float_pob", 1, 4, 1), ("float_stall", 2, 4, 8)])
def test_verify_passes_and_fails_with_an_exit_code(tmp_path, impl_name, impl, v, o):
    """--verify = compute_ref + verify of sw/host.cpp:33-74 (float32 CSR loop, absolute 1e-4), own code in benchmark.cpp (not the oracle)."""
    m, path = _matrix(tmp_path, impl)
    r = run(impl_name, path, v, o, "--values", "keep", "--verify")
    assert r.returncode == 0, r.stdout + r.stderr
    assert "INFO : verify PASSED" in r.stdout and RESULT_LINE.search(r.stdout)
    r = run(impl_name, path, v, o, "--values", "keep", "--from-csr", "--verify")
    assert r.returncode == 0 and "INFO : verify PASSED" in r.stdout, r.stdout
    # an epsilon no fixed-point / float32 result can meet: the reference's failure lines, the first failing row, exit code 3
    r = run(impl_name, path, v, o, "--values", "keep", "--verify-eps", "1e-30")
    assert r.returncode == 3, r.stdout
    assert "Error: Result mismatch" in r.stdout and re.search(r"i = \d+  Reference result = \S+  Kernel result = \S+", r.stdout)
    assert "INFO : verify FAILED" in r.stdout and "Benchmark Finished" not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 8])
def test_n_slabs_sharing_one_gpu(tmp_path, n):
    """--gpus N --share-gpu: the row-slab path with N contexts on ONE device -- the split, N streams, hs_bind_device_result into the gather
    buffers and the gather by peer stores all execute with N > 1 (only RCCL is left out); y assembled from device 0's gather buffer must be
    the oracle's and pass the driver's own --verify."""
    m, path = _matrix(tmp_path, 0, rows=20000, cols=3000, density=0.004, seed=9)
    xf, yf = tmp_path / "x.bin", tmp_path / "y.bin"
    r = run("fixed", path, 4, 8, "--values", "keep", "--gpus", n, "--share-gpu", "--verify", "--dump-x", xf, "--dump-y", yf)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"({n} row slabs sharing ONE GPU" in r.stdout and r.stdout.count("  slab ") == n
    assert "compute only" in r.stdout and "pushed slabs identical to their sources" in r.stdout and "INFO : verify PASSED" in r.stdout
    x = np.fromfile(xf, dtype=np.uint32)
    cp, want = _oracle_y(m, 0, 4096, 8192, x)
    y = np.fromfile(yf, dtype=np.uint32)
    assert y.size == 20000 and np.array_equal(y, want[:20000])


# ---- round 6: the reference's own eight dataset cases of csim (spmv_csim/csim.cpp:481-591) ---------------------------------------------------
# Every one of them overwrites the values with `1 / mat_f.num_cols` -- INTEGER division, i.e. 0.0f (csim.cpp:485; sw/benchmark.cpp:411 does the
# same) -- so what csim really asserts on gplus / ogbl_ppa / pokec / hollywood / ogbn_products / mouse_gene / transformer_50_t / transformer_95_t
# is: the full-size matrix goes through the formatter and the partition-by-partition launch loop (top_wrapper, csim.cpp:22-46) and y comes out
# ALL ZERO against compute_ref.  Here: the stand-ins of the same shapes (datasets.py; the files themselves are absent), `--values literal`,
# `--partition-loop`, `--verify` (the harness's own check), y dumped and asserted to be all-zero words.  Every matrix in one numeric mode
# (cycling), mouse_gene and transformer_95 in all three.
_CSIM_CASES = ["gplus", "ogbl_ppa", "pokec", "hollywood", "ogbn_products", "mouse_gene", "transformer_50", "transformer_95"]
_MODES = [("fixed", 4, 8), ("float_pob", 4, 1), ("float_stall", 4, 8)]


def _csim_literal_params():
    out = [(name, _MODES[k % 3]) for k, name in enumerate(_CSIM_CASES)]
    for name in ("mouse_gene", "transformer_95"):
        out += [(name, m) for m in _MODES if (name, m) not in out]
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode", _csim_literal_params(), ids=lambda p: p if isinstance(p, str) else p[0])
def test_csim_dataset_cases_with_the_reference_literal_values(tmp_path, name, mode):
    from hisparse_amd import datasets
    impl_name, v, o = mode
    c = datasets.CONFIGS[name]
    spec = f"synth:{c.kind}:{c.rows}:{c.cols}:{c.a}:{c.b}:{c.c}:{c.seed}"
    yf = tmp_path / "y.bin"
    r = run(impl_name, spec, v, o, "--values", "literal", "--partition-loop", "--runs", 2, "--verify", "--dump-y", yf, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert RESULT_LINE.search(r.stdout) and "INFO : verify PASSED" in r.stdout
    assert int(re.search(r"row_partitions: (\d+)", r.stdout).group(1)) >= 1
    y = np.fromfile(yf, dtype=np.uint32)
    assert y.size >= c.rows and not y.any(), f"{int((y != 0).sum())} non-zero words in a product with an all-zero matrix"
