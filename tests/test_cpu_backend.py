"""libhisparse_cpu.so: the drop-in C-ABI on host threads for machines without a GPU ("a CPU backend with the
same symbols for config (1)").  A separate library, loaded here in a child process through HISPARSE_HIP_LIB -- the default library
never falls back to it (tests/test_capi.py: without a GPU hs_create of libhisparse_hip.so fails).  Checked against the oracle:
bit-exact fixed point, 1e-4 float; BASELINE config 1 (1k x 1k, 1 %), multi-partition banks, the partition loop, the CSR entry point."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "hisparse_amd", "lib", "libhisparse_cpu.so")

CHILD = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
from hisparse_amd import device, host
from oracle import oracle as orc
import cases

def oracle_y(cp, impl, xw):
    return orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)

checked = 0
for impl in (0, 1, 2):
    for rows, cols, density, vb, ob, skip in [(1000, 1000, 0.01, 0, 0, True), (2500, 300, 0.03, 4, 8 if impl == 2 else 1, False), (300, 5000, 0.02, 16, 8, True)]:
        m = cases.random_csr(rows, cols, density, 7 + rows, impl)
        csr = host.CSRMatrix.from_scipy(m)
        kw = dict(vb_bank=vb, ob_bank=ob) if vb else {}
        cp = host.format_matrix(csr, impl, skip_empty_rows=skip, **kw)
        xw = host.pack_vector(impl, cases.random_x(cp.num_cols, rows, impl))
        want = oracle_y(cp, impl, xw)
        same = (lambda a: np.array_equal(a, want)) if impl == 0 else (lambda a: cases.float_close(a, want))
        with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
            eng.load_matrix(cp)
            assert eng.stats()["nnz"] == m.nnz
            eng.load_vector(xw)
            eng.run()
            assert same(eng.read_result())
            eng.load_vector(np.zeros_like(xw)); eng.run(); assert not eng.read_result().any()
            eng.load_vector(xw)
            for j in range(cp.num_row_partitions):                      # the reference's launch loop (sw/benchmark.cpp:318-338)
                eng.run_partition(j, cp.part_len(j))
            assert same(eng.read_result())
            try:
                eng.run_partition(0, cp.part_len(0) + 1)
                raise SystemExit("a wrong part_len was accepted")
            except device.DeviceError:
                pass
        with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:   # straight from CSR
            eng.load_matrix_csr(csr)
            eng.load_vector(xw)
            eng.run()
            assert same(eng.read_result())
            try:
                eng.feedback(0, 0)
                raise SystemExit("an extension answered on the CPU backend")
            except device.DeviceError as e:
                assert e.code == -6
        checked += 1
# saturation and rounding (SURVEY 8c: unpinned by the reference, pinned here against the oracle's restatement)
ip = np.array([0, 3] + [3] * 127, dtype=np.uint32)
csr = host.CSRMatrix.from_arrays(128, 8, ip, np.array([0, 1, 2], dtype=np.uint32), np.array([200.0, 100.0, 0.3333333], dtype=np.float32))
cp = host.format_matrix(csr, 0, skip_empty_rows=True)
xw = host.pack_vector(0, np.array([1.5, 1.0, 0.7, 0, 0, 0, 0, 0], dtype=np.float32))
with device.SpmvEngine(0) as eng:
    eng.load_matrix(cp); eng.load_vector(xw); eng.run()
    y = eng.read_result()
assert y[0] == 0xffffffff and np.array_equal(y, oracle_y(cp, 0, xw))
print("cpu backend ok", checked)
"""


def test_cpu_backend_is_a_separate_library_and_matches_the_oracle():
    if not os.path.exists(CPU_LIB):
        subprocess.check_call(["make", "-C", ROOT, "cpu"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, HISPARSE_HIP_LIB=CPU_LIB)
    env.pop("HISPARSE_STREAM_FORMAT", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "cpu backend ok 9" in r.stdout, r.stdout + r.stderr


def test_default_library_has_no_cpu_path():
    """libhisparse_hip.so does not link, load or name the CPU library."""
    hip = os.path.join(ROOT, "hisparse_amd", "lib", "libhisparse_hip.so")
    needed = subprocess.run(["readelf", "-d", hip], capture_output=True, text=True).stdout
    assert "hisparse_cpu" not in needed
    with open(hip, "rb") as f:
        assert b"libhisparse_cpu" not in f.read()
