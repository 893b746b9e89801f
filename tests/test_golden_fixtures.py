"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py):
CPU: product host pipeline -> channel buffer hashes, oracle -> y words;  GPU: the HIP path -> y words."""
import glob
import hashlib
import os

import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(HERE, "*.npz")))
assert len(FIXTURES) >= 8


def load(path):
    f = np.load(path)
    csr = host.CSRMatrix.from_arrays(int(f["shape"][0]), int(f["shape"][1]), f["indptr"], f["indices"], f["data"])
    cp = host.format_matrix(csr, int(f["impl"]), vb_bank=int(f["vb_bank"]), ob_bank=int(f["ob_bank"]), skip_empty_rows=bool(f["skip_empty_rows"]))
    return f, cp


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_host_pipeline_and_oracle_reproduce_fixture(path):
    f, cp = load(path)
    assert [cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions] == f["padded"].tolist()
    for c in range(16):
        a = cp.channel(c)
        assert a.shape[0] == f["channel_packets"][c]
        assert hashlib.sha256(a.tobytes()).hexdigest() == str(f["channel_sha256"][c]), f"channel {c}"
    impl = int(f["impl"])
    xw = host.pack_vector(impl, f["x"])
    y = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                 cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    assert np.array_equal(y, f["y_words"])          # the oracle is deterministic, float modes included


def test_csim_fixture_is_the_reference_known_answer():
    # csim's acceptance: |y - compute_ref| < 1e-4 with integer row sums (spmv_csim/csim.cpp:143-184,443-466)
    f = np.load(os.path.join(HERE, "csim_basic_dense_fixed.npz"))
    y = orc.unpack_result(0, f["y_words"])
    assert np.array_equal(y, np.full(128, f["x"][:128].sum(), dtype=np.float32))
    f = np.load(os.path.join(HERE, "kat_fixed_round_saturate.npz"))
    assert f["y_words"][0] == 0xFFFFFFFF and f["y_words"][1] == 0xFFFFFFFF   # AP_SAT
    assert 0 < f["y_words"][2] < 2000                                           # a few LSBs survive AP_RND


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_gpu_reproduces_fixture(path):
    f, cp = load(path)
    impl = int(f["impl"])
    eng = device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank)
    eng.load_matrix(cp)
    eng.load_vector(host.pack_vector(impl, f["x"]))
    eng.run()
    y = eng.read_result()
    eng.close()
    if impl == 0:
        assert np.array_equal(y, f["y_words"])
    else:
        assert np.allclose(y.view(np.float32), f["y_words"].view(np.float32), rtol=1e-4, atol=1e-4)
