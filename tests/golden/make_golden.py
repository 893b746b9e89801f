"""Generate tests/golden/*.npz — small input/output vectors for the SpMV path.

Run from the repo root:  python tests/golden/make_golden.py
What the fixtures hold (data only): a CSR matrix, x as float32, the numeric mode and bank sizes, the SHA-256 of every
channel buffer the host pipeline must produce, and the packed y words.  Provenance of the expected values:
  * csim_*: the reference's own synthetic cases (spmv_csim/csim.cpp:443-466: all-ones matrices, x = glibc rand() % 2
    with the default seed, expected y = compute_ref's integer row sums) — known answers of the reference itself;
  * kat_*: seeded random cases whose expected channel hashes come from oracle/cpsr_format.py (line-by-line restatement
    of sw/data_formatter.h + sw/benchmark.cpp:127-195) and whose y comes from oracle/cpu_ref.c (restatement of
    csim's top_wrapper).  The reference cannot be executed in this image (Vitis HLS headers + cnpy are absent), so
    these pin the oracle and the product to each other and to this commit, not to a run of the reference.
"""
import ctypes
import hashlib
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cpsr_format as of  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def glibc_x(n, skip=0):
    libc = ctypes.CDLL("libc.so.6")
    libc.srand(1)
    for _ in range(skip):
        libc.rand()
    return np.array([libc.rand() % 2 for _ in range(n)], dtype=np.float32)


def emit(name, impl, m, x, vb, ob, skip):
    m = m.tocsr()
    m.sort_indices()
    chans, rows, cols, rp, cp = of.format_matrix(impl, m.shape[0], m.shape[1], m.data.astype(np.float32), m.indices, m.indptr, ob, vb, skip)
    xp = np.zeros(cols, dtype=np.float32)
    xp[:x.size] = x
    xw = orc.pack_vector(impl, xp)
    y = orc.spmv(impl, chans, xw, rows, cols, rp, cp, ob, vb)
    sha = np.array([hashlib.sha256(np.ascontiguousarray(c).tobytes()).hexdigest() for c in chans])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), impl=impl, vb_bank=vb, ob_bank=ob, skip_empty_rows=int(skip),
                        shape=np.array(m.shape), indptr=m.indptr.astype(np.uint32), indices=m.indices.astype(np.uint32),
                        data=m.data.astype(np.float32), x=xp, padded=np.array([rows, cols, rp, cp]), channel_sha256=sha,
                        channel_packets=np.array([c.shape[0] for c in chans]), y_words=y)
    print(name, m.shape, "nnz", m.nnz, "partitions", rp, cp)


def uniform(rows, cols, per_row):   # spmv_csim/csim.cpp:411-435
    step = cols // per_row
    idx = np.array([[(step * j + i) % cols for j in range(per_row)] for i in range(rows)])
    return sp.csr_matrix((np.ones(rows * per_row, dtype=np.float32), idx.ravel(), np.arange(0, rows * per_row + 1, per_row)), shape=(rows, cols))


if __name__ == "__main__":
    rng = np.random.default_rng(2026)
    # the reference's own known answers (default banks of each mode)
    emit("csim_basic_dense_fixed", 0, sp.csr_matrix(np.ones((128, 128), dtype=np.float32)), glibc_x(128), 4096, 8192, False)
    emit("csim_basic_sparse_fixed", 0, uniform(1000, 1024, 10), glibc_x(1024, 128), 4096, 8192, False)
    emit("csim_basic_sparse_float_pob", 1, uniform(1000, 1024, 10), glibc_x(1024, 128), 4096, 1024, False)
    emit("csim_basic_sparse_float_stall", 2, uniform(1000, 1024, 10), glibc_x(1024, 128), 4096, 8192, False)
    # seeded cases with small banks: headers, start offsets, last-partition part_len, interleave, skip counts > 1
    def rand(rows, cols, dens, signed):
        m = sp.random(rows, cols, density=dens, random_state=np.random.RandomState(7), format="csr", dtype=np.float32)
        m.data = (rng.normal(0, 1, m.nnz) if signed else rng.uniform(0, 2, m.nnz)).astype(np.float32)
        return m
    emit("kat_fixed_multi_partition", 0, rand(300, 50, 0.06, False), rng.uniform(0, 3, 50).astype(np.float32), 1, 1, True)
    emit("kat_float_pob_multi_partition", 1, rand(300, 50, 0.06, True), rng.normal(0, 1, 50).astype(np.float32), 1, 1, True)
    emit("kat_float_stall_interleave", 2, rand(2100, 50, 0.03, True), rng.normal(0, 1, 50).astype(np.float32), 1, 8, True)
    dense = np.zeros((256, 512), dtype=np.float32)
    dense[0, :] = 200.0                       # every product saturates, so does the row
    dense[1, :300] = 1.0                      # the running sum saturates
    dense[2, ::7] = 2.0 ** -22                # products far below one LSB: AP_RND decides between 0 and 1 LSB
    dense[5, 3] = 0.75
    emit("kat_fixed_round_saturate", 0, sp.csr_matrix(dense), np.linspace(0.5, 3.0, 512).astype(np.float32), 4096, 8192, True)
