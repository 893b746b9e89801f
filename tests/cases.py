"""Shared seeded test matrices / helpers for the CPU and GPU suites."""
import numpy as np
import scipy.sparse as sp

from hisparse_amd import host


def random_csr(rows, cols, density, seed, impl, low_rows=None):
    """Seeded scipy CSR with values suited to the numeric mode (non-negative for fixed point)."""
    rng = np.random.default_rng(seed)
    m = sp.random(rows, cols, density=density, random_state=np.random.RandomState(seed), format="csr", dtype=np.float32)
    if impl == 0:
        m.data = rng.uniform(0.0, 2.0, m.nnz).astype(np.float32)
    else:
        m.data = rng.normal(0.0, 1.0, m.nnz).astype(np.float32)
    m.sort_indices()
    return m


def random_x(n, seed, impl):
    rng = np.random.default_rng(seed + 1000)
    if impl == 0:
        return rng.uniform(0.0, 3.0, n).astype(np.float32)
    return rng.normal(0.0, 1.0, n).astype(np.float32)


def formatted(m, impl, vb_bank, ob_bank, skip_empty_rows):
    """(csr handle, ChannelPackets) through the product's host library."""
    csr = host.CSRMatrix.from_scipy(m)
    cp = host.format_matrix(csr, impl, vb_bank=vb_bank, ob_bank=ob_bank, skip_empty_rows=skip_empty_rows)
    return csr, cp


def float_close(y_words, ref_words, rtol=1e-4, atol=1e-4):
    """float-mode parity: |y - ref| <= atol + rtol * |ref| (north_star: 1e-4 relative; csim verify: 1e-4 absolute)."""
    a, b = y_words.view(np.float32), ref_words.view(np.float32)
    return np.allclose(a, b, rtol=rtol, atol=atol)
