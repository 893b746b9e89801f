"""oracle/oracle.py — ctypes binding of oracle/liboracle.so (oracle/cpu_ref.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "liboracle.so")
_lib = None

ERRORS = {-1: "bad argument", -2: "decoded row outside the partition's output bank", -3: "out of memory",
          -4: "column index outside the vector bank"}


class OracleError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-C", _DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
        l = C.CDLL(_LIB_PATH)
        u32, u32p, f32p, vpp = C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_void_p)
        l.oracle_q_from_float.argtypes = [C.c_float]
        l.oracle_q_from_float.restype = u32
        l.oracle_q_to_float.argtypes = [u32]
        l.oracle_q_to_float.restype = C.c_float
        l.oracle_q_mul.argtypes = [u32, u32]
        l.oracle_q_mul.restype = u32
        l.oracle_q_add.argtypes = [u32, u32]
        l.oracle_q_add.restype = u32
        l.oracle_top_wrapper.argtypes = [C.c_int, vpp, u32p, u32p, u32, u32, u32, u32, u32, u32, u32]
        l.oracle_spmv.argtypes = [C.c_int, vpp, u32p, u32p, u32, u32, u32, u32, u32, u32]
        l.oracle_spmv_per_channel_threads.argtypes = [C.c_int, vpp, u32p, u32p, u32, u32, u32, u32, u32, u32, C.c_int]
        l.oracle_compute_ref.argtypes = [u32, u32p, u32p, f32p, f32p, f32p]
        l.oracle_compute_ref.restype = None
        l.oracle_compute_ref_parallel.argtypes = [u32, u32p, u32p, f32p, f32p, f32p, C.c_int]
        l.oracle_compute_ref_parallel.restype = None
        l.oracle_spmspv.argtypes = [C.c_int, u32p, u32p, u32p, u32, u32, u32p, u32p, u32, u32p]
        l.oracle_verify.argtypes = [f32p, f32p, C.c_uint64]
        l.oracle_verify.restype = C.c_int64
        _lib = l
    return _lib


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _channel_ptrs(channels):
    """channels: 16 numpy (n,16) uint32 arrays, or 16 integer addresses."""
    arr = (C.c_void_p * 16)()
    keep = []
    for i, ch in enumerate(channels):
        if isinstance(ch, (int, np.integer)):
            arr[i] = int(ch)
        else:
            ch = np.ascontiguousarray(ch, dtype=np.uint32)
            keep.append(ch)
            arr[i] = ch.ctypes.data
    return arr, keep


def q_from_float(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    l = lib()
    return np.array([l.oracle_q_from_float(float(v)) for v in x.ravel()], dtype=np.uint32).reshape(x.shape)


def q_to_float(w):
    w = np.ascontiguousarray(w, dtype=np.uint32)
    return (w.astype(np.float64) / 16777216.0).astype(np.float32)


def pack_vector(impl, x):
    """float -> value words, independent of the product's hsf_pack_vector."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if impl != 0:
        return x.view(np.uint32).copy()
    d = x.astype(np.float64)
    s = np.floor(d * 16777216.0 + 0.5)
    s = np.where(d > 0.0, s, 0.0)
    return np.minimum(s, 4294967295.0).astype(np.uint32)


def unpack_result(impl, words):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    return words.view(np.float32).copy() if impl != 0 else q_to_float(words)


def spmv(impl, channels, x_words, num_rows, num_cols, num_row_partitions, num_col_partitions, ob_bank, vb_bank):
    """All row partitions through oracle_top_wrapper; returns packed y words (num_rows,)."""
    x_words = np.ascontiguousarray(x_words, dtype=np.uint32)
    if x_words.size != num_cols:
        raise OracleError("x must have num_cols (padded) words")
    y = np.zeros(num_rows, dtype=np.uint32)  # host zero-initialises y (sw/benchmark.cpp:217-222)
    ptrs, keep = _channel_ptrs(channels)
    rc = lib().oracle_spmv(impl, ptrs, _u32p(x_words), _u32p(y), num_rows, num_cols, num_row_partitions,
                           num_col_partitions, ob_bank, vb_bank)
    del keep
    if rc != 0:
        raise OracleError(f"oracle_spmv failed: {ERRORS.get(rc, rc)}")
    return y


def spmv_per_channel_threads(impl, channels, x_words, num_rows, num_cols, num_row_partitions, num_col_partitions, ob_bank, vb_bank,
                             threads=16):
    """oracle_spmv with one host thread per cluster (same result; CPU-baseline context only)."""
    x_words = np.ascontiguousarray(x_words, dtype=np.uint32)
    if x_words.size != num_cols:
        raise OracleError("x must have num_cols (padded) words")
    y = np.zeros(num_rows, dtype=np.uint32)
    ptrs, keep = _channel_ptrs(channels)
    rc = lib().oracle_spmv_per_channel_threads(impl, ptrs, _u32p(x_words), _u32p(y), num_rows, num_cols, num_row_partitions,
                                               num_col_partitions, ob_bank, vb_bank, threads)
    del keep
    if rc != 0:
        raise OracleError(f"oracle_spmv failed: {ERRORS.get(rc, rc)}")
    return y


def top_wrapper(impl, channels, x_words, y_words, row_part_id, part_len, num_col_partitions, num_partitions, num_cols,
                ob_bank, vb_bank):
    """One row partition, in place on y_words — csim's top_wrapper (spmv_csim/csim.cpp:22-46)."""
    ptrs, keep = _channel_ptrs(channels)
    rc = lib().oracle_top_wrapper(impl, ptrs, _u32p(x_words), _u32p(y_words), row_part_id, part_len,
                                  num_col_partitions, num_partitions, num_cols, ob_bank, vb_bank)
    del keep
    if rc != 0:
        raise OracleError(f"oracle_top_wrapper failed: {ERRORS.get(rc, rc)}")


def compute_ref(num_rows, indptr, indices, data, x):
    """float32 CSR loop of spmv_csim/csim.cpp:143-158."""
    indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
    indices = np.ascontiguousarray(indices, dtype=np.uint32)
    data = np.ascontiguousarray(data, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros(num_rows, dtype=np.float32)
    lib().oracle_compute_ref(num_rows, _u32p(indptr), _u32p(indices), _f32p(data), _f32p(x), _f32p(y))
    return y


def usable_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def compute_ref_parallel(num_rows, indptr, indices, data, x, out=None, threads=None):
    """OpenMP version of compute_ref over the usable host cores (CPU-baseline context only)."""
    y = np.zeros(num_rows, dtype=np.float32) if out is None else out
    lib().oracle_compute_ref_parallel(num_rows, _u32p(indptr), _u32p(indices), _f32p(data), _f32p(x), _f32p(y), threads or usable_cores())
    return y


def spmspv(impl, indptr, row_indices, value_words, num_rows, num_cols, x_index, x_words):
    """SpMSpV over a CSC matrix (extension): packed y words (num_rows,)."""
    indptr, row_indices, value_words, x_index, x_words = (np.ascontiguousarray(a, dtype=np.uint32) for a in (indptr, row_indices, value_words, x_index, x_words))
    y = np.zeros(num_rows, dtype=np.uint32)
    rc = lib().oracle_spmspv(impl, _u32p(indptr), _u32p(row_indices), _u32p(value_words), num_rows, num_cols, _u32p(x_index), _u32p(x_words),
                             x_index.size, _u32p(y))
    if rc != 0:
        raise OracleError(f"oracle_spmspv failed: {ERRORS.get(rc, rc)}")
    return y


def verify(reference, kernel):
    """csim's verify (:160-184): index of the first |k - r| >= 1e-4, or -1 when all match."""
    reference = np.ascontiguousarray(reference, dtype=np.float32)
    kernel = np.ascontiguousarray(kernel, dtype=np.float32)
    if reference.size != kernel.size:
        return 0
    return int(lib().oracle_verify(_f32p(reference), _f32p(kernel), reference.size))
