"""hisparse_amd.host — ctypes binding of libhisparse_host.so (include/hisparse_host.h).

Python mirror of the reference's host interface for the SpMV path:
  CSRMatrix                    ~ spmv::io::CSRMatrix<float>             (sw/data_loader.h:19-30)
  load_csr_matrix_from_float_npz                                         (sw/data_loader.h:51-70)
  format_matrix(csr, impl, v, o, skip_empty_rows) -> ChannelPackets      (sw/benchmark.cpp:110-195:
        util_round_csr_matrix_dim + csr_matrix_convert_from_float + csr2cpsr + channel assembly)
  pack_vector / unpack_result                                            (sw/benchmark.cpp:207-212, csim.cpp:172)
All heavy lifting happens in the C++ library; this file only marshals numpy arrays.
"""
import ctypes as C
import os

import numpy as np

IMPL_FIXED, IMPL_FLOAT_POB, IMPL_FLOAT_STALL = 0, 1, 2
IMPL_NAMES = {"fixed": IMPL_FIXED, "float_pob": IMPL_FLOAT_POB, "float_stall": IMPL_FLOAT_STALL}
PACK_SIZE = 8
NUM_HBM_CHANNELS = 16

_LIB_PATH = os.environ.get("HISPARSE_HOST_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhisparse_host.so")      # (the override: tools/sanitize/run.sh)


class HostError(RuntimeError):
    pass


class _Info(C.Structure):
    _fields_ = [("impl", C.c_int32), ("interleave", C.c_uint32), ("ob_bank", C.c_uint32), ("vb_bank", C.c_uint32),
                ("num_rows", C.c_uint32), ("num_cols", C.c_uint32), ("num_row_partitions", C.c_uint32),
                ("num_col_partitions", C.c_uint32), ("nnz", C.c_uint64), ("streamed_bytes", C.c_uint64),
                ("skip_empty_rows", C.c_int32)]


_lib = None


def lib():
    """Load libhisparse_host.so (built by `make host` / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise HostError(f"{_LIB_PATH} is missing: run `make host` (or __graft_entry__.build()) first")
        l = C.CDLL(_LIB_PATH)
        vp, u32, u64, f32p, u32p = C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        l.hsf_last_error.restype = C.c_char_p
        l.hsf_csr_load_npz.argtypes = [C.c_char_p, C.POINTER(vp)]
        l.hsf_csr_from_arrays.argtypes = [u32, u32, u64, u32p, u32p, f32p, C.POINTER(vp)]
        l.hsf_csr_dims.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64)]
        l.hsf_csr_copy.argtypes = [vp, u32p, u32p, f32p]
        l.hsf_csr_fill.argtypes = [vp, C.c_float]
        l.hsf_csr_normalize_by_outdegree.argtypes = [vp]
        l.hsf_csr_free.argtypes = [vp]
        l.hsf_csr_free.restype = None
        l.hsf_csr_generate.argtypes = [C.c_char_p, u32, u32, C.c_double, C.c_double, C.c_double, u64, C.POINTER(vp)]
        l.hsf_format.argtypes = [vp, C.c_int, u32, u32, C.c_int, C.POINTER(vp)]
        l.hsf_matrix_get_info.argtypes = [vp, C.POINTER(_Info)]
        l.hsf_matrix_channel.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u64)]
        l.hsf_matrix_part_len.argtypes = [vp, u32, C.POINTER(u32)]
        l.hsf_matrix_free.argtypes = [vp]
        l.hsf_matrix_free.restype = None
        l.hsf_pack_vector.argtypes = [C.c_int, f32p, u64, u32p]
        l.hsf_unpack_result.argtypes = [C.c_int, u32p, u64, f32p]
        l.hsf_split_rows_by_nnz.argtypes = [u32p, u32, u32, u32, u32p]
        l.hsf_csr_to_csc.argtypes = [vp, C.c_int, u32p, u32p, u32p]
        _lib = l
    return _lib


def _check(rc):
    if rc != 0:
        raise HostError(f"hisparse host error {rc}: {lib().hsf_last_error().decode()}")


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def impl_id(impl):
    return IMPL_NAMES[impl] if isinstance(impl, str) else int(impl)


def default_banks(impl):
    """(v, o) in words: the bank sizes the shipped bitstreams use (sw/bm.sh:21-27, common.h:164-165)."""
    return 4096, (1024 if impl_id(impl) == IMPL_FLOAT_POB else 8192)


class CSRMatrix:
    """Owning handle on a C++ spmv::io::CSRMatrix<float>."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.hsf_csr_free(self._h)
            self._h = C.c_void_p(None)

    @classmethod
    def from_arrays(cls, num_rows, num_cols, indptr, indices, data):
        indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        data = np.ascontiguousarray(data, dtype=np.float32)
        if indptr.size != num_rows + 1 or indices.size != data.size:
            raise HostError("from_arrays: inconsistent CSR array lengths")
        h = C.c_void_p()
        _check(lib().hsf_csr_from_arrays(num_rows, num_cols, data.size, _u32p(indptr), _u32p(indices), _f32p(data), C.byref(h)))
        return cls(h.value)

    @classmethod
    def from_scipy(cls, m):
        m = m.tocsr()
        return cls.from_arrays(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)

    @classmethod
    def generate(cls, kind, num_rows, num_cols, a=0.0, b=0.0, c=1.0, seed=0):
        h = C.c_void_p()
        _check(lib().hsf_csr_generate(kind.encode(), num_rows, num_cols, float(a), float(b), float(c), seed, C.byref(h)))
        return cls(h.value)

    @property
    def dims(self):
        r, c, n = C.c_uint32(), C.c_uint32(), C.c_uint64()
        _check(lib().hsf_csr_dims(self._h, C.byref(r), C.byref(c), C.byref(n)))
        return r.value, c.value, n.value

    num_rows = property(lambda self: self.dims[0])
    num_cols = property(lambda self: self.dims[1])
    nnz = property(lambda self: self.dims[2])

    def arrays(self):
        """(indptr, indices, data) copies as numpy arrays."""
        r, _, n = self.dims
        indptr = np.empty(r + 1, dtype=np.uint32)
        indices = np.empty(n, dtype=np.uint32)
        data = np.empty(n, dtype=np.float32)
        _check(lib().hsf_csr_copy(self._h, _u32p(indptr), _u32p(indices), _f32p(data)))
        return indptr, indices, data

    def fill(self, value):
        _check(lib().hsf_csr_fill(self._h, float(value)))

    def normalize_by_outdegree(self):
        """util_normalize_csr_matrix_by_outdegree (sw/data_formatter.h:33-47)."""
        _check(lib().hsf_csr_normalize_by_outdegree(self._h))


def load_csr_matrix_from_float_npz(path):
    h = C.c_void_p()
    _check(lib().hsf_csr_load_npz(os.fsencode(path), C.byref(h)))
    return CSRMatrix(h.value)


class ChannelPackets:
    """The drop-in boundary payload: 16 channel packet buffers + geometry (owned by the C++ side)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        info = _Info()
        _check(lib().hsf_matrix_get_info(self._h, C.byref(info)))
        for name, _ in _Info._fields_:
            setattr(self, name, getattr(info, name))
        self.num_partitions = self.num_row_partitions * self.num_col_partitions

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.hsf_matrix_free(self._h)
            self._h = C.c_void_p(None)

    def channel_ptr(self, c):
        """(address, packet count) of channel c's 64-byte packets."""
        p, n = C.c_void_p(), C.c_uint64()
        _check(lib().hsf_matrix_channel(self._h, c, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def channel(self, c):
        """numpy COPY (n_packets, 16) uint32 of channel c: words 0-7 indices, 8-15 values."""
        addr, n = self.channel_ptr(c)
        if n == 0:
            return np.zeros((0, 16), dtype=np.uint32)
        buf = (C.c_uint32 * (n * 16)).from_address(addr)
        return np.frombuffer(buf, dtype=np.uint32).reshape(n, 16).copy()

    def part_len(self, row_partition):
        v = C.c_uint32()
        _check(lib().hsf_matrix_part_len(self._h, row_partition, C.byref(v)))
        return v.value


def format_matrix(csr, impl, vb_bank=None, ob_bank=None, skip_empty_rows=True):
    """CSR -> CPSR -> channel buffers.  Pads `csr`'s dimensions in place, like the reference."""
    impl = impl_id(impl)
    dv, do = default_banks(impl)
    h = C.c_void_p()
    _check(lib().hsf_format(csr._h, impl, ob_bank or do, vb_bank or dv, 1 if skip_empty_rows else 0, C.byref(h)))
    return ChannelPackets(h.value)


def pack_vector(impl, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    words = np.empty(x.size, dtype=np.uint32)
    _check(lib().hsf_pack_vector(impl_id(impl), _f32p(x), x.size, _u32p(words)))
    return words


def unpack_result(impl, words):
    words = np.ascontiguousarray(words, dtype=np.uint32)
    y = np.empty(words.size, dtype=np.float32)
    _check(lib().hsf_unpack_result(impl_id(impl), _u32p(words), words.size, _f32p(y)))
    return y


def split_rows_by_nnz_native(indptr, parts, granule):
    """The C++ routine the `benchmark --gpus N` driver uses (include/hisparse/row_sharding.h); sharding.py is its Python twin."""
    indptr = np.ascontiguousarray(indptr, dtype=np.uint32)
    bounds = np.zeros(parts + 1, dtype=np.uint32)
    _check(lib().hsf_split_rows_by_nnz(_u32p(indptr), indptr.size - 1, parts, granule, _u32p(bounds)))
    return [int(b) for b in bounds]


def csr_to_csc(csr, impl):
    """(indptr[num_cols + 1], row_indices[nnz], value_words[nnz]) -- csr2csc + conversion to the numeric mode's value words
    (sw/data_loader.h:109-157), the matrix form of the SpMSpV extension."""
    rows, cols, nnz = csr.dims
    indptr = np.zeros(cols + 1, dtype=np.uint32)
    idx = np.zeros(max(nnz, 1), dtype=np.uint32)
    words = np.zeros(max(nnz, 1), dtype=np.uint32)
    _check(lib().hsf_csr_to_csc(csr._h, impl_id(impl), _u32p(indptr), _u32p(idx), _u32p(words)))
    return indptr, idx[:nnz], words[:nnz]
