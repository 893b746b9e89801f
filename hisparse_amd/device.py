"""hisparse_amd.device — ctypes binding of libhisparse_hip.so (include/hisparse_hip.h).

`SpmvEngine` is the Python face of the drop-in boundary: the object a reference driver would hold in
place of its cl::Context / cl::Kernel / cl::Buffer set (sw/benchmark.cpp:63-71,228-298).  Every
method is one C-ABI call; there is no Python or CPU compute path here — if the HIP library or a
gfx950 device is missing, construction raises.
"""
import ctypes as C
import os

import numpy as np

from . import host

_LIB_PATH = os.environ.get("HISPARSE_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libhisparse_hip.so")
_lib = None

EXPORTS = [
    "hs_strerror", "hs_last_error", "hs_create", "hs_destroy", "hs_load_matrix", "hs_load_vector", "hs_run",
    "hs_load_matrix_csr", "hs_run_batch", "hs_run_partition", "hs_sync", "hs_read_result", "hs_set_stream", "hs_get_stream", "hs_device_vector", "hs_device_result",
    "hs_bind_device_vector", "hs_bind_device_result", "hs_push_result", "hs_set_option", "hs_feedback", "hs_iterate", "hs_load_matrix_csc", "hs_spmspv", "hs_spmspv_device", "hs_read_spmspv_result", "hs_spmspv_status", "hs_spmm", "hs_spmm_device", "hs_get_stats", "hs_time_runs", "hs_time_kernel", "hs_debug_read_tiles", "hs_debug_read_mfma_image", "hs_tiles_build", "hs_tiles_info",
    "hs_tiles_copy", "hs_tiles_free", "hs_tiles_last_error",
]


class DeviceError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"hisparse_hip error {code}: {message}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [("nnz", C.c_uint64), ("cpsr_bytes", C.c_uint64), ("stream_bytes", C.c_uint64), ("stream_elements", C.c_uint64),
                ("num_blocks", C.c_uint32), ("num_units", C.c_uint32), ("num_workgroups", C.c_uint32), ("lds_bytes", C.c_uint32),
                ("num_compute_units", C.c_uint32), ("col_slices", C.c_uint32), ("ring_buffers", C.c_uint32), ("stream_format", C.c_uint32),
                ("load_seconds", C.c_double), ("retiled_on_gpu", C.c_uint32), ("light_kernel", C.c_uint32), ("stream_resident", C.c_uint32), ("reserved0", C.c_uint32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


STREAM_FORMATS = ("pairs", "delta", "bitmap", "owner", "pairs24", "owner24", "sweep")   # HS_STREAM_* (include/hisparse_hip.h)
CONSUMER_WAVES = 14
# device-side descriptors (hisparse_amd/csrc/stream_tiles.h)
BLOCK_DTYPE = np.dtype([("row0", "<u4"), ("nrows", "<u4"), ("row_part", "<u4"), ("unit_begin", "<u4"), ("unit_end", "<u4"),
                        ("flags", "<u4"), ("out_offset", "<u4"), ("next", "<u4"), ("wave_offset", "<u8", (CONSUMER_WAVES,)),
                        ("total_steps", "<u4", (CONSUMER_WAVES,)), ("first_end", "<u4", (CONSUMER_WAVES,)), ("first_col0", "<u4"),
                        ("first_ncols", "<u4"), ("last_part", "<u4"), ("next_part", "<u4"), ("pad", "<u4", (12,))])
UNIT_DTYPE = np.dtype([("col0", "<u4"), ("ncols", "<u4"), ("end_step", "<u4", (CONSUMER_WAVES,))])


def lib():
    """Load libhisparse_hip.so.  Loading needs the ROCm runtime library but no GPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise DeviceError(-3, f"{_LIB_PATH} is missing: run `make hip` (or __graft_entry__.build()); there is no fallback path")
        l = C.CDLL(_LIB_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        l.hs_strerror.restype = C.c_char_p
        l.hs_strerror.argtypes = [C.c_int]
        l.hs_last_error.restype = C.c_char_p
        l.hs_last_error.argtypes = [vp]
        l.hs_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, u32, u32]
        l.hs_destroy.argtypes = [vp]
        l.hs_load_matrix.argtypes = [vp, C.POINTER(vp), C.POINTER(u64), u32, u32, u32, u32]
        l.hs_load_vector.argtypes = [vp, vp, u32]
        l.hs_run.argtypes = [vp]
        l.hs_run_partition.argtypes = [vp, u32, u32]
        l.hs_sync.argtypes = [vp]
        l.hs_read_result.argtypes = [vp, vp, u32]
        l.hs_set_stream.argtypes = [vp, vp]
        l.hs_get_stream.argtypes = [vp, C.POINTER(vp)]
        l.hs_device_vector.argtypes = [vp, C.POINTER(vp)]
        l.hs_device_result.argtypes = [vp, C.POINTER(vp)]
        l.hs_bind_device_vector.argtypes = [vp, vp]
        l.hs_bind_device_result.argtypes = [vp, vp]
        l.hs_push_result.argtypes = [vp, C.POINTER(vp), u32, u32]
        l.hs_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
        l.hs_feedback.argtypes = [vp, u32, u32]
        l.hs_iterate.argtypes = [vp, u32, u32, u32]
        l.hs_load_matrix_csc.argtypes = [vp, vp, vp, vp, u32, u32]
        l.hs_spmspv.argtypes = [vp, vp, u32]
        l.hs_spmspv_device.argtypes = [vp, vp, u32]
        l.hs_read_spmspv_result.argtypes = [vp, vp, u32]
        l.hs_spmspv_status.argtypes = [vp, C.POINTER(u32), C.POINTER(vp)]
        l.hs_get_stats.argtypes = [vp, C.POINTER(Stats)]
        l.hs_time_runs.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.hs_run_batch.argtypes = [vp, u32]
        l.hs_time_kernel.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
        l.hs_debug_read_tiles.argtypes = [vp, vp, u64, vp, vp]
        l.hs_debug_read_mfma_image.argtypes = [vp, vp, u64, C.POINTER(u64)]
        l.hs_load_matrix_csr.argtypes = [vp, u32, u32, vp, vp, vp, C.POINTER(u32), C.POINTER(u32)]
        l.hs_spmm.argtypes = [vp, vp, u32, u32, vp, u32]
        l.hs_spmm_device.argtypes = [vp, vp, u64, vp, u64, u32]
        # (HISPARSE_HIP_LIB may name libhisparse_cpu.so, the separate host-thread build of the same boundary for machines without a
        # GPU: it has no re-tiling to introspect.  The default library must export everything: tests/test_capi.py.)
        if hasattr(l, "hs_tiles_build") or os.path.basename(_LIB_PATH) == "libhisparse_hip.so":
            l.hs_tiles_build.argtypes = [C.POINTER(vp), C.POINTER(u64), C.c_int, u32, u32, u32, u32, u32, u32, u32, C.POINTER(vp)]
            l.hs_tiles_info.argtypes = [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u64),
                                        C.POINTER(u64), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
            l.hs_tiles_copy.argtypes = [vp, vp, vp, vp, vp, vp]
            l.hs_tiles_free.argtypes = [vp]
            l.hs_tiles_free.restype = None
            l.hs_tiles_last_error.restype = C.c_char_p
        _lib = l
    return _lib


def _channel_arrays(packets):
    """(void*[16], uint64[16]) for a host.ChannelPackets or a list of 16 (n,16) uint32 arrays."""
    ptrs = (C.c_void_p * 16)()
    counts = (C.c_uint64 * 16)()
    keep = []
    for c in range(16):
        if isinstance(packets, host.ChannelPackets):
            addr, n = packets.channel_ptr(c)
        else:
            a = np.ascontiguousarray(packets[c], dtype=np.uint32)
            keep.append(a)
            addr, n = a.ctypes.data, a.shape[0]
        ptrs[c] = addr
        counts[c] = n
    return ptrs, counts, keep


class SpmvEngine:
    def __init__(self, impl, device_id=0, ob_bank=0, vb_bank=0):
        self._h = C.c_void_p()
        self.impl = host.impl_id(impl)
        rc = lib().hs_create(C.byref(self._h), device_id, self.impl, ob_bank, vb_bank)
        if rc != 0:
            raise DeviceError(rc, lib().hs_last_error(None).decode())
        default_vb, default_ob = host.default_banks(self.impl)
        self.ob_bank, self.vb_bank = ob_bank or default_ob, vb_bank or default_vb
        self.num_rows = self.num_cols = 0
        self.row_parts = self.col_parts = 0

    def _check(self, rc):
        if rc != 0:
            raise DeviceError(rc, lib().hs_last_error(self._h).decode() or lib().hs_strerror(rc).decode())

    def close(self):
        if self._h:
            lib().hs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    # ---- the boundary -------------------------------------------------------------------------
    def load_matrix(self, packets, num_rows=None, num_cols=None, num_row_partitions=None, num_col_partitions=None):
        """packets: host.ChannelPackets (dims taken from it) or 16 raw (n,16) uint32 arrays + explicit dims."""
        if isinstance(packets, host.ChannelPackets):
            num_rows, num_cols = packets.num_rows, packets.num_cols
            num_row_partitions, num_col_partitions = packets.num_row_partitions, packets.num_col_partitions
        ptrs, counts, keep = _channel_arrays(packets)
        self._check(lib().hs_load_matrix(self._h, ptrs, counts, num_rows, num_cols, num_row_partitions, num_col_partitions))
        del keep
        self.num_rows, self.num_cols = num_rows, num_cols
        self.row_parts, self.col_parts = num_row_partitions, num_col_partitions

    def load_matrix_csr(self, csr):
        """Straight from a host.CSRMatrix (or (num_rows, num_cols, indptr, indices, data) arrays) without csr2cpsr: the device pads,
        converts and re-tiles (hs_load_matrix_csr).  Sets num_rows / num_cols to the padded dimensions x and y then have."""
        if isinstance(csr, host.CSRMatrix):
            rows, cols = csr.num_rows, csr.num_cols
            indptr, indices, data = csr.arrays()
        else:
            rows, cols, indptr, indices, data = csr
        indptr, indices = (np.ascontiguousarray(a, dtype=np.uint32) for a in (indptr, indices))
        data = np.ascontiguousarray(data, dtype=np.float32)
        pr, pc = C.c_uint32(), C.c_uint32()
        self._check(lib().hs_load_matrix_csr(self._h, rows, cols, indptr.ctypes.data, indices.ctypes.data if indices.size else None,
                                             data.ctypes.data if data.size else None, C.byref(pr), C.byref(pc)))
        self.num_rows, self.num_cols = pr.value, pc.value
        self.row_parts, self.col_parts = -(-pr.value // (128 * self.ob_bank)), -(-pc.value // (8 * self.vb_bank))

    def load_vector(self, x_words):
        x_words = np.ascontiguousarray(x_words, dtype=np.uint32)
        self._check(lib().hs_load_vector(self._h, x_words.ctypes.data, x_words.size))

    def run(self):
        self._check(lib().hs_run(self._h))

    def run_batch(self, steps):
        """hs_run_batch: `steps` SpMVs from one call (option batch_graph = 1: replayed from a captured hipGraph)."""
        self._check(lib().hs_run_batch(self._h, int(steps)))

    def run_partition(self, row_part_id, part_len):
        self._check(lib().hs_run_partition(self._h, row_part_id, part_len))

    def sync(self):
        self._check(lib().hs_sync(self._h))

    def read_result(self):
        y = np.empty(self.num_rows, dtype=np.uint32)
        self._check(lib().hs_read_result(self._h, y.ctypes.data, y.size))
        return y

    def read_tiles(self):
        """What hs_load_matrix left on the device: dict(image, blocks, units) (tests compare it with build_tiles)."""
        st = self.stats()
        nbytes, nblocks, nunits = st["stream_bytes"], st["num_blocks"], st["num_units"]
        image = np.zeros(max(nbytes, 1), dtype=np.uint8)
        blocks = np.zeros(max(nblocks, 1), dtype=BLOCK_DTYPE)
        units = np.zeros(max(nunits, 1), dtype=UNIT_DTYPE)
        self._check(lib().hs_debug_read_tiles(self._h, image.ctypes.data, image.size, blocks.ctypes.data, units.ctypes.data))
        return dict(image=image[:nbytes], blocks=blocks[:nblocks], units=units[:nunits])

    def read_mfma_image(self):
        """The second image of a float BITMAP matrix (SpMM on the matrix engine), as bytes; empty when there is none."""
        n = C.c_uint64()
        self._check(lib().hs_debug_read_mfma_image(self._h, None, 0, C.byref(n)))
        words = np.zeros(max(n.value, 1), dtype=np.uint8)
        if n.value:
            self._check(lib().hs_debug_read_mfma_image(self._h, words.ctypes.data, words.size, C.byref(n)))
        return words[:n.value]

    def set_option(self, key, value=None):
        """hs_set_option: a tuning switch of this context (plan-time keys apply to the next load_matrix*); None clears it."""
        self._check(lib().hs_set_option(self._h, str(key).encode(), None if value is None else str(value).encode()))

    # ---- zero-copy hooks ----------------------------------------------------------------------
    def set_stream(self, hip_stream):
        self._check(lib().hs_set_stream(self._h, C.c_void_p(hip_stream or None)))

    def get_stream(self):
        p = C.c_void_p()
        self._check(lib().hs_get_stream(self._h, C.byref(p)))
        return p.value

    def device_vector(self):
        p = C.c_void_p()
        self._check(lib().hs_device_vector(self._h, C.byref(p)))
        return p.value

    def device_result(self):
        p = C.c_void_p()
        self._check(lib().hs_device_result(self._h, C.byref(p)))
        return p.value

    def bind_device_vector(self, ptr):
        self._check(lib().hs_bind_device_vector(self._h, C.c_void_p(ptr or None)))

    def bind_device_result(self, ptr):
        self._check(lib().hs_bind_device_result(self._h, C.c_void_p(ptr or None)))

    # ---- SpMSpV extension ---------------------------------------------------------------------------
    def load_matrix_csc(self, indptr, row_indices, value_words, num_rows):
        indptr, row_indices, value_words = (np.ascontiguousarray(a, dtype=np.uint32) for a in (indptr, row_indices, value_words))
        self._check(lib().hs_load_matrix_csc(self._h, indptr.ctypes.data, row_indices.ctypes.data if row_indices.size else None,
                                            value_words.ctypes.data if value_words.size else None, num_rows, indptr.size - 1))
        self.csc_rows = num_rows

    def spmspv(self, x_index, x_words):
        """y = A x for x = {(x_index[k], x_words[k])}; returns the dense packed y."""
        pairs = np.empty((len(x_index), 2), dtype=np.uint32)       # IDX_VAL_T: {index, val}
        pairs[:, 0] = x_index
        pairs[:, 1] = x_words
        self._check(lib().hs_spmspv(self._h, pairs.ctypes.data if pairs.size else None, len(x_index)))
        y = np.empty(self.csc_rows, dtype=np.uint32)
        self._check(lib().hs_read_spmspv_result(self._h, y.ctypes.data, y.size))
        return y

    def spmspv_async(self, x_index, x_words):
        """hs_spmspv without reading y back (asynchronous): for timing loops; read_spmspv_result() fetches the last result."""
        pairs = np.empty((len(x_index), 2), dtype=np.uint32)
        pairs[:, 0] = x_index
        pairs[:, 1] = x_words
        self._check(lib().hs_spmspv(self._h, pairs.ctypes.data if pairs.size else None, len(x_index)))

    def spmspv_device(self, pairs_dev, count):
        """hs_spmspv_device: `count` IDX_VAL_T pairs already in device memory (pointer as int)."""
        self._check(lib().hs_spmspv_device(self._h, C.c_void_p(pairs_dev), count))

    def read_spmspv_result(self):
        y = np.empty(self.csc_rows, dtype=np.uint32)
        self._check(lib().hs_read_spmspv_result(self._h, y.ctypes.data, y.size))
        return y

    def spmm(self, x_words):
        """Y = A X for X of shape (k, num_cols) value words (one packed vector per row); returns (k, num_rows) words."""
        x_words = np.ascontiguousarray(x_words, dtype=np.uint32)
        k = x_words.shape[0]
        y = np.empty((k, self.num_rows), dtype=np.uint32)
        self._check(lib().hs_spmm(self._h, x_words.ctypes.data, x_words.shape[1], k, y.ctypes.data, self.num_rows))
        return y

    def spmm_device(self, x_dev, ldx, y_dev, ldy, k):
        self._check(lib().hs_spmm_device(self._h, C.c_void_p(x_dev), ldx, C.c_void_p(y_dev), ldy, k))

    # ---- measurement ----------------------------------------------------------------------------
    def stats(self):
        s = Stats()
        self._check(lib().hs_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def feedback(self, scale_word, shift_word):
        """x[i] = scale (*) y[i] (+) shift on the device (extension for iterative callers; words from host.pack_vector)."""
        self._check(lib().hs_feedback(self._h, int(scale_word), int(shift_word)))

    def iterate(self, iterations, scale_word, shift_word):
        """`iterations` x { run; feedback }, replayed from one captured hipGraph."""
        self._check(lib().hs_iterate(self._h, int(iterations), int(scale_word), int(shift_word)))

    def push_result(self, dst_ptrs, num_words):
        """hs_push_result: the first num_words result words into the given device buffers (stream-ordered, plain stores)."""
        arr = (C.c_void_p * len(dst_ptrs))(*dst_ptrs)
        self._check(lib().hs_push_result(self._h, arr, len(dst_ptrs), num_words))

    def time_runs(self, warmup, runs, kernel=True):
        """(total_ms for `runs` SpMVs, summed duration of the dominant kernel over those runs or None)."""
        total, kern = C.c_float(), C.c_float()
        self._check(lib().hs_time_runs(self._h, warmup, runs, C.byref(total), C.byref(kern) if kernel else None))
        return total.value, (kern.value if kernel else None)

    def time_kernel(self, warmup, runs):
        """hs_time_kernel: ms for `runs` back-to-back launches of the SpMV kernel alone (one HIP event pair around the loop)."""
        kern = C.c_float()
        self._check(lib().hs_time_kernel(self._h, warmup, runs, C.byref(kern)))
        return kern.value


def build_tiles(packets, impl, ob_bank, vb_bank, num_rows, num_cols, num_row_partitions, num_col_partitions, max_workgroups):
    """What hs_load_matrix would upload (no GPU involved): dict(image, blocks, units, wg_first, block_order, ...)."""
    l = lib()
    ptrs, counts, keep = _channel_arrays(packets)
    h = C.c_void_p()
    rc = l.hs_tiles_build(ptrs, counts, host.impl_id(impl), ob_bank, vb_bank, num_rows, num_cols, num_row_partitions,
                          num_col_partitions, max_workgroups, C.byref(h))
    del keep
    if rc != 0:
        raise DeviceError(rc, l.hs_tiles_last_error().decode())
    try:
        nbytes, nnz, elems = C.c_uint64(), C.c_uint64(), C.c_uint64()
        nblocks, nunits, nwg, maxrows, slices, ring, fmt = (C.c_uint32() for _ in range(7))
        l.hs_tiles_info(h, C.byref(nbytes), C.byref(nblocks), C.byref(nunits), C.byref(nwg), C.byref(maxrows), C.byref(nnz), C.byref(elems),
                        C.byref(slices), C.byref(ring), C.byref(fmt))
        image = np.zeros(max(nbytes.value, 1), dtype=np.uint8)
        blocks = np.zeros(max(nblocks.value, 1), dtype=BLOCK_DTYPE)
        units = np.zeros(max(nunits.value, 1), dtype=UNIT_DTYPE)
        wg_first = np.zeros(nwg.value + 1, dtype=np.uint32)
        order = np.zeros(max(nblocks.value, 1), dtype=np.uint32)
        l.hs_tiles_copy(h, image.ctypes.data, blocks.ctypes.data, units.ctypes.data, wg_first.ctypes.data, order.ctypes.data)
        return dict(image=image[:nbytes.value], blocks=blocks[:nblocks.value], units=units[:nunits.value], wg_first=wg_first,
                    block_order=order[:nblocks.value], num_workgroups=nwg.value, max_block_rows=maxrows.value, nnz=nnz.value,
                    elements=elems.value, col_slices=slices.value, ring_buffers=ring.value,
                    format=STREAM_FORMATS[fmt.value])
    finally:
        l.hs_tiles_free(h)
