"""hisparse_amd.peer_gather — the y slabs of a row-sharded SpMV gathered by PEER STORES between the processes of one node.

bench.py shards one matrix by rows over N ranks (one process per GPU, BASELINE.json configs[4]); the reference leaves y sharded
(sw/benchmark.cpp:318-338), an iterative caller wants all of it on every GPU.  The collective way is RCCL's all-gather -- latency-bound at
these sizes (a slab of y is 20-300 KB).  This is the other way: every rank allocates its gather buffers with hipMalloc, exports them with
hipIpcGetMemHandle, opens the other ranks' with hipIpcOpenMemHandle (peer access over xGMI is enabled lazily by the runtime), and after every
SpMV one small kernel of the producer -- hs_push_result (hisparse_hip.h) -- stores its slab into its slot of every peer's buffer with plain
16-byte stores.  No collective on the critical path; the handles are exchanged once, through torch.distributed's object collectives.

Ordering: hs_push_result is stream-ordered behind the rank's own SpMV.  A consumer learns that ALL slabs of a step have arrived from a
barrier of the process group after the ranks have synchronised their streams (bench.py does that once around the timed region and once for
the correctness check); a per-step cross-process event is deliberately not part of this helper.

Works between processes that share ONE GPU as well (the "peers" are then other processes' buffers on the same device): that is how the
-m gpu test exercises it on the single-GPU box (tests/test_gpu_peer_gather.py)."""
import ctypes as C

import numpy as np

HANDLE_BYTES = 64            # hipIpcMemHandle_t
LAZY_PEER_ACCESS = 1         # hipIpcMemLazyEnablePeerAccess


class _IpcHandle(C.Structure):
    _fields_ = [("reserved", C.c_char * HANDLE_BYTES)]


def _runtime():
    rt = C.CDLL("libamdhip64.so")
    rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    rt.hipFree.argtypes = [C.c_void_p]
    rt.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    rt.hipSetDevice.argtypes = [C.c_int]
    rt.hipIpcGetMemHandle.argtypes = [C.POINTER(_IpcHandle), C.c_void_p]
    rt.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), _IpcHandle, C.c_uint]
    rt.hipIpcCloseMemHandle.argtypes = [C.c_void_p]
    rt.hipGetErrorString.restype = C.c_char_p
    rt.hipGetErrorString.argtypes = [C.c_int]
    return rt


class PeerGatherError(RuntimeError):
    pass


class PeerGather:
    """`buffers` gather buffers of world x chunk_words 32-bit words on this rank's device; slot r of every buffer belongs to rank r."""

    def __init__(self, dist, rank, world, chunk_words, device_id, buffers=2, runtime=None):
        if chunk_words % 4:
            raise PeerGatherError("chunk_words must be a multiple of 4 (hs_push_result stores 16 bytes per lane)")
        if world - 1 > 8:
            raise PeerGatherError("hs_push_result takes at most 8 destinations")
        self.rt, self.rank, self.world, self.chunk, self.device_id = runtime or _runtime(), rank, world, chunk_words, device_id      # (runtime: the tests' stand-in)
        self.bytes = world * chunk_words * 4
        self.mine, self.peers = [], []       # [buffer] -> own base pointer; [buffer][rank] -> that rank's base pointer as seen from here
        self._opened = []
        # Set-up is COLLECTIVE-SAFE: a rank whose allocation / export / open fails still takes part in both object collectives and every
        # rank raises together afterwards -- a rank that left early would leave the others waiting in a collective for ever (and with them
        # the bench line this helper is only an extra of).
        handles, err = [], None
        try:
            self._check(self.rt.hipSetDevice(device_id), "hipSetDevice")
            for _ in range(buffers):
                p = C.c_void_p()
                self._check(self.rt.hipMalloc(C.byref(p), self.bytes), "hipMalloc")
                self.mine.append(p.value)
                self._check(self.rt.hipMemset(p, 0, self.bytes), "hipMemset")
                h = _IpcHandle()
                self._check(self.rt.hipIpcGetMemHandle(C.byref(h), p), "hipIpcGetMemHandle")
                handles.append(C.string_at(C.byref(h), HANDLE_BYTES))
        except PeerGatherError as e:
            err = f"rank {rank}: {e}"
        everyone = [None] * world
        dist.all_gather_object(everyone, (handles, err))
        self._raise_together([e for _, e in everyone])
        try:
            for b in range(buffers):
                row = []
                for r in range(world):
                    if r == rank:
                        row.append(self.mine[b])
                        continue
                    h = _IpcHandle()
                    C.memmove(C.byref(h), everyone[r][0][b], HANDLE_BYTES)
                    p = C.c_void_p()
                    self._check(self.rt.hipIpcOpenMemHandle(C.byref(p), h, LAZY_PEER_ACCESS), f"hipIpcOpenMemHandle(rank {r})")
                    self._opened.append(p.value)
                    row.append(p.value)
                self.peers.append(row)
        except PeerGatherError as e:
            err = f"rank {rank}: {e}"
        opened = [None] * world
        dist.all_gather_object(opened, err)
        self._raise_together(opened)

    def _raise_together(self, errors):
        """every rank holds the same list: the first error anywhere ends the set-up on all of them"""
        first = next((e for e in errors if e), None)
        if first:
            self.close()
            raise PeerGatherError(first)

    def _check(self, rc, what):
        if rc != 0:
            raise PeerGatherError(f"{what}: {self.rt.hipGetErrorString(rc).decode()} (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed on this driver)")

    def my_slot(self, b):
        """device pointer of this rank's slot in its own buffer b: bind it as the SpMV's result (hs_bind_device_result)"""
        return self.mine[b] + self.rank * self.chunk * 4

    def targets(self, b):
        """this rank's slot in every OTHER rank's buffer b: the destinations of hs_push_result"""
        return [self.peers[b][r] + self.rank * self.chunk * 4 for r in range(self.world) if r != self.rank]

    def read(self, b):
        """this rank's gather buffer b as world x chunk words (after the ranks have synchronised)"""
        out = np.empty(self.world * self.chunk, dtype=np.uint32)
        self._check(self.rt.hipMemcpy(out.ctypes.data, C.c_void_p(self.mine[b]), self.bytes, 2), "hipMemcpy")
        return out.reshape(self.world, self.chunk)

    def close(self):
        for p in self._opened:
            self.rt.hipIpcCloseMemHandle(C.c_void_p(p))
        for p in self.mine:
            self.rt.hipFree(C.c_void_p(p))
        self._opened, self.mine = [], []
