"""hisparse_amd — MI355X-native SpMV engine behind HiSparse's host interface.

Layout:
  host.py      ctypes face of libhisparse_host.so  (CSR ingest, CSR -> CPSR formatter, channel assembly)
  device.py    ctypes face of libhisparse_hip.so   (the drop-in C-ABI: load / run / read back, gfx950 kernels)
  datasets.py  the reference's benchmark matrices as seeded stand-ins
  sharding.py  row-block sharding of one matrix across the GPUs of a node (+ RCCL gather of y)
  csrc/        C++ / HIP sources of the two libraries and the `benchmark` driver
"""
from . import host  # noqa: F401
from . import device  # noqa: F401
from . import datasets  # noqa: F401

__all__ = ["host", "device", "datasets"]
