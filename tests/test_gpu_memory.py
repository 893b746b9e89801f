"""Device memory hygiene: repeated loads over every path (CPSR / CSR source, all formats, host and GPU re-tile, SpMM, SpMSpV, errors
on the way) must give the memory back -- hipMemGetInfo before and after, through the runtime the library itself uses."""
import ctypes as C

import numpy as np
import pytest

from hisparse_amd import device, host

import cases

pytestmark = pytest.mark.gpu


def _free_bytes():
    device.lib()
    rt = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert rt.hipDeviceSynchronize() == 0
    assert rt.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_reloads_do_not_leak_device_memory(monkeypatch):
    mats = []
    for impl, kind, rows, cols, kw in [(0, "powerlaw", 60000, 80000, dict(a=3e6, b=0.3)), (2, "powerlaw", 90000, 120000, dict(a=400000, b=0.0)),
                                       (1, "bernoulli", 600, 4096, dict(b=0.4)), (0, "powerlaw", 4000, 3000, dict(a=40000, b=0.5))]:
        csr = host.CSRMatrix.generate(kind, rows, cols, c=1.0, seed=rows, **kw)
        mats.append((impl, csr, host.format_matrix(csr, impl, skip_empty_rows=True)))
    with device.SpmvEngine(0) as warm:                       # code objects, copy paths, allocator pools: paid once, before the first reading
        warm.load_matrix(mats[0][2])
    before = _free_bytes()
    for round_ in range(6):
        for impl, csr, cp in mats:
            for fmt in ("", "pairs", "delta", "owner", "sweep"):
                if fmt:
                    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", fmt)
                else:
                    monkeypatch.delenv("HISPARSE_STREAM_FORMAT", raising=False)
                monkeypatch.setenv("HISPARSE_RETILE", "host" if (round_ + len(fmt)) % 3 == 0 else "")
                with device.SpmvEngine(impl) as eng:
                    eng.load_matrix(cp)
                    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 1, impl))
                    eng.load_vector(xw)
                    eng.run()
                    eng.load_matrix_csr(csr)                  # reload of the same context from the other source
                    eng.run()
                    eng.spmm(np.stack([xw, xw, xw]))
                    ip, ix, vw = host.csr_to_csc(csr, impl)
                    eng.load_matrix_csc(ip, ix, vw, csr.num_rows)
                    eng.spmspv(np.arange(0, 50, dtype=np.uint32), xw[:50])
                    with pytest.raises(device.DeviceError):  # a failing load in between
                        eng.load_matrix_csr((3, 10, np.array([0, 2, 1, 3], dtype=np.uint32), np.array([4, 5, 1], dtype=np.uint32), np.ones(3, dtype=np.float32)))
    after = _free_bytes()
    assert before - after < (64 << 20), f"{(before - after) >> 20} MB of device memory did not come back"
