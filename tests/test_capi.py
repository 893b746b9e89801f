"""The two shared libraries load on a machine without a GPU, export every symbol their headers declare, and the
device entry points fail loudly (no CPU fallback) when there is no gfx950 device."""
import ctypes
import os
import re

import pytest

from hisparse_amd import device, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:hs|hsf)_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    names = declared_functions("hisparse_hip.h")
    assert "hs_create" in names and "hs_run_partition" in names and "hs_tiles_build" in names
    lib = device.lib()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(device.EXPORTS) == names


def test_host_library_exports_every_declared_symbol():
    names = declared_functions("hisparse_host.h")
    lib = host.lib()
    for n in names:
        assert hasattr(lib, n), n


def test_error_strings():
    lib = device.lib()
    assert lib.hs_strerror(0) == b"ok"
    assert b"gfx950" in lib.hs_strerror(-2)
    assert lib.hs_strerror(-123) == b"unknown error"


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_cpu_fallback_without_gpu():
    with pytest.raises(device.DeviceError) as e:
        device.SpmvEngine(0)
    assert e.value.code in (-2, -3)


def test_create_rejects_bad_arguments():
    lib = device.lib()
    h = ctypes.c_void_p()
    assert lib.hs_create(None, 0, 0, 0, 0) == -1
    assert lib.hs_create(ctypes.byref(h), 0, 9, 0, 0) == -1 and not h.value
    assert lib.hs_run(None) == -1 and lib.hs_sync(None) == -1 and lib.hs_destroy(None) == 0
