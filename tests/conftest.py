import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    # Build whatever is missing (host library with g++, oracle with gcc, HIP library with hipcc).
    # On the GPU box the prebuilt in-tree .so files travel with the snapshot, so this is a no-op there.
    need = [os.path.join(ROOT, "hisparse_amd", "lib", "libhisparse_host.so"),
            os.path.join(ROOT, "hisparse_amd", "lib", "libhisparse_hip.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "host", "hip", "oracle"], stdout=subprocess.DEVNULL)


def pytest_collection_modifyitems(config, items):
    # A bare `pytest tests/` on a machine without a GPU skips the gpu tests instead of failing them.
    if config.getoption("-m"):
        return
    try:
        has_gpu = os.path.exists("/dev/kfd") and any(n.startswith("renderD") for n in os.listdir("/dev/dri"))
    except OSError:
        has_gpu = False
    if not has_gpu:
        skip = pytest.mark.skip(reason="no GPU visible")
        for item in items:
            if "gpu" in item.keywords:
                item.add_marker(skip)
