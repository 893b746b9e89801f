"""A slice of the randomized soak (tests/gpu_fuzz_soak.py) inside `pytest -m gpu`: random shapes / densities / bank sizes / numeric modes /
stream formats / column slices, 4 launches + the partition-by-partition loop per case against the oracle, and every image the device
built against the host builder's, byte for byte.  The three hazards the SpMV kernels work around by hand (compiler
copies of in-flight registers, scalar-base hazards in inline asm, LDS atomics that outlive lgkmcnt) were all FOUND by this kind of run;
tests/test_isa_invariants.py pins their fixes in the shipped code, this pins the behaviour."""
import pytest

import gpu_fuzz_soak

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("profile,cases,seed", [(None, 60, 20260928), ("dense", 25, 3), ("large", 15, 4)])
def test_fuzz_soak_slice(profile, cases, seed):
    fails, on_gpu = gpu_fuzz_soak.run(cases, seed, profile, verbose=False)
    assert fails == 0
    assert on_gpu > 0
