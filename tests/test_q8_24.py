"""ap_ufixed<32,8,AP_RND,AP_SAT> as restated in three places that must agree: the product host library
(include/hisparse/q8_24.h via hsf_pack_vector), the oracle (oracle/cpu_ref.c) and exact integer arithmetic here.
The reference never exercises rounding or saturation: these rules follow the documented semantics of
the Xilinx type and are pinned only among the three restatements."""
import numpy as np
from hypothesis import given, settings, strategies as st

from hisparse_amd import host
from oracle import oracle as orc

MAX = 0xFFFFFFFF
u32 = st.integers(min_value=0, max_value=MAX)


def exact_mul(a, b):
    return min((a * b + (1 << 23)) >> 24, MAX)      # exact Q16.48 product, + half LSB (AP_RND), truncate, clamp (AP_SAT)


@settings(max_examples=300, deadline=None)
@given(u32, u32)
def test_mul_round_saturate(a, b):
    assert orc.lib().oracle_q_mul(a, b) == exact_mul(a, b)


@settings(max_examples=200, deadline=None)
@given(u32, u32)
def test_add_saturate(a, b):
    assert orc.lib().oracle_q_add(a, b) == min(a + b, MAX)


def test_mul_edges():
    one = 1 << 24
    q = orc.lib().oracle_q_mul
    assert q(one, one) == one
    assert q(MAX, MAX) == MAX                       # 256 * 256 saturates
    assert q(1, 1 << 23) == 1 and q(1, (1 << 23) - 1) == 0   # exactly half an LSB rounds up, just below rounds down
    assert q(3, 1 << 23) == 2                       # 1.5 LSB -> 2 (round half up)
    assert q(0, MAX) == 0


def test_saturating_accumulation_is_order_free():
    # the property the GPU path relies on: sat(sat(a+b)+c) == min(a+b+c, MAX) for non-negative terms, any order
    rng = np.random.default_rng(0)
    q_add = orc.lib().oracle_q_add
    for _ in range(200):
        terms = rng.integers(0, 2 ** 31, size=8, dtype=np.uint64).tolist()
        want = min(sum(terms), MAX)
        for perm in (terms, terms[::-1], sorted(terms)):
            acc = 0
            for t in perm:
                acc = q_add(acc, int(t))
            assert acc == want


@settings(max_examples=300, deadline=None)
@given(st.floats(min_value=-4.0, max_value=300.0, allow_nan=False, width=32))
def test_float_to_fixed(f):
    import math
    d = float(np.float32(f))
    want = 0 if not d > 0 else min(math.floor(d * 2 ** 24 + 0.5), MAX)
    v = np.array([f], dtype=np.float32)
    assert int(host.pack_vector(0, v)[0]) == want == int(orc.q_from_float(v)[0])


def test_fixed_to_float_roundtrip_of_exact_values():
    v = np.array([0.0, 1.0, 0.5, 255.0, 2.0 ** -24, 3.25], dtype=np.float32)
    w = host.pack_vector(0, v)
    assert np.array_equal(host.unpack_result(0, w), v)
    assert int(host.pack_vector(0, np.array([np.nan], dtype=np.float32))[0]) == 0
