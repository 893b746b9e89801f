"""hs_set_option (hisparse_hip.h): the library's configuration surface -- per context, winning over the environment -- and the refusal of
the profiling switches in the product library."""
import numpy as np
import pytest

from hisparse_amd import device, host

import cases

pytestmark = pytest.mark.gpu


def _case(impl=0):
    m = cases.random_csr(6000, 30000, 0.004, 11, impl)
    csr = host.CSRMatrix.from_scipy(m)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 5, impl))
    return cp, xw


def test_options_select_the_plan_per_context_and_beat_the_environment(monkeypatch):
    cp, xw = _case()
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "delta")
    results = {}
    for fmt in ("pairs", "delta", "owner24", None):
        with device.SpmvEngine(0) as eng:
            if fmt:
                eng.set_option("stream_format", fmt)           # without the prefix, lower case
            eng.set_option("HISPARSE_COL_SLICES", "2")           # the environment spelling works too
            eng.load_matrix(cp)
            st = eng.stats()
            assert device.STREAM_FORMATS[st["stream_format"]] == (fmt or "delta")       # None: the environment is the fallback
            assert st["col_slices"] == 2
            eng.load_vector(xw)
            eng.run()
            results[fmt] = eng.read_result()
            eng.set_option("col_slices", None)                   # cleared: the next load plans by itself
            eng.set_option("stream_format", "pairs")
            eng.load_matrix(cp)
            assert device.STREAM_FORMATS[eng.stats()["stream_format"]] == "pairs"
    ys = list(results.values())
    assert all(np.array_equal(ys[0], y) for y in ys[1:])         # no option changes WHAT is computed


def test_unknown_and_profiling_keys_are_refused():
    with device.SpmvEngine(0) as eng:
        for key in ("no_such_switch", "ablate", "HISPARSE_DEPTH"):
            with pytest.raises(device.DeviceError) as e:
                eng.set_option(key, "1")
            assert e.value.code == -1


def test_product_library_does_not_run_with_a_profiling_switch_in_the_environment(monkeypatch):
    cp, xw = _case()
    with device.SpmvEngine(0) as eng:
        eng.load_matrix(cp)
        eng.load_vector(xw)
        eng.run()
        want = eng.read_result()
        monkeypatch.setenv("HISPARSE_ABLATE", "3")
        with pytest.raises(device.DeviceError) as e:
            eng.run()
        assert e.value.code == -1 and "libhisparse_hip_prof.so" in str(e.value)
        with pytest.raises(device.DeviceError):
            eng.time_runs(0, 2)
        monkeypatch.delenv("HISPARSE_ABLATE")
        eng.run()
        assert np.array_equal(eng.read_result(), want)


def test_run_batch_plain_and_graph_replay_give_the_same_words():
    """hs_run_batch (round 5): K steps from one call -- enqueued from the library's C loop, or (batch_graph = 1) replayed from a captured
    hipGraph that is re-captured when the vector / result target changes and dropped by the next load."""
    import ctypes as C
    for impl in (0, 1):
        cp, xw = _case(impl)
        with device.SpmvEngine(impl) as eng:
            eng.set_option("col_slices", "3")                    # a step of two launches
            eng.load_matrix(cp)
            eng.load_vector(xw)
            eng.run()
            want = eng.read_result()
            for graph in ("0", "1"):
                eng.set_option("batch_graph", graph)
                for k in (1, 7, 7, 20):
                    eng.run_batch(k)
                    assert np.array_equal(eng.read_result(), want)
            # another vector: the captured graph baked the old one in only through the context's own buffer, which is overwritten in place
            x2 = host.pack_vector(impl, cases.random_x(cp.num_cols, 6, impl))
            eng.load_vector(x2)
            eng.run()
            want2 = eng.read_result()
            eng.run_batch(7)
            assert np.array_equal(eng.read_result(), want2) and not np.array_equal(want2, want)
            # a bound result buffer: re-capture with the new target
            rt = C.CDLL("libamdhip64.so")
            rt.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
            rt.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            rt.hipFree.argtypes = [C.c_void_p]
            buf = C.c_void_p()
            assert rt.hipMalloc(C.byref(buf), cp.num_rows * 4) == 0
            try:
                eng.bind_device_result(buf.value)
                eng.run_batch(7)
                eng.sync()
                y = np.empty(cp.num_rows, dtype=np.uint32)
                assert rt.hipMemcpy(y.ctypes.data, buf, y.nbytes, 2) == 0
                assert np.array_equal(y, want2)
                eng.bind_device_result(None)
            finally:
                rt.hipFree(buf)
            eng.load_matrix(cp)                                  # drops the graph
            eng.load_vector(xw)
            eng.run_batch(3)
            assert np.array_equal(eng.read_result(), want)


# ---- round 6: the row-block kernels' stream loads with or without the non-temporal hint (plan-time, hs_stats.stream_resident) ----------------
@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("fmt,slices", [("pairs", 1), ("pairs", 4), ("delta", 1), ("delta", 3)])
def test_rowblock_stream_policy_does_not_change_the_result(impl, fmt, slices):
    """Ring<kRing | 4> (spmv_kernels.hip): the `sc1` instantiation of the PAIRS / DELTA kernels for images that stay in the Infinity Cache, and
    the `nt` one, against the oracle -- single launches, a burst (the carried combine on sliced plans) and the partition loop."""
    from oracle import oracle as orc
    rows, cols = 40000, 70000
    g = host.CSRMatrix.generate("powerlaw", rows, cols, a=2.2e6, b=0.35, c=1.0 if impl == 0 else 2.0, seed=71 + impl)
    cp = host.format_matrix(g, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 13, impl))
    want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    got = {}
    for resident in ("0", "1"):
        with device.SpmvEngine(impl) as eng:
            eng.set_option("stream_format", fmt)
            eng.set_option("col_slices", str(slices))
            eng.set_option("light", "0")
            eng.set_option("stream_resident", resident)
            eng.load_matrix(cp)
            st = eng.stats()
            assert device.STREAM_FORMATS[st["stream_format"]] == fmt and st["col_slices"] == slices and st["stream_resident"] == int(resident)
            eng.load_vector(xw)
            eng.run()
            one = eng.read_result()
            eng.run_batch(4)
            burst = eng.read_result()
            for j in range(cp.num_row_partitions):
                eng.run_partition(j, cp.part_len(j))
            parts = eng.read_result()
        for y in (one, burst, parts):
            assert np.array_equal(y, want) if impl == 0 else cases.float_close(y, want)
        got[resident] = one
    assert np.array_equal(got["0"], got["1"])          # the cache policy of a load cannot change a bit, float modes included


def test_stream_resident_is_a_plan_time_choice_by_format_and_size():
    # what the plan takes by itself: a small multi-unit PAIRS / DELTA image -> resident; OWNER24 / BITMAP / LIGHT images never (their kernels have one policy)
    cp, xw = _case()
    for fmt, light, want in (("pairs", "0", 1), ("delta", "0", 1), ("owner24", "0", 0), ("pairs", "1", 0)):
        with device.SpmvEngine(0) as eng:
            eng.set_option("stream_format", fmt)
            eng.set_option("light", light)
            eng.load_matrix(cp)
            st = eng.stats()
            assert st["stream_resident"] == want, (fmt, light, st)
