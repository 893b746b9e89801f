"""ISA invariants of the shipped gfx950 code (CPU test: unbundles hisparse_amd/lib/libhisparse_hip.so and disassembles it).

The SpMV kernels rely on a few things hipcc does not guarantee and that were each found by a failing measurement or a soak
(DESIGN.md section 4): the element stream is parked in ACCUMULATOR registers that the compiler must never touch on its own, every
hand-placed stream load sits behind `s_nop 4` (the hazard recogniser does not look into inline asm: scalar base written -> vector
memory use), every read of a parked register sits behind a counted `s_waitcnt vmcnt`, nothing spills to scratch, and the hot path has
no memory-side atomics.  A compiler upgrade that breaks one of these must fail HERE, not in a soak three days later.
"""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "hisparse_amd", "lib", "libhisparse_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(tmp_path):
    """The gfx950 code objects of every translation unit in the library's .hip_fatbin (clang offload bundles, one per .hip file)."""
    data = open(LIB, "rb").read()
    out, pos = [], 0
    while True:
        at = data.find(MAGIC, pos)
        if at < 0:
            break
        (entries,) = struct.unpack_from("<Q", data, at + len(MAGIC))
        off = at + len(MAGIC) + 8
        for _ in range(entries):
            o, size, tlen = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24: off + 24 + tlen].decode()
            off += 24 + tlen
            if triple.startswith("hip") and size:
                assert triple.endswith("gfx950"), f"unexpected offload target {triple}: this library carries gfx950 code only"
                path = tmp_path / f"co{len(out)}.co"
                path.write_bytes(data[at + o: at + o + size])
                out.append(str(path))
        pos = at + len(MAGIC)
    return out


def _metadata(co):
    """kernel name -> {key: int} from the code object's AMDGPU metadata note."""
    text = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"\s+(?:- )?\.(\w+):\s+(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.groups()
        if line.lstrip().startswith("- .") and key in ("agpr_count", "args"):     # first key of a kernel entry (keys are sorted)
            cur = {}
        if cur is None:
            continue
        if key == "name" and not line.lstrip().startswith("- ") and "symbol" not in cur:
            cur["name"] = val
        if key == "symbol":
            cur["symbol"] = val
            kernels[cur.get("name", val)] = cur
        if val.isdigit():
            cur[key] = int(val)
    return kernels


def _disassembly(co):
    """symbol -> list of instruction strings."""
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    funcs, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is not None and line.startswith("\t"):
            ins = line.split("//")[0].strip()
            if ins:
                cur.append(ins)
    return funcs


@pytest.fixture(scope="module")
def shipped(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.skip("libhisparse_hip.so has not been built")
    if not (os.path.exists(f"{LLVM}/llvm-objdump") and os.path.exists(f"{LLVM}/llvm-readelf")):
        pytest.skip("no llvm-objdump / llvm-readelf")
    tmp = tmp_path_factory.mktemp("isa")
    meta, code = {}, {}
    for co in _code_objects(tmp):
        meta.update(_metadata(co))
        code.update(_disassembly(co))
    shutil.rmtree(tmp, ignore_errors=True)
    assert meta and code
    return meta, code


ROWBLOCK = re.compile(r"spmv_rowblock_kernelILb([01])ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])E")
AGPR = re.compile(r"\ba(\d+|\[\d+:\d+\])")


def _hot_kernels(names):
    """the kernels of the SpMV hot path: the product variants of spmv_rowblock_kernel (ablate == 0), spmv_bitmap_kernel, the slice combine"""
    hot = []
    for n in names:
        m = ROWBLOCK.search(n)
        if m and int(m.group(3)) == 0:
            hot.append(n)
        elif "spmv_bitmap_kernel" in n and re.search(r"spmv_bitmap_kernelILb[01]ELi0E", n):
            hot.append(n)
        elif "combine_slices_kernel" in n or "spmv_sweep_kernel" in n or "spmv_light_kernel" in n:
            hot.append(n)
    return hot


def test_hot_kernels_have_no_scratch_no_memory_atomics_no_mfma(shipped):
    meta, code = shipped
    hot = _hot_kernels(meta)
    assert len([n for n in hot if "rowblock" in n]) >= 8 and any("bitmap" in n for n in hot) and any("combine" in n for n in hot), hot
    for n in hot:
        assert meta[n].get("private_segment_fixed_size", 0) == 0, f"{n} spills to scratch"
        body = code[n]
        bad = [i for i in body if re.match(r"(global|flat|buffer)_atomic", i)]
        assert not bad, f"{n}: memory-side atomics on the hot path: {bad[:3]}"
        assert not [i for i in body if i.startswith("v_mfma")], f"{n}: MFMA in a bandwidth-bound gather kernel"
        assert not [i for i in body if i.startswith("scratch_")], f"{n}: scratch access"


def test_element_ring_lives_in_accumulator_registers_behind_counted_waits(shipped):
    meta, code = shipped
    rowblock = [n for n in meta if ROWBLOCK.search(n) and int(ROWBLOCK.search(n).group(3)) == 0]     # the product variants (profiling builds aside)
    assert len(rowblock) >= 8
    for n in rowblock:
        assert meta[n]["agpr_count"] == 32, f"{n}: the ring is a0..a31, the compiler allocates none itself"
        body = code[n]
        loads = reads = 0
        for k, ins in enumerate(body):
            if not AGPR.search(ins.split(" ", 1)[1] if " " in ins else ""):
                continue
            prev = body[k - 1] if k else ""
            if ins.startswith("global_load_dword"):
                # a hand-placed stream load: behind `s_nop 4` (scalar base -> vector memory hazard), or the second load of the same asm block
                loads += 1
                ok = prev == "s_nop 4" or (prev.startswith("global_load_dword") and AGPR.search(prev))
                assert ok, f"{n}: `{ins}` follows `{prev}` instead of `s_nop 4`"
            elif ins.startswith("v_accvgpr_read_b32"):
                # a parked register becomes a compiler-visible value only behind the counted wait of the same asm block
                reads += 1
                ok = prev.startswith("s_waitcnt vmcnt(") or prev.startswith("v_accvgpr_read_b32")
                assert ok, f"{n}: `{ins}` follows `{prev}` instead of a counted s_waitcnt"
            else:
                raise AssertionError(f"{n}: `{ins}` touches an accumulator register outside the hand-written ring")
        assert loads and reads, f"{n}: no ring found ({loads} loads, {reads} reads)"


def test_sweep_ring_holds_chunks_and_gathers_behind_one_counted_wait(shipped):
    """spmv_sweep_kernel (spmv_sweep.hip): four chunks (a0..a7, dwordx2, nt) and four gathers of x (a8..a11, dword) per wavefront in
    accumulator registers the compiler allocates none of; every ring load behind `s_nop 4`; a step's three parked registers read behind
    ONE counted wait that leaves the 6 younger loads in flight; LDS atomics for the row sums, no x staging (no LDS-DMA)."""
    meta, code = shipped
    names = [n for n in meta if "spmv_sweep_kernel" in n]
    assert len(names) == 4, names      # fixed point and float, each with `nt` stream loads and -- for images that fit the Infinity Cache -- with `sc1` ones
    for n in names:
        assert meta[n]["agpr_count"] == 12 and meta[n].get("private_segment_fixed_size", 0) == 0
        body = code[n]
        chunks = gathers = reads = 0
        for k, ins in enumerate(body):
            if not AGPR.search(ins.split(" ", 1)[1] if " " in ins else ""):
                continue
            prev = body[k - 1] if k else ""
            if ins.startswith("global_load_dwordx2"):
                chunks += 1
                assert prev == "s_nop 4" and ins.endswith(" nt" if n.endswith("Lb1EEEvPKhPKNS0_5BlockEPKjPjiS9_NS1_14CarriedCombineE") else " sc1"), f"{n}: `{ins}` after `{prev}`"
            elif ins.startswith("global_load_dword "):
                gathers += 1
                assert prev == "s_nop 4" and not ins.endswith(" nt"), f"{n}: `{ins}` after `{prev}`"      # x is meant to stay in L2
            elif ins.startswith("v_accvgpr_read_b32"):
                reads += 1
                assert prev == "s_waitcnt vmcnt(6)" or prev.startswith("v_accvgpr_read_b32"), f"{n}: `{ins}` follows `{prev}`"
            else:
                raise AssertionError(f"{n}: `{ins}` touches an accumulator register outside the hand-written ring")
        assert chunks >= 8 and gathers >= 8 and reads == 3 * 4, (n, chunks, gathers, reads)      # prime + steady state; 4 steps x 3 registers
        # float: double sums; fixed point: wrapping 32-bit sums whose returned old value gives the carry, and a carry bit per row
        if "ILb1E" in n:
            assert any(i.startswith("ds_add_f64") for i in body)
        else:
            assert any(i.startswith("ds_add_rtn_u32") for i in body) and any(i.startswith("ds_or_b32") for i in body)
        assert not any("global_load_lds" in i for i in body)


def test_stream_loads_carry_their_plans_cache_policy_and_loaders_use_lds_dma(shipped):
    """kRing & 4 (round 6): the PAIRS / DELTA instantiations for images that stay in the Infinity Cache stream with `sc1`, every other
    row-block instantiation with `nt` -- all stream loads of one kernel with the same policy, and each policy exists for fixed and float."""
    meta, code = shipped
    seen = set()
    for n in meta:
        m = ROWBLOCK.search(n)
        if not m or int(m.group(3)) != 0:
            continue
        ring = int(m.group(2))
        assert ring in (0, 1, 2, 3, 4, 5), f"{n}: unknown ring format"
        policy = " sc1" if ring & 4 else " nt"
        seen.add((int(m.group(1)), ring))
        body = code[n]
        stream = [i for i in body if i.startswith("global_load_dword") and AGPR.search(i.split(" ", 1)[1])]
        assert stream and all(i.endswith(policy) for i in stream), f"{n}: stream loads without the{policy} policy"
        assert any("global_load_lds_dwordx4" in i for i in body), f"{n}: the x ring is not refilled by LDS-DMA"
        assert any(i.startswith("s_setprio") for i in body), f"{n}: loader wavefronts without raised priority"
    for is_float in (0, 1):
        for ring in (0, 1, 4, 5):
            assert (is_float, ring) in seen, f"no row-block kernel for float={is_float}, ring={ring}"


def test_spmm_kernel_uses_the_matrix_engine(shipped):
    """The one place in this library where a dense contraction exists: the SpMM over a float BITMAP matrix runs on v_mfma_f32_16x16x4_f32
    (spmm_mfma.hip), without scratch."""
    meta, code = shipped
    names = [n for n in meta if "spmm_mfma_kernel" in n]
    assert len(names) == 1
    body = code[names[0]]
    assert len([i for i in body if i.startswith("v_mfma_f32_16x16x4_f32")]) == 16
    assert meta[names[0]].get("private_segment_fixed_size", 0) == 0


def test_product_library_carries_no_profiling_instantiation(shipped):
    """the ablation / timeline / prefetch-depth instantiations -- most give wrong results by design -- are
    compiled only into libhisparse_hip_prof.so (-DHISPARSE_PROFILING).  Every spmv_rowblock_kernel / spmv_bitmap_kernel in the product
    library has ablate == 0 and the one depth its format ships with."""
    meta, _ = shipped
    rowblock = [ROWBLOCK.search(n) for n in meta if ROWBLOCK.search(n)]
    assert rowblock
    for m in rowblock:
        is_float, ring, ablate, depth, owner = (int(g) for g in m.groups())
        assert ablate == 0, f"profiling instantiation in the product library: {m.group(0)}"
        assert depth == (3 if ring == 3 else 8), f"experimental prefetch depth in the product library: {m.group(0)}"
        assert ring in (0, 1, 2, 3, 4, 5) and not (owner and ring & 4), f"unexpected ring format: {m.group(0)}"
    bitmap = [re.search(r"spmv_bitmap_kernelILb([01])ELi(\d+)E", n) for n in meta if "spmv_bitmap_kernel" in n]
    assert bitmap and all(int(m.group(2)) == 0 for m in bitmap)
    light = [re.search(r"spmv_light_kernelILb([01])ELi(\d+)E", n) for n in meta if "spmv_light_kernel" in n]
    assert all(m is None or int(m.group(2)) == 0 for m in light)
    sweep = [re.search(r"spmv_sweep_kernelILb([01])ELi(\d+)E", n) for n in meta if "spmv_sweep_kernel" in n]
    assert len(sweep) == 4 and all(int(m.group(2)) == 0 for m in sweep)      # fixed | float x `nt` | `sc1` stream loads


def test_product_library_refuses_profiling_switches():
    """...and an inherited HISPARSE_ABLATE cannot change what a production SpMV computes: the launch entry reports it (no GPU needed:
    the check sits in front of any HIP call)."""
    import ctypes as C
    lib = C.CDLL(LIB)
    probe = "_ZN8hisparse3dev22profiling_switch_errorEv"
    assert hasattr(lib, probe)
    fn = getattr(lib, probe)
    fn.restype = C.c_char_p
    old = {k: os.environ.pop(k, None) for k in ("HISPARSE_ABLATE", "HISPARSE_DEPTH")}
    try:
        assert fn() is None
        os.environ["HISPARSE_ABLATE"] = "3"
        assert b"libhisparse_hip_prof.so" in fn()
        os.environ["HISPARSE_ABLATE"] = "0"
        assert fn() is None
        os.environ["HISPARSE_DEPTH"] = "16"
        assert b"HISPARSE_DEPTH" in fn()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
