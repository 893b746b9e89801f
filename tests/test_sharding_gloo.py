"""Row-slab sharding across ranks (hisparse_amd/sharding.py) with a real process group: world_size 2, gloo, CPU.
Each rank formats its own slab with the product host library, computes its y slab with the oracle (no GPU here),
and the slabs are all-gathered the way bench.py gathers them over RCCL.  The assembled y must equal the y of the
unsharded matrix."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, impl, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from hisparse_amd import host, sharding
    from oracle import oracle as orc
    import cases

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    granule = 128 * (8 if impl == 2 else 1)
    m = cases.random_csr(5000, 400, 0.02, 77, impl)            # same seeded matrix on every rank
    x = cases.random_x(400, 77, impl)
    bounds = sharding.split_rows_by_nnz(m.indptr, world, granule)
    lo, hi = bounds[rank], bounds[rank + 1]
    ip, ix, dv = sharding.slab_arrays(m.indptr, m.indices, m.data, lo, hi)
    csr = host.CSRMatrix.from_arrays(hi - lo, 400, ip, ix, dv)
    cp = host.format_matrix(csr, impl, vb_bank=16, ob_bank=8, skip_empty_rows=True)
    xw = host.pack_vector(impl, np.concatenate([x, np.zeros(cp.num_cols - 400, dtype=np.float32)]))
    y = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                 cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    rows_all = [None] * world
    dist.all_gather_object(rows_all, (hi - lo, cp.num_rows))
    chunk = max(p for _, p in rows_all)
    mine = torch.zeros(chunk, dtype=torch.int32)
    mine[:cp.num_rows] = torch.from_numpy(y.view(np.int32))
    gathered = torch.zeros(chunk * world, dtype=torch.int32)
    dist.all_gather_into_tensor(gathered, mine)
    if rank == 0:
        layout = (chunk, [(i * chunk, r) for i, (r, _) in enumerate(rows_all)])
        np.save(out_path, sharding.assemble(gathered.numpy().view(np.uint32), layout))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("impl", [0, 2])
def test_two_rank_row_slabs_reassemble(tmp_path, impl):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hisparse_amd import host
    from oracle import oracle as orc
    import cases

    out = str(tmp_path / "y.npy")
    mp.spawn(_worker, args=(2, _free_port(), impl, out), nprocs=2, join=True)
    got = np.load(out)
    m = cases.random_csr(5000, 400, 0.02, 77, impl)
    csr = host.CSRMatrix.from_scipy(m)
    cp = host.format_matrix(csr, impl, vb_bank=16, ob_bank=8, skip_empty_rows=True)
    xw = host.pack_vector(impl, np.concatenate([cases.random_x(400, 77, impl), np.zeros(cp.num_cols - 400, dtype=np.float32)]))
    want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                    cp.num_col_partitions, cp.ob_bank, cp.vb_bank)[:5000]
    assert got.shape == (5000,)
    if impl == 0:
        assert np.array_equal(got, want)
    else:
        assert cases.float_close(got, want)


def test_split_rows_by_nnz_properties():
    from hisparse_amd import sharding
    rng = np.random.default_rng(0)
    deg = rng.zipf(1.7, size=20000).clip(max=5000)
    indptr = np.concatenate([[0], np.cumsum(deg)])
    for parts in (1, 2, 4, 8):
        b = sharding.split_rows_by_nnz(indptr, parts, 128)
        assert b[0] == 0 and b[-1] == 20000 and len(b) == parts + 1 and all(x <= y for x, y in zip(b, b[1:]))
        assert all(v % 128 == 0 for v in b[1:-1])
        loads = [indptr[b[i + 1]] - indptr[b[i]] for i in range(parts)]
        assert max(loads) <= indptr[-1] / parts * 1.25 + 5000 * 128
    chunk, spans = sharding.gather_layout([300, 129, 0], 128)
    assert chunk == 384 and spans == [(0, 300), (384, 129), (768, 0)]


def test_cpp_and_python_row_splits_agree():
    """include/hisparse/row_sharding.h (the C++ benchmark's --gpus N) and hisparse_amd/sharding.py cut the same slabs."""
    from hisparse_amd import host, sharding
    rng = np.random.default_rng(3)
    for trial in range(60):
        rows = int(rng.integers(1, 9000))
        deg = rng.integers(0, 40, rows)
        if trial % 3 == 0:
            deg[rng.integers(0, rows, 3)] += 20000          # a few very heavy rows
        indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.uint32)
        for parts in (1, 2, 3, 8):
            for granule in (128, 1024):
                want = sharding.split_rows_by_nnz(indptr, parts, granule)
                got = host.split_rows_by_nnz_native(indptr, parts, granule)
                assert got == want, (rows, parts, granule)
                assert got[0] == 0 and got[-1] == rows and all(a <= b for a, b in zip(got, got[1:]))
                assert all(b % granule == 0 for b in got[1:-1] if b != rows)
                if rows >= parts * granule:
                    assert all(b > a for a, b in zip(got, got[1:]))      # nobody is left without rows


class _FakeHip:
    """A stand-in for the HIP runtime of hisparse_amd/peer_gather.py (no GPU here): host buffers, handles = the buffer's address; rank
    `fail_rank` cannot open its peers' handles (two GPUs without peer access)."""

    def __init__(self, rank, fail_rank, fail_in):
        import ctypes as C
        self.C, self.rank, self.fail = C, rank, (rank == fail_rank, fail_in)
        self.keep = []

    def hipSetDevice(self, _):
        return 0

    def hipMalloc(self, pp, n):
        buf = self.C.create_string_buffer(n)
        self.keep.append(buf)
        pp._obj.value = self.C.addressof(buf)
        return 101 if self.fail == (True, "malloc") else 0

    def hipMemset(self, *_):
        return 0

    def hipIpcGetMemHandle(self, ph, p):
        return 0

    def hipIpcOpenMemHandle(self, pp, h, flags):
        pp._obj.value = 0x1000
        return 17 if self.fail == (True, "open") else 0

    def hipIpcCloseMemHandle(self, _):
        return 0

    def hipFree(self, _):
        return 0

    def hipGetErrorString(self, rc):
        return f"fake error {rc}".encode()


def _peer_setup_worker(rank, world, port, fail_in, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hisparse_amd import peer_gather
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        pg = peer_gather.PeerGather(dist, rank, world, 64, device_id=0, runtime=_FakeHip(rank, 1 if fail_in else -1, fail_in))
        outcome = f"ok {len(pg.targets(0))}"
    except peer_gather.PeerGatherError as e:
        outcome = f"error {e}"
    dist.barrier()                       # every rank is still in step with the others, whatever happened
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(outcome)
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_in", [None, "malloc", "open"])
def test_peer_gather_setup_fails_on_every_rank_together(tmp_path, fail_in):
    # bench_dist.py treats the peer-store gather as an extra: when ONE rank cannot allocate, export or open a buffer, every rank must leave
    # the set-up with the same error -- a rank that raised before a collective used to leave the others waiting in it
    import torch.multiprocessing as mp
    mp.spawn(_peer_setup_worker, args=(2, _free_port(), fail_in, str(tmp_path)), nprocs=2, join=True)
    got = [open(tmp_path / f"rank{r}.txt").read() for r in range(2)]
    if fail_in is None:
        assert got == ["ok 1", "ok 1"]
    else:
        assert all(g.startswith("error rank 1: ") for g in got) and got[0] == got[1]


def _worker8(rank, world, port, rows, out_path):
    """world_size 8 over gloo, fewer granules than ranks: some ranks own NO rows (bench_dist.py refuses such a run; the layout helpers and
    the collective must still be right for a caller that keeps the ranks in) and the slabs that exist are unequal (the last one is short)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from hisparse_amd import host, sharding
    from oracle import oracle as orc
    import cases

    impl, granule, cols = 0, 128, 304
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    m = cases.random_csr(rows, cols, 0.03, 5, impl)
    x = cases.random_x(cols, 5, impl)
    bounds = sharding.split_rows_by_nnz(m.indptr, world, granule)
    lo, hi = bounds[rank], bounds[rank + 1]
    y = np.zeros(0, dtype=np.uint32)
    if hi > lo:
        ip, ix, dv = sharding.slab_arrays(m.indptr, m.indices, m.data, lo, hi)
        cp = host.format_matrix(host.CSRMatrix.from_arrays(hi - lo, cols, ip, ix, dv), impl, vb_bank=16, ob_bank=8, skip_empty_rows=True)
        xw = host.pack_vector(impl, np.concatenate([x, np.zeros(cp.num_cols - cols, dtype=np.float32)]))
        y = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                     cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
        assert y.size == sharding.padded_rows(hi - lo, granule)
    slab_rows = [None] * world
    dist.all_gather_object(slab_rows, hi - lo)
    assert slab_rows == [bounds[i + 1] - bounds[i] for i in range(world)]          # every rank computed the same split
    layout = sharding.gather_layout(slab_rows, granule)
    chunk = layout[0]
    mine = torch.zeros(chunk, dtype=torch.int32)
    mine[:y.size] = torch.from_numpy(y.view(np.int32))
    gathered = torch.zeros(chunk * world, dtype=torch.int32)
    dist.all_gather_into_tensor(gathered, mine)
    whole = sharding.assemble(gathered.numpy().view(np.uint32), layout)
    sums = [None] * world
    dist.all_gather_object(sums, int(whole.astype(np.uint64).sum()))
    assert len(set(sums)) == 1                                                      # ... and holds the same assembled y
    if rank == world - 1:
        np.save(out_path, np.concatenate([np.asarray(slab_rows, dtype=np.uint32), whole]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rows", [700, 1500])
def test_eight_ranks_with_empty_and_unequal_slabs(tmp_path, rows):
    # 700 rows = 5 granules + 60 rows over 8 ranks: two ranks own nothing, the last slab is short; 1500 rows: every rank owns rows, unequal counts
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hisparse_amd import host, sharding
    from oracle import oracle as orc
    import cases

    out = str(tmp_path / "y.npy")
    mp.spawn(_worker8, args=(8, _free_port(), rows, out), nprocs=8, join=True)
    got = np.load(out)
    slab_rows, y = got[:8].astype(np.int64), got[8:]
    assert slab_rows.sum() == rows and y.shape == (rows,)
    if rows == 700:
        assert (slab_rows == 0).sum() == 2 and slab_rows.max() == 128 and 60 in slab_rows
    else:
        assert (slab_rows > 0).all() and len(set(slab_rows.tolist())) > 1
    m = cases.random_csr(rows, 304, 0.03, 5, 0)
    cp = host.format_matrix(host.CSRMatrix.from_scipy(m), 0, vb_bank=16, ob_bank=8, skip_empty_rows=True)
    xw = host.pack_vector(0, np.concatenate([cases.random_x(304, 5, 0), np.zeros(cp.num_cols - 304, dtype=np.float32)]))
    want = orc.spmv(0, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                    cp.num_col_partitions, cp.ob_bank, cp.vb_bank)[:rows]
    assert np.array_equal(y, want)
