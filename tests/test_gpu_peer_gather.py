"""hisparse_amd/peer_gather.py + hs_push_result between PROCESSES: two ranks (gloo for the handle exchange and the barriers) share this
box's one GPU, each runs its row slab of one matrix through the HIP path with y bound into its slot of its own gather buffer and pushes the
slab into the other process's buffer (hipIpcGetMemHandle / hipIpcOpenMemHandle).  After a barrier both gather buffers must hold the whole y
of the unsharded matrix, bit for bit (oracle).  On an 8-GPU node the same code runs with one GPU per rank and the stores cross xGMI
(bench.py: `exchange_push`)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from hisparse_amd import device, host, peer_gather, sharding
    import cases

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    m = cases.random_csr(6000, 900, 0.02, 31, 0)
    x = cases.random_x(904, 31, 0)
    bounds = sharding.split_rows_by_nnz(m.indptr, world, 128)
    lo, hi = bounds[rank], bounds[rank + 1]
    ip, ix, dv = sharding.slab_arrays(m.indptr, m.indices, m.data, lo, hi)
    csr = host.CSRMatrix.from_arrays(hi - lo, 900, ip, ix, dv)
    cp = host.format_matrix(csr, 0, skip_empty_rows=True)
    xw = host.pack_vector(0, x[:cp.num_cols])
    rows_all = [None] * world
    dist.all_gather_object(rows_all, (hi - lo, cp.num_rows))
    chunk = max(p for _, p in rows_all)
    pg = peer_gather.PeerGather(dist, rank, world, chunk, device_id=0)
    with device.SpmvEngine(0) as eng:
        eng.load_matrix(cp)
        eng.load_vector(xw)
        for step in range(4):                      # both buffers, twice
            b = step & 1
            eng.bind_device_result(pg.my_slot(b))
            eng.run()
            eng.push_result(pg.targets(b), cp.num_rows)
        eng.sync()
        dist.barrier()                             # every rank's pushes have completed
        got = [pg.read(b) for b in (0, 1)]
        eng.bind_device_result(None)
    dist.barrier()                                 # nobody frees a buffer a peer still reads
    pg.close()
    np.save(os.path.join(out_dir, f"gathered_{rank}.npy"), np.stack(got))
    np.save(os.path.join(out_dir, f"rows_{rank}.npy"), np.array(rows_all))
    dist.destroy_process_group()


def test_two_processes_push_their_slabs_into_each_other():
    import tempfile
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from hisparse_amd import host, sharding
    from oracle import oracle as orc
    import cases

    with tempfile.TemporaryDirectory() as out:
        mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
        gathered = [np.load(os.path.join(out, f"gathered_{r}.npy")) for r in (0, 1)]
        rows_all = np.load(os.path.join(out, "rows_0.npy"))
    m = cases.random_csr(6000, 900, 0.02, 31, 0)
    cp = host.format_matrix(host.CSRMatrix.from_scipy(m), 0, skip_empty_rows=True)
    xw = host.pack_vector(0, cases.random_x(904, 31, 0)[:cp.num_cols])
    want = orc.spmv(0, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)[:6000]
    chunk = gathered[0].shape[2]
    layout = (chunk, [(i * chunk, int(r)) for i, (r, _) in enumerate(rows_all)])
    for rank in (0, 1):
        for b in (0, 1):
            y = sharding.assemble(gathered[rank][b].reshape(-1), layout)
            assert np.array_equal(y, want), f"rank {rank}, buffer {b}"
