"""Run BASELINE.json's configs on one GPU: parity against the oracle + kernel timing.  usage: python tests/gpu_run_configs.py [names...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
from hisparse_amd import datasets, device, host
from oracle import oracle as orc

names = sys.argv[1:] or ["csim_1k", "ogbl_ppa", "transformer_50", "mouse_gene", "ogbn_products"]
for name in names:
    t0 = time.time()
    cfg, csr = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    cp = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
    t_host = time.time() - t0
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 2, cp.num_cols).astype(np.float32) if impl == 0 else rng.normal(size=cp.num_cols).astype(np.float32)
    xw = host.pack_vector(impl, x)
    eng = device.SpmvEngine(impl)
    eng.load_matrix(cp)
    eng.load_vector(xw)
    st = eng.stats()
    eng.run()
    y = eng.read_result()
    t0 = time.time()
    want = orc.spmv(impl, [cp.channel_ptr(c)[0] for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions,
                    cp.num_col_partitions, cp.ob_bank, cp.vb_bank)
    t_cpu = time.time() - t0
    if impl == 0:
        ok = np.array_equal(y, want)
        detail = f"mismatches {int((y != want).sum())}, saturated rows {int((want == 0xFFFFFFFF).sum())}"
    else:
        a, b = y.view(np.float32), want.view(np.float32)
        ok = np.allclose(a, b, rtol=1e-4, atol=1e-4)
        detail = f"max abs err {np.abs(a - b).max():.3e}, max |y| {np.abs(b).max():.3e}"
    runs = 50
    tot, kern = eng.time_runs(5, runs)
    k = kern / runs
    print(f"{name:15s} {cfg.impl:11s} {cp.num_rows}x{cp.num_cols} nnz {cp.nnz} parts {cp.num_row_partitions}x{cp.num_col_partitions} "
          f"blocks {st['num_blocks']} units {st['num_units']} wgs {st['num_workgroups']} lds {st['lds_bytes']} | "
          f"{'PARITY OK' if ok else 'PARITY FAIL'} ({detail}) | kernel {k*1e3:.1f} us = {8*cp.nnz/(k*1e-3)/1e9:.0f} GB/s ({8*cp.nnz/(k*1e-3)/8e12*100:.1f} % roofline), "
          f"{2*cp.nnz/(k*1e-3)/1e9:.0f} GOPS | host gen+format {t_host:.1f}s load {st['load_seconds']:.2f}s oracle {t_cpu*1e3:.0f} ms", flush=True)
    eng.close()
