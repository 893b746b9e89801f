"""Host pre-processing (csr2cpsr + packet assembly) wall time per configuration, phases with HISPARSE_FORMAT_DEBUG=1:
python tools/format_time.py [config ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hisparse_amd import host, datasets

for name in sys.argv[1:] or ["ogbl_ppa", "mouse_gene", "ogbn_products", "transformer_50"]:
    cfg, csr = datasets.load(name)
    impl = host.impl_id(cfg.impl)
    best = 1e9
    for k in range(3):
        print("---- %s run %d" % (name, k), file=sys.stderr, flush=True)
        t0 = time.perf_counter()
        cp = host.format_matrix(csr, impl, skip_empty_rows=cfg.skip_empty_rows)
        best = min(best, time.perf_counter() - t0)
        del cp
    print("%-16s format_matrix best of 3: %.3f s (%d host threads)" % (name, best, os.cpu_count()), flush=True)
