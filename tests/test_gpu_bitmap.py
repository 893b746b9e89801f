"""spmv_bitmap_kernel beyond the shapes of tests/test_gpu_parity.py (which forces BITMAP onto every case small enough): the limits of the
path that keeps a block's stretch of x in LDS.  Through the C-ABI, against the oracle (bit-exact in fixed point, 1e-4 in the float modes)."""
import numpy as np
import pytest

from hisparse_amd import device, host
from oracle import oracle as orc

import cases

pytestmark = pytest.mark.gpu

IMPLS = [0, 1, 2]


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("rows,cols,slices,x_lds", [(64, 36864, "", None), (64, 36870, "", None), (64, 40000, "", None), (6, 30000, "3", None),
                                                    (300, 7000, "", None), (300, 7000, "", "0"), (1, 20000, "", "1"), (700, 4100, "2", None)])
def test_bitmap_x_in_lds_limits(impl, rows, cols, slices, x_lds, monkeypatch):
    """BITMAP with the block's stretch of x in LDS (spmv_bitmap.hip, kXLds): exactly 576 groups (the most the LDS holds), a last group that
    hangs over the end of x, one group too many (x through L2 again), column slices (several stretches per workgroup), workgroups with
    several blocks of one slice (the stretch is copied once), a one-row block, and both paths forced."""
    monkeypatch.setenv("HISPARSE_STREAM_FORMAT", "bitmap")
    monkeypatch.delenv("HISPARSE_AUX_BITS", raising=False)
    if slices:
        monkeypatch.setenv("HISPARSE_COL_SLICES", slices)
    if x_lds is not None:
        monkeypatch.setenv("HISPARSE_BITMAP_X_LDS", x_lds)
    csr = host.CSRMatrix.generate("bernoulli", rows, cols, b=0.3, c=1.0 if impl == 0 else 0.05, seed=rows + cols)
    cp = host.format_matrix(csr, impl, skip_empty_rows=True)
    xw = host.pack_vector(impl, cases.random_x(cp.num_cols, 23, impl))
    want = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                    cp.ob_bank, cp.vb_bank)
    with device.SpmvEngine(impl, ob_bank=cp.ob_bank, vb_bank=cp.vb_bank) as eng:
        eng.load_matrix(cp)
        eng.load_vector(xw)
        assert device.STREAM_FORMATS[eng.stats()["stream_format"]] == "bitmap"
        for _ in range(2):
            eng.run()
            got = eng.read_result()
            assert np.array_equal(got, want) if impl == 0 else cases.float_close(got, want)
        # another vector through the same LDS copy path: nothing of the first one may survive
        xw2 = host.pack_vector(impl, cases.random_x(cp.num_cols, 24, impl))
        want2 = orc.spmv(impl, [cp.channel(c) for c in range(16)], xw2, cp.num_rows, cp.num_cols, cp.num_row_partitions, cp.num_col_partitions,
                         cp.ob_bank, cp.vb_bank)
        eng.load_vector(xw2)
        eng.run()
        got = eng.read_result()
        assert np.array_equal(got, want2) if impl == 0 else cases.float_close(got, want2)
