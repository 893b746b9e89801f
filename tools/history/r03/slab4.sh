#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1200 python tools/slab_probe.py mouse_gene 8 "default:"
timeout 1200 python tools/slab_probe.py mouse_gene 4 "default:"
timeout 1200 python tools/slab_probe.py mouse_gene 2 "default:"
timeout 900 python tools/probe_variants.py nn_small "default:" "4sl:HISPARSE_COL_SLICES=4" 2>&1 | tail -2
timeout 900 python tools/probe_variants.py gplus "default:" "7sl:HISPARSE_COL_SLICES=7" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_retile.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/r03/slab4.log 2>&1
cat gpurun_out/r03/slab4.log
