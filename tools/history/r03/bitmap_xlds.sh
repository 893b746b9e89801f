#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_spmm.py tests/test_gpu_fullsize.py -x -q -m gpu -k "bitmap or transformer or spmm or dense_rows" 2>&1 | tail -3
for m in transformer_50 transformer_80; do
ROUNDS=4 timeout 900 python tools/probe_variants.py $m "x-in-lds:" "x-through-l2:HISPARSE_BITMAP_X_LDS=0" 2>&1 | tail -2
done
IMPL=fixed ROUNDS=4 timeout 900 python tools/probe_variants.py transformer_50 "x-in-lds:" "x-through-l2:HISPARSE_BITMAP_X_LDS=0" 2>&1 | tail -2
python tools/bitmap_timeline.py transformer_50 2>&1 | tail -16
} > gpurun_out/r03/bitmap_xlds.log 2>&1
cat gpurun_out/r03/bitmap_xlds.log
