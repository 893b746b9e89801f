#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_gpu_retile.py -x -q -m gpu -k "bitmap" 2>&1 | tail -3
for m in transformer_50 transformer_80 transformer_60; do
ROUNDS=4 timeout 900 python tools/probe_variants.py $m "default:" "no-round:HISPARSE_BITMAP_NO_ROUND=1" "equal:HISPARSE_BITMAP_SKEW=100/100/100/100" "180/115/70/35:HISPARSE_BITMAP_SKEW=180/115/70/35" "165/125/75/35:HISPARSE_BITMAP_SKEW=165/125/75/35" 2>&1 | tail -5
done
python tools/bitmap_timeline.py transformer_50 2>&1 | tail -7
} > gpurun_out/r03/bitmap_skew2.log 2>&1
cat gpurun_out/r03/bitmap_skew2.log
