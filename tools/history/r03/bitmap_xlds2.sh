#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
for m in transformer_50 transformer_80; do
ROUNDS=4 timeout 900 python tools/probe_variants.py $m "165/125/75/35:" "180/125/65/30:HISPARSE_BITMAP_SKEW=180/125/65/30" "175/130/70/25:HISPARSE_BITMAP_SKEW=175/130/70/25" "190/120/60/30:HISPARSE_BITMAP_SKEW=190/120/60/30" "170/140/65/25:HISPARSE_BITMAP_SKEW=170/140/65/25" 2>&1 | tail -5
done
} > gpurun_out/r03/bitmap_xlds2.log 2>&1
cat gpurun_out/r03/bitmap_xlds2.log
