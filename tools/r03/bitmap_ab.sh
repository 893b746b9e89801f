#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
for m in transformer_50 transformer_80; do
ROUNDS=5 timeout 900 python tools/probe_variants.py $m "new:" "old:HISPARSE_ABLATE=512" "nothing:HISPARSE_ABLATE=7" 2>&1 | tail -3
done
} > gpurun_out/r03/bitmap_ab.log 2>&1
cat gpurun_out/r03/bitmap_ab.log
