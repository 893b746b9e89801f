#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
for m in ogbn_products ogbn_half_cols ogbn_quarter_cols pokec pokec_quarter_cols; do
timeout 1200 python tools/probe_variants.py $m "default:" 2>&1 | tail -1
done
} > gpurun_out/r03/heavy_units.log 2>&1
cat gpurun_out/r03/heavy_units.log
