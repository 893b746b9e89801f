#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
IMPL=fixed timeout 1200 python tools/probe_variants.py pokec "default:" "rows16369:HISPARSE_MAX_ROWS=16369" "rows12287:HISPARSE_MAX_ROWS=12287" "rows8191:HISPARSE_MAX_ROWS=8191" "5sl:HISPARSE_COL_SLICES=5" "8sl:HISPARSE_COL_SLICES=8" "8sl-16369:HISPARSE_COL_SLICES=8,HISPARSE_MAX_ROWS=16369" "pairs:HISPARSE_STREAM_FORMAT=pairs" "delta:HISPARSE_STREAM_FORMAT=delta" 2>&1 | tail -9
} > gpurun_out/r03/pokec_plans.log 2>&1
cat gpurun_out/r03/pokec_plans.log
