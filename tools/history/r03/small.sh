#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python tools/probe_variants.py gplus "default:" "2sl:HISPARSE_COL_SLICES=2" "4sl:HISPARSE_COL_SLICES=4" "8sl:HISPARSE_COL_SLICES=8" "delta:HISPARSE_STREAM_FORMAT=delta" "delta-4sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=4" 2>&1 | tail -6
IMPL=fixed timeout 900 python tools/probe_variants.py transformer_90 "default:" "2sl:HISPARSE_COL_SLICES=2" "4sl:HISPARSE_COL_SLICES=4" "bitmap:HISPARSE_STREAM_FORMAT=bitmap" "delta:HISPARSE_STREAM_FORMAT=delta" 2>&1 | tail -5
IMPL=fixed timeout 900 python tools/probe_variants.py transformer_95 "default:" "4sl:HISPARSE_COL_SLICES=4" "bitmap:HISPARSE_STREAM_FORMAT=bitmap" 2>&1 | tail -3
timeout 900 python tools/probe_variants.py ppa_small "default:" "2sl:HISPARSE_COL_SLICES=2" "4sl:HISPARSE_COL_SLICES=4" "8sl:HISPARSE_COL_SLICES=8" 2>&1 | tail -4
} > gpurun_out/r03/small.log 2>&1
cat gpurun_out/r03/small.log
