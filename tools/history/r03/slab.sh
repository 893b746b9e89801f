#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python tools/probe_variants.py mouse_gene_slab8 "default:" "pairs:HISPARSE_STREAM_FORMAT=pairs" "delta:HISPARSE_STREAM_FORMAT=delta" \
  "2sl:HISPARSE_COL_SLICES=2" "3sl:HISPARSE_COL_SLICES=3" "6sl:HISPARSE_COL_SLICES=6" "rows64:HISPARSE_MAX_ROWS=64" "rows11:HISPARSE_MAX_ROWS=11" \
  "delta-6sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=6" "delta-3sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=3" "bitmap:HISPARSE_STREAM_FORMAT=bitmap" 2>&1 | tail -12
timeout 900 python tools/probe_variants.py mouse_gene_slab4 "default:" "delta:HISPARSE_STREAM_FORMAT=delta" "3sl:HISPARSE_COL_SLICES=3" "delta-3sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=3" 2>&1 | tail -5
timeout 300 python tools/rowblock_timeline.py mouse_gene_slab8 2>&1 | tail -16
} > gpurun_out/r03/slab.log 2>&1
cat gpurun_out/r03/slab.log
