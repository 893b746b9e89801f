#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python tools/probe_variants.py mouse_gene_slab4 "default:" "2sl:HISPARSE_COL_SLICES=2" "3sl:HISPARSE_COL_SLICES=3" "6sl:HISPARSE_COL_SLICES=6" "8sl:HISPARSE_COL_SLICES=8" "delta-6sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=6" 2>&1 | tail -6
timeout 900 python tools/probe_variants.py mouse_gene_slab2 "default:" "2sl:HISPARSE_COL_SLICES=2" "3sl:HISPARSE_COL_SLICES=3" "6sl:HISPARSE_COL_SLICES=6" "delta:HISPARSE_STREAM_FORMAT=delta" "delta-3sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=3" 2>&1 | tail -6
timeout 900 python tools/probe_variants.py mouse_gene "default:" "2sl:HISPARSE_COL_SLICES=2" "3sl:HISPARSE_COL_SLICES=3" "6sl:HISPARSE_COL_SLICES=6" 2>&1 | tail -4
timeout 900 python tools/probe_variants.py gplus "default:" "pairs-7sl:HISPARSE_COL_SLICES=7" "delta-7sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=7" 2>&1 | tail -3
} > gpurun_out/r03/slab2.log 2>&1
cat gpurun_out/r03/slab2.log
