#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1200 python tools/slab_probe.py mouse_gene 4 "default:" "delta:HISPARSE_STREAM_FORMAT=delta" "delta-3sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=3" "delta-6sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=6" "pairs-6sl:HISPARSE_COL_SLICES=6"
timeout 1200 python tools/slab_probe.py mouse_gene 2 "default:" "delta:HISPARSE_STREAM_FORMAT=delta" "delta-3sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=3" "delta-6sl:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=6"
} > gpurun_out/r03/slab3.log 2>&1
cat gpurun_out/r03/slab3.log
