#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1200 python tools/slab_probe.py mouse_gene 8 "default:" "depth16:HISPARSE_DEPTH=16" 2>&1 | head -6
timeout 1200 python tools/slab_probe.py mouse_gene 4 "default:" "pairs:HISPARSE_STREAM_FORMAT=pairs" "pairs-depth16:HISPARSE_STREAM_FORMAT=pairs,HISPARSE_DEPTH=16" 2>&1 | head -6
timeout 900 python tools/probe_variants.py gplus "default:" "pairs:HISPARSE_STREAM_FORMAT=pairs" "pairs-depth16:HISPARSE_STREAM_FORMAT=pairs,HISPARSE_DEPTH=16" 2>&1 | tail -3
timeout 900 python tools/probe_variants.py mouse_gene_slab8 "default:" "depth16:HISPARSE_DEPTH=16" 2>&1 | tail -2
} > gpurun_out/r03/slab_depth.log 2>&1
cat gpurun_out/r03/slab_depth.log
