#!/bin/bash
# round 3: blocks assigned to XCDs by column slice (x of ogbn-products no longer misses L2) + what bounds the OWNER24 kernel
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1500 python tools/probe_variants.py ogbn_products "o24-affine:" "o24-spread:HISPARSE_XCD_AFFINITY=0" "o24-aff-8sl:HISPARSE_COL_SLICES=8" "o24-aff-4sl:HISPARSE_COL_SLICES=4" \
   "o24-aff-6sl:HISPARSE_COL_SLICES=6" "o24-aff-3sl:HISPARSE_COL_SLICES=3" "owner8B-affine:HISPARSE_STREAM_FORMAT=owner" \
   "abl1-noLDS:HISPARSE_ABLATE=1" "abl4-norefill:HISPARSE_ABLATE=4" "abl5:HISPARSE_ABLATE=5" "abl8-nobarrier:HISPARSE_ABLATE=8" "abl12:HISPARSE_ABLATE=12" "abl13-stream-only:HISPARSE_ABLATE=13" 2>&1 | tail -14
timeout 600 python tools/probe_variants.py ogbl_ppa "ppa-default:" "ppa-affine:HISPARSE_XCD_AFFINITY=1" 2>&1 | tail -3
} > gpurun_out/r03/owner24_second.log 2>&1
cat gpurun_out/r03/owner24_second.log
