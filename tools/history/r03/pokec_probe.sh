#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
IMPL=float_stall timeout 900 python tools/probe_variants.py pokec "o24:" "abl1-noLDS:HISPARSE_ABLATE=1" "abl4-norefill:HISPARSE_ABLATE=4" "abl5:HISPARSE_ABLATE=5" "abl8-nobarrier:HISPARSE_ABLATE=8" "abl12:HISPARSE_ABLATE=12" "abl13-stream:HISPARSE_ABLATE=13" \
   "o24-4sl:HISPARSE_COL_SLICES=4" "o24-5sl:HISPARSE_COL_SLICES=5" "o24-8sl:HISPARSE_COL_SLICES=8" 2>&1 | tail -11
sed -i 's/cfg, csr = datasets.load(name)/cfg, csr = datasets.load(name); cfg = cfg.__class__(**{**cfg.__dict__, "impl": "float_stall"})/' tools/owner_profile.py
timeout 600 python tools/owner_profile.py pokec 2>&1 | tail -5
timeout 600 python tools/owner_profile.py ogbn_products 2>&1 | tail -5
} > gpurun_out/r03/pokec_probe.log 2>&1
cat gpurun_out/r03/pokec_probe.log
