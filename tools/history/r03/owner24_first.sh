#!/bin/bash
# round 3, first GPU call: OWNER24 parity + device/host image identity + same-box timing against the 8-byte OWNER form
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "owner" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_retile.py tests/test_gpu_load_csr.py -x -q -k "owner" 2>&1 | tail -5
timeout 900 python tools/probe_variants.py ogbn_products "owner24-d3:" "owner24-d2:HISPARSE_DEPTH=2" "owner24-d4:HISPARSE_DEPTH=4" "owner:HISPARSE_STREAM_FORMAT=owner" 2>&1 | tail -6
timeout 600 python tests/gpu_fuzz_soak.py 120 7 2>&1 | tail -8
} > gpurun_out/r03/owner24_first.log 2>&1
cat gpurun_out/r03/owner24_first.log
