#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "owner24" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_retile.py tests/test_gpu_load_csr.py -x -q -k "owner24" 2>&1 | tail -4
timeout 900 python tools/probe_variants.py pokec "owner24:" "pairs:HISPARSE_STREAM_FORMAT=pairs" "o24-spread:HISPARSE_XCD_AFFINITY=0" 2>&1 | tail -4
IMPL=fixed timeout 900 python tools/probe_variants.py ogbn_products "owner24:" "pairs:HISPARSE_STREAM_FORMAT=pairs" 2>&1 | tail -3
IMPL=float_stall timeout 900 python tools/probe_variants.py pokec "owner24:" "owner:HISPARSE_STREAM_FORMAT=owner" 2>&1 | tail -3
timeout 600 python tests/gpu_fuzz_soak.py 100 11 2>&1 | tail -5
} > gpurun_out/r03/fixed_owner.log 2>&1
cat gpurun_out/r03/fixed_owner.log
