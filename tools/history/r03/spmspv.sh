#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_spmspv.py tests/test_gpu_memory.py -q -m gpu 2>&1 | tail -4
timeout 600 python tools/spmspv_probe.py ogbl_ppa 2>&1 | tail -6
HISPARSE_SPMSPV=atomic timeout 600 python tools/spmspv_probe.py ogbl_ppa 2>&1 | tail -6
timeout 600 python tools/spmspv_probe.py mouse_gene 2>&1 | tail -6
HISPARSE_SPMSPV=atomic timeout 600 python tools/spmspv_probe.py mouse_gene 2>&1 | tail -6
} > gpurun_out/r03/spmspv.log 2>&1
cat gpurun_out/r03/spmspv.log
