#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_spmm.py -x -q -m gpu 2>&1 | tail -3
for k in 4 5 8 12 16 20 32; do timeout 600 python tools/spmm_probe.py transformer_50 $k 2>&1 | tail -1; done
HISPARSE_SPMM_MFMA=0 timeout 600 python tools/spmm_probe.py transformer_50 8 2>&1 | tail -1
} > gpurun_out/r03/mfma2.log 2>&1
cat gpurun_out/r03/mfma2.log
