#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
RUNS=50 bash tools/ab_lib.sh hisparse_amd/lib/libhisparse_hip_before.so ogbl_ppa mouse_gene ogbn_products
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
python tools/rowblock_timeline.py ogbl_ppa 2>&1 | head -16
} > gpurun_out/r03/prologue_ab2.log 2>&1
cat gpurun_out/r03/prologue_ab2.log
