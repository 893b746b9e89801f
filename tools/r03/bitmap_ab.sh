#!/bin/bash
# spmv_bitmap_kernel: the build in the tree against hisparse_amd/lib/libhisparse_hip_before.so, alternating on one box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
RUNS=50 bash tools/ab_lib.sh hisparse_amd/lib/libhisparse_hip_before.so transformer_50 transformer_80
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_spmm.py -x -q -m gpu -k "bitmap or transformer or spmm" 2>&1 | tail -3
} > gpurun_out/r03/bitmap_ab.log 2>&1
cat gpurun_out/r03/bitmap_ab.log
