#!/bin/bash
# spmv_bitmap_kernel: the build in the tree against hisparse_amd/lib/libhisparse_hip_before.so, alternating on one box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_spmm.py tests/test_gpu_fullsize.py -x -q -m gpu -k "bitmap or transformer or spmm or dense" 2>&1 | tail -3
RUNS=50 bash tools/ab_lib.sh hisparse_amd/lib/libhisparse_hip_before.so transformer_50 transformer_80
python tools/bitmap_timeline.py transformer_50 2>&1 | head -12
} > gpurun_out/r03/bitmap_ab.log 2>&1
cat gpurun_out/r03/bitmap_ab.log
