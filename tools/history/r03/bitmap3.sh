#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
BEFORE=$PWD/hisparse_amd/lib/libhisparse_hip_before.so
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bitmap" 2>&1 | tail -3
for cfg in transformer_50 transformer_80; do
  for round in 1 2 3; do
    TAG="new" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
    HISPARSE_HIP_LIB=$BEFORE TAG="before" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
  done
done
timeout 600 python tools/bitmap_timeline.py transformer_50 2>&1 | tail -10
} > gpurun_out/r03/bitmap3.log 2>&1
cat gpurun_out/r03/bitmap3.log
