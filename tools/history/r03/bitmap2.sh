#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bitmap" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_spmm.py tests/test_gpu_fullsize.py -x -q -k "transformer or spmm" 2>&1 | tail -4
for cfg in transformer_50 transformer_80; do TAG=new python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"; done
IMPL=fixed timeout 300 python tools/probe_variants.py transformer_50 "fixed:" 2>&1 | tail -1
IMPL=fixed timeout 300 python tools/probe_variants.py transformer_70 "fixed:" 2>&1 | tail -1
timeout 600 python tools/bitmap_timeline.py transformer_50 2>&1 | tail -10
FUZZ_PROFILE=dense timeout 600 python tests/gpu_fuzz_soak.py 150 5 2>&1 | tail -4
} > gpurun_out/r03/bitmap2.log 2>&1
cat gpurun_out/r03/bitmap2.log
