#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1500 python -m pytest tests/test_perf_model.py -q -m gpu -s 2>&1 | grep -v "^re-tile\|^plan\|^  csr2" | tail -60
timeout 900 python -m pytest tests/test_gpu_soak.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -8
} > gpurun_out/r03/tests_a.log 2>&1
cat gpurun_out/r03/tests_a.log
