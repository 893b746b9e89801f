#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 1500 python tools/r06/float_owner_ab.py > gpurun_out/r06/float_pairs_vs_owner24.txt 2>&1
timeout 1800 python tools/planner_check.py --json gpurun_out/r06/planner_check_after3.json > gpurun_out/r06/planner_check_after3.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check_after3.txt | tail -27 | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_retile.py tests/test_gpu_planner.py -m gpu -q 2>&1 | tail -4
FUZZ_PROFILE=large timeout 900 python tests/gpu_fuzz_soak.py 150 6901 | tail -2
