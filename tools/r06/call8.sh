#!/bin/bash
# round 6, call 8: hub rows (per-lane sums for DELTA blocks with a dominant row; DELTA kept where hub rows hold >= 30 %): parity, the out-of-sample check, the reference's matrices
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_retile.py tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_carry.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tests/gpu_fuzz_soak.py 200 6701 | tail -3
FUZZ_PROFILE=dense timeout 600 python tests/gpu_fuzz_soak.py 100 6702 | tail -2
timeout 1800 python tools/planner_check.py --json gpurun_out/r06/planner_check_after2.json > gpurun_out/r06/planner_check_after2.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check_after2.txt | tail -27 | cut -c1-200
timeout 2400 python tools/planner_check.py --reference --json gpurun_out/r06/planner_check_reference2.json > gpurun_out/r06/planner_check_reference2.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check_reference2.txt | tail -21 | cut -c1-200
