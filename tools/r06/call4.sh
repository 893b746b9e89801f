#!/bin/bash
# round 6, GPU call 4: the planner with the tile census + the refitted SWEEP model: (a) device-built == host-built images, (b) out-of-sample check, (c) the reference's own matrices
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 1200 python -m pytest tests/test_gpu_retile.py tests/test_gpu_sweep.py tests/test_gpu_load_csr.py tests/test_gpu_light.py -m gpu -x -q > gpurun_out/r06/call4_tests.log 2>&1
tail -3 gpurun_out/r06/call4_tests.log
timeout 1800 python tools/planner_check.py --json gpurun_out/r06/planner_check_after.json > gpurun_out/r06/planner_check_after.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check_after.txt | tail -30
timeout 2400 python tools/planner_check.py --reference --json gpurun_out/r06/planner_check_reference.json > gpurun_out/r06/planner_check_reference.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check_reference.txt | tail -25
