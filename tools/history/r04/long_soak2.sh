#!/bin/bash
# Round 4 closing soak on the final library (SWEEP among the forced formats and the planner's choices, images built on the device compared with
# the host builder's): three fuzz profiles with fresh seeds, then the repeated-run soak
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/r04_long_soak2.log
timeout 1500 python tests/gpu_fuzz_soak.py 700 411 2>&1 | tail -6 >> gpurun_out/r04_long_soak2.log
FUZZ_PROFILE=large timeout 1500 python tests/gpu_fuzz_soak.py 250 412 2>&1 | tail -6 >> gpurun_out/r04_long_soak2.log
FUZZ_PROFILE=dense timeout 1500 python tests/gpu_fuzz_soak.py 300 413 2>&1 | tail -6 >> gpurun_out/r04_long_soak2.log
HISPARSE_STREAM_FORMAT_ONLY=sweep timeout 1500 python tests/gpu_fuzz_soak.py 300 414 2>&1 | tail -6 >> gpurun_out/r04_long_soak2.log
timeout 600 python tests/gpu_soak_ppa.py 2>&1 | tail -6 >> gpurun_out/r04_long_soak2.log
cat gpurun_out/r04_long_soak2.log
