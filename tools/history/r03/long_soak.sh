#!/bin/bash
# Round 3 closing soak: three fuzz profiles with fresh seeds (every case: 4 launches + the partition loop against the oracle, the
# device-built image -- BITMAP included since this round -- against the host builder's), then the repeated-run soak on the headline matrix.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/r03_long_soak.log
timeout 1500 python tests/gpu_fuzz_soak.py 700 301 2>&1 | tail -6 >> gpurun_out/r03_long_soak.log
FUZZ_PROFILE=large timeout 1500 python tests/gpu_fuzz_soak.py 200 302 2>&1 | tail -6 >> gpurun_out/r03_long_soak.log
FUZZ_PROFILE=dense timeout 1500 python tests/gpu_fuzz_soak.py 300 303 2>&1 | tail -6 >> gpurun_out/r03_long_soak.log
timeout 600 python tests/gpu_soak_ppa.py 2>&1 | tail -6 >> gpurun_out/r03_long_soak.log
cat gpurun_out/r03_long_soak.log
