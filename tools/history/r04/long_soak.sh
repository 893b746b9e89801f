#!/bin/bash
# Round 4 soak: three fuzz profiles with fresh seeds (every case: 4 launches + the partition loop against the oracle, the device-built
# image against the host builder's; "auto" cases now take the LIGHT plan whenever they are small enough), then the repeated-run soak
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
: > gpurun_out/r04_long_soak.log
timeout 1500 python tests/gpu_fuzz_soak.py 700 401 2>&1 | tail -6 >> gpurun_out/r04_long_soak.log
FUZZ_PROFILE=large timeout 1500 python tests/gpu_fuzz_soak.py 200 402 2>&1 | tail -6 >> gpurun_out/r04_long_soak.log
FUZZ_PROFILE=dense timeout 1500 python tests/gpu_fuzz_soak.py 300 403 2>&1 | tail -6 >> gpurun_out/r04_long_soak.log
timeout 600 python tests/gpu_soak_ppa.py 2>&1 | tail -6 >> gpurun_out/r04_long_soak.log
cat gpurun_out/r04_long_soak.log
