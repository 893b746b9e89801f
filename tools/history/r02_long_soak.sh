#!/bin/bash
# Round 2 closing soak: three fuzz profiles with fresh seeds (every case: 4 launches + the partition loop against the oracle, the
# device-built image against the host builder's), then the repeated-run soak on the headline matrix.
mkdir -p gpurun_out
: > gpurun_out/r02_long_soak.log
timeout 1200 python tests/gpu_fuzz_soak.py 600 201 2>&1 | tail -6 >> gpurun_out/r02_long_soak.log
FUZZ_PROFILE=large timeout 1200 python tests/gpu_fuzz_soak.py 200 202 2>&1 | tail -6 >> gpurun_out/r02_long_soak.log
FUZZ_PROFILE=dense timeout 1200 python tests/gpu_fuzz_soak.py 200 203 2>&1 | tail -6 >> gpurun_out/r02_long_soak.log
timeout 600 python tests/gpu_soak_ppa.py 2>&1 | tail -6 >> gpurun_out/r02_long_soak.log
cat gpurun_out/r02_long_soak.log
