#!/bin/bash
# round 5 closing soak on the final sources: fresh seeds, the three profiles; every case: 4 single launches, bursts of 2 and 3 (the carried
# combine), the partition loop (row blocks that cross partition borders), the device-built image against the host builder's
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
( timeout 3000 python tests/gpu_fuzz_soak.py 500 5601 | tail -12
  FUZZ_PROFILE=dense timeout 1200 python tests/gpu_fuzz_soak.py 150 5602 | tail -8
  FUZZ_PROFILE=large timeout 2400 python tests/gpu_fuzz_soak.py 200 5603 | tail -8 ) 2>&1 | tee gpurun_out/r05/long_soak6.log
