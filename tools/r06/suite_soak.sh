#!/bin/bash
# round 6: the whole GPU suite on the final sources, then a soak with fresh seeds (planner with the tile census, row-block stream policies)
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06/gpu_suite.txt
tail -8 gpurun_out/r06/gpu_suite.txt
( timeout 2400 python tests/gpu_fuzz_soak.py 500 6601 | tail -12
  FUZZ_PROFILE=dense timeout 1200 python tests/gpu_fuzz_soak.py 150 6602 | tail -8
  FUZZ_PROFILE=large timeout 2400 python tests/gpu_fuzz_soak.py 200 6603 | tail -8 ) 2>&1 | tee gpurun_out/r06/long_soak.log
