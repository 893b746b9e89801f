#!/bin/bash
# round 3: rocprofv3 --kernel-trace --stats + the two counter passes for every measured configuration (tools/profile_cfg.sh)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in ogbl_ppa transformer_50 ogbn_products mouse_gene ogbl_ppa_rmat pokec hollywood gplus; do
  timeout 600 bash tools/profile_cfg.sh $cfg 30 > gpurun_out/prof_$cfg.log 2>&1
  tail -12 gpurun_out/prof_$cfg/summary.txt | head -3
done
# the reference sweep's matrices under 50 % of the roofline that are not in the list above, in the sweep's numeric mode (fixed point)
for cfg in transformer_80 transformer_95; do
  PROFILE_IMPL=fixed timeout 600 bash tools/profile_cfg.sh $cfg 30 > gpurun_out/prof_$cfg.log 2>&1
  tail -12 gpurun_out/prof_$cfg/summary.txt | head -3
done
timeout 600 python -m pytest tests/test_benchmark_cli.py -q -m gpu 2>&1 | tail -3
