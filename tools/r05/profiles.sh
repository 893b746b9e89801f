#!/bin/bash
# round 5: rocprofv3 --kernel-trace --stats + the two counter passes for every measured configuration (tools/profile_cfg.sh over the
# re-written tools/probe_cfg.py: spin-up, then PLAIN back-to-back launches -- no event pair around a launch)
cd "${GRAFT_REPO_ROOT:-.}"
for cfg in ogbl_ppa transformer_50 ogbn_products mouse_gene ogbl_ppa_rmat pokec hollywood gplus; do
  timeout 900 bash tools/profile_cfg.sh $cfg 200 > gpurun_out/prof_$cfg.log 2>&1
  grep -E "consistency|roofline_frac_rocprof|kernel_avg_us" gpurun_out/prof_$cfg/summary.txt | head -4
done
for cfg in transformer_80 transformer_95; do
  PROFILE_IMPL=fixed timeout 900 bash tools/profile_cfg.sh $cfg 200 > gpurun_out/prof_$cfg.log 2>&1
  grep -E "consistency|roofline_frac_rocprof|kernel_avg_us" gpurun_out/prof_$cfg/summary.txt | head -4
done
