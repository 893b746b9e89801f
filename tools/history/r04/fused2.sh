#!/bin/bash
mkdir -p gpurun_out
for cfg in ogbl_ppa pokec; do
  IMPL=fixed RUNS=200 ROUNDS=3 python tools/probe_variants.py $cfg "fused:" "placement-only:HISPARSE_FUSED_COMBINE=2" "separate:HISPARSE_FUSED_COMBINE=0"
done 2>&1 | tee gpurun_out/r04_fused_combine_placement.txt
