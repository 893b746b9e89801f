#!/bin/bash
# round 4: where does the SWEEP kernel's time go on the real matrices?  Profiling builds: 1 = no LDS accumulation, 2 = no gather (one line near the chunk base), 3 = both
mkdir -p gpurun_out
export HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib/libhisparse_hip_prof.so
(for c in "pokec fixed" "pokec float_stall" "ogbn_products fixed" "ogbn_products float_stall"; do set -- $c
  for a in 0 1 2 3; do HISPARSE_SWEEP=1 HISPARSE_ABLATE=$a TAG="ablate=$a" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep -E "step us"; done
done) > gpurun_out/r04_sweep_ablations.txt 2>&1
cat gpurun_out/r04_sweep_ablations.txt | cut -c1-200
