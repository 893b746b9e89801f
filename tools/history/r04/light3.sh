#!/bin/bash
# round 4: would the LIGHT plan beat BITMAP on the pruned-NN layers above its 2 M non-zero limit? (long rows, ascending columns: cheap gathers)
mkdir -p gpurun_out
for cfg in transformer_50 transformer_60 transformer_70 transformer_80 transformer_90; do
  for impl in fixed float_pob; do
    IMPL=$impl RUNS=300 ROUNDS=3 python tools/probe_variants.py $cfg "auto-$impl:" "light-$impl:HISPARSE_STREAM_FORMAT=pairs,HISPARSE_LIGHT=1"
  done
done 2>&1 | tee gpurun_out/r04_light_vs_bitmap.txt
for cfg in transformer_70 transformer_80; do HISPARSE_STREAM_FORMAT=pairs HISPARSE_LIGHT=1 python tools/probe_cfg.py $cfg fixed | head -1; python tools/probe_cfg.py $cfg fixed | head -1; done 2>&1 | tee -a gpurun_out/r04_light_vs_bitmap.txt
