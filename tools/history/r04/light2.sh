#!/bin/bash
# round 4: LIGHT plan, workgroups per CU (blocks = WGS x 256), same box
mkdir -p gpurun_out
python -m pytest tests/test_gpu_light.py -x -q 2>&1 | tail -3
for cfg in transformer_95 transformer_90 mouse_gene_slab8 ppa_small csim_1k; do
  RUNS=300 ROUNDS=3 IMPL=fixed python tools/probe_variants.py $cfg "rowblock:HISPARSE_LIGHT=0" "light4:HISPARSE_LIGHT=1,HISPARSE_LIGHT_WGS=4" "light6:HISPARSE_LIGHT=1,HISPARSE_LIGHT_WGS=6" "light2:HISPARSE_LIGHT=1,HISPARSE_LIGHT_WGS=2"
done 2>&1 | tee gpurun_out/r04_light_wgs.txt
for cfg in transformer_95 mouse_gene_slab8; do HISPARSE_LIGHT=1 python tools/probe_cfg.py $cfg fixed; HISPARSE_LIGHT=0 python tools/probe_cfg.py $cfg fixed; done 2>&1 | tee -a gpurun_out/r04_light_wgs.txt
