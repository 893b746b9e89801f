#!/bin/bash
# round 4: the LIGHT plan -- parity first, then timings against the row-block plans on the small matrices (same box, same process per config)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_light.py tests/test_gpu_options.py -x -q 2>&1 | tail -15
python -m pytest tests/test_gpu_retile.py -x -q -k "duplicate" 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -k "light" 2>&1 | tail -5
for cfg in transformer_95 transformer_90 mouse_gene_slab8 mouse_gene_slab4 csim_1k ppa_small gplus; do
  RUNS=300 ROUNDS=3 IMPL=fixed python tools/probe_variants.py $cfg "auto:" "light:HISPARSE_LIGHT=1" "rowblock:HISPARSE_LIGHT=0"
done 2>&1 | tee gpurun_out/r04_light_first.txt
