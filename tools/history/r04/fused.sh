#!/bin/bash
# round 4: fused slice combine (the last block of a row range adds the slices; all slices of a range on one XCD) against the separate combine kernel
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "slices or partition or medium" 2>&1 | tail -4
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_retile.py -x -q 2>&1 | tail -4
for cfg in ogbl_ppa gplus hollywood mouse_gene_slab8 mouse_gene_slab4 pokec ogbl_ppa_rmat; do
  IMPL=fixed RUNS=200 ROUNDS=3 python tools/probe_variants.py $cfg "fused:" "separate:HISPARSE_FUSED_COMBINE=0"
done 2>&1 | tee gpurun_out/r04_fused_combine.txt
IMPL=float_stall RUNS=200 ROUNDS=3 python tools/probe_variants.py ogbn_products "fused:" "separate:HISPARSE_FUSED_COMBINE=0" 2>&1 | tee -a gpurun_out/r04_fused_combine.txt
for cfg in ogbl_ppa mouse_gene_slab8 gplus; do python tools/probe_cfg.py $cfg fixed | head -1; HISPARSE_FUSED_COMBINE=0 python tools/probe_cfg.py $cfg fixed | head -1; done 2>&1 | tee -a gpurun_out/r04_fused_combine.txt
