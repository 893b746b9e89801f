#!/bin/bash
# what would OWNER24 gain on ogbn-products without its unit barriers (upper bound: HISPARSE_ABLATE=8, wrong results), without refills (4), both (12)?
mkdir -p gpurun_out
export HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib/libhisparse_hip_prof.so
(for impl in float_stall fixed; do for a in 0 8 4 12 1 13; do HISPARSE_ABLATE=$a TAG="ablate=$a" timeout 300 python tools/probe_cfg.py ogbn_products $impl 2>&1 | grep "step us"; done; done) > gpurun_out/r04_owner24_ablations_ogbn.txt 2>&1
cat gpurun_out/r04_owner24_ablations_ogbn.txt | cut -c1-170
