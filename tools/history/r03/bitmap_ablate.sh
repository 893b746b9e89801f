#!/bin/bash
# ablation builds of spmv_bitmap_kernel (HISPARSE_ABLATE bits: 1 no value loads, 2 no x loads, 4 no arithmetic, 8 no run, 16 no row sums)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
for m in transformer_50 transformer_80; do
ROUNDS=4 timeout 900 python tools/probe_variants.py $m "full:" "no-values:HISPARSE_ABLATE=1" "no-x:HISPARSE_ABLATE=2" "no-loads:HISPARSE_ABLATE=3" "nothing:HISPARSE_ABLATE=7" "no-run:HISPARSE_ABLATE=15" 2>&1 | tail -6
done
} > gpurun_out/r03/bitmap_ablate2.log 2>&1
cat gpurun_out/r03/bitmap_ablate2.log
