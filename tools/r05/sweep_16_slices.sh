#!/bin/bash
# round 5: SWEEP images in up to 16 column slices (default) against at most 8 (forced): one rank's slab of an 8-way split, whole step
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/sweep_16_slices.txt; : > $out
for m in ogbn_products pokec; do
  timeout 600 python tools/slab_probe.py $m 8 "default:" "cs8:HISPARSE_COL_SLICES=8" "cs12:HISPARSE_COL_SLICES=12" "cs16:HISPARSE_COL_SLICES=16" 2>&1 | grep "way slab [036]" >> $out
done
timeout 600 python tools/slab_probe.py ogbn_products 4 "default:" 2>&1 | grep "way slab 0" >> $out
for i in fixed float_stall; do timeout 300 python tools/probe_cfg.py pokec $i 2>&1 | grep "^pokec .*step" | cut -c1-60,126-200 >> $out; done
cat $out
