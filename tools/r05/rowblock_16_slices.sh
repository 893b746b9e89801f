#!/bin/bash
# round 5: the row-block planner with more than eight column slices (forced: HISPARSE_COL_SLICES = 10 .. 16) on one rank's slab of an 8-way split
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/rowblock_16_slices.txt; : > $out
for m in hollywood mouse_gene; do
  timeout 500 python tools/slab_probe.py $m 8 "default:" "cs10:HISPARSE_COL_SLICES=10" "cs12:HISPARSE_COL_SLICES=12" "cs16:HISPARSE_COL_SLICES=16" "default2:" 2>&1 | grep "way slab [03]" >> $out
done
timeout 300 python tools/slab_probe.py hollywood 4 "default:" "cs12:HISPARSE_COL_SLICES=12" "cs16:HISPARSE_COL_SLICES=16" 2>&1 | grep "way slab 0" >> $out
cat $out
