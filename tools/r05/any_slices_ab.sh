#!/bin/bash
# round 5: column-slice counts restricted to 1, 2, 4, 8 for matrices of more than sixteen sub-tiles (HISPARSE_POW2_SLICES=1, rounds 1-4) against
# whatever the cost model likes (default since round 5): whole step, plain back-to-back launches, alternating, one box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/any_slices_ab.txt; : > $out
for cfg in ogbl_ppa ogbl_ppa_rmat hollywood; do
  for rep in 1 2 3; do
    for pow2 in 1 0; do
      if [ $pow2 = 1 ]; then export HISPARSE_POW2_SLICES=1; else unset HISPARSE_POW2_SLICES; fi
      TAG="pow2_slices=$pow2" timeout 300 python tools/probe_cfg.py $cfg 2>&1 | grep -E "^$cfg .*step" | cut -c1-60,126-205 >> $out
    done
  done
done
unset HISPARSE_POW2_SLICES
echo "# wide synthetic matrices (tools/r05/wide_slices.py: LIGHT off): auto = the model's choice, pow2 = the old rule, csN forced" >> $out
timeout 600 python tools/r05/wide_slices.py >> $out 2>&1
cat $out
