#!/bin/bash
# round 5: DELTA runs dealt to a wavefront's lanes "wave" (a contiguous 1/14 of the unit per wavefront) against "spread" (lane-major, like
# PAIRS' chunks): whole step, plain back-to-back launches, alternating
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/delta_deal.txt; : > $out
for cfg in ${CFGS:-ogbl_ppa_rmat ogbl_ppa hollywood mouse_gene gplus transformer_80}; do
  for rep in 1 2; do
    for deal in wave spread; do
      HISPARSE_STREAM_FORMAT=delta HISPARSE_DELTA_DEAL=$deal TAG="delta deal=$deal" timeout 300 python tools/probe_cfg.py $cfg 2>&1 | grep -E "^$cfg +delta" >> $out
    done
  done
done
cat $out
