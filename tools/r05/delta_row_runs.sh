#!/bin/bash
# round 5: DELTA with spread dealing: per-lane register sums along a run (HISPARSE_ROW_RUNS=1, the dense-row path) for every block against none
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/delta_row_runs.txt; : > $out
for cfg in ${CFGS:-ogbl_ppa_rmat ogbl_ppa gplus hollywood}; do
  for rep in 1 2; do
    for rr in auto 0 1; do
      if [ $rr = auto ]; then unset HISPARSE_ROW_RUNS; else export HISPARSE_ROW_RUNS=$rr; fi
      HISPARSE_STREAM_FORMAT=delta HISPARSE_DELTA_DEAL=spread TAG="delta spread row_runs=$rr" timeout 300 python tools/probe_cfg.py $cfg 2>&1 | grep -E "^$cfg +delta" | cut -c1-150 >> $out
    done
  done
done
cat $out
