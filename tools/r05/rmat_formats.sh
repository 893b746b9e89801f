#!/bin/bash
# round 5: what a per-block DELTA / PAIRS mix could buy on the heavy-tailed stand-ins -- the whole image in either format, same plan otherwise
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/rmat_formats.txt; : > $out
for cfg in ogbl_ppa_rmat gplus; do
  for fmt in pairs delta; do
    HISPARSE_PLAN_DEBUG=1 HISPARSE_STREAM_FORMAT=$fmt TAG=$fmt timeout 300 python tools/probe_cfg.py $cfg 2>&1 | grep -E "^$cfg|bridge|DELTA|delta" | tail -6 >> $out
  done
done
cat $out
