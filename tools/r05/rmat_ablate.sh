#!/bin/bash
# round 5: where the R-MAT stand-in's DELTA image loses against PAIRS (profiling build: HISPARSE_ABLATE 1 = no LDS adds, 2 = no x reads, 3 = neither)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/rmat_ablate.txt; : > $out
for fmt in pairs delta; do
  for ab in 0 1 2 3; do
    HISPARSE_ABLATE=$ab HISPARSE_STREAM_FORMAT=$fmt TAG="$fmt ablate=$ab" timeout 300 python tools/probe_cfg.py ogbl_ppa_rmat 2>&1 | grep -E "^ogbl_ppa_rmat +(pairs|delta)" >> $out
  done
done
cat $out
