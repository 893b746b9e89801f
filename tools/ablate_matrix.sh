#!/bin/bash
# where the kernel time goes: the profiling builds (HISPARSE_ABLATE) of one fixed-point configuration, per format
# bash tools/ablate_matrix.sh <config> "<format> <row_runs>" ...
cfg=$1; shift
for spec in "$@"; do
  set -- $spec; fmt=$1; runs=${2:-}
  for ab in 0 3 4 8 15 31 47 127; do
    HISPARSE_STREAM_FORMAT=$fmt HISPARSE_ROW_RUNS=$runs HISPARSE_ABLATE=$ab TAG="$fmt runs=$runs" timeout 300 python tools/probe_cfg.py $cfg fixed 2>&1 | grep "kernel us"
  done
done
