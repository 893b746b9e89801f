#!/bin/bash
# column slices 2..8 (forced) on one configuration: bash tools/slices_probe.sh <config> [max_rows]
cfg=${1:-ogbn_products}
[ -n "${2:-}" ] && export HISPARSE_MAX_ROWS=$2
for cs in 2 3 4 5 6 7 8; do
  HISPARSE_COL_SLICES=$cs TAG="slices=$cs rows<=${2:-auto}" RUNS=20 timeout 120 python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
done
