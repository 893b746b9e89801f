#!/bin/bash
# kernel time of one configuration under every stream format that can hold it: bash tools/format_matrix.sh <config> [impl]
cfg=${1:-mouse_gene}; impl=${2:-}
for fmt in pairs delta owner bitmap; do
  for runs in "" 0 1; do
    [ "$fmt" != delta ] && [ -n "$runs" ] && continue
    HISPARSE_STREAM_FORMAT=$fmt HISPARSE_ROW_RUNS=$runs TAG="$fmt runs=$runs" timeout 300 python tools/probe_cfg.py $cfg $impl 2>&1 | grep "kernel us"
  done
done
