#!/bin/bash
# stream ring depth 8 vs 16 (fixed point) per format: bash tools/depth_probe.sh <config> ...
for cfg in "$@"; do
  for fmt in pairs delta; do
    for d in 8 16; do
      HISPARSE_STREAM_FORMAT=$fmt HISPARSE_DEPTH=$d TAG="$fmt depth=$d" timeout 300 python tools/probe_cfg.py $cfg fixed 2>&1 | grep "kernel us"
    done
  done
done
