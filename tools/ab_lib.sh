#!/bin/bash
# A/B of two builds of libhisparse_hip.so on the same box: bash tools/ab_lib.sh <other .so> <config> ...   (alternating, 3 rounds)
other=$1; shift
for cfg in "$@"; do
  for round in 1 2 3; do
    TAG="current" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
    HISPARSE_HIP_LIB=$other TAG="other" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
  done
done
