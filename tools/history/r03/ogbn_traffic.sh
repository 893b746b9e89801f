#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
timeout 900 bash tools/profile_cfg.sh ogbn_products 30 > /dev/null 2>&1
cp gpurun_out/prof_ogbn_products/summary.txt gpurun_out/r03/ogbn_affine_summary.txt
HISPARSE_XCD_AFFINITY=0 timeout 900 bash tools/profile_cfg.sh ogbn_products 30 > /dev/null 2>&1
cp gpurun_out/prof_ogbn_products/summary.txt gpurun_out/r03/ogbn_spread_summary.txt
HISPARSE_STREAM_FORMAT=owner timeout 900 bash tools/profile_cfg.sh ogbn_products 30 > /dev/null 2>&1
cp gpurun_out/prof_ogbn_products/summary.txt gpurun_out/r03/ogbn_owner8_summary.txt
grep -h "kernel_avg_us\|hbm_bytes_per_launch\|hbm_read\|hbm_write\|stream_bytes\|roofline_frac" gpurun_out/r03/ogbn_*_summary.txt
