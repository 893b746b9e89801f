#!/bin/bash
# round 3: the prologue change (loaders DMA sub-tile 0, consumers only zero) against the build before it, same box, alternating
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
for cfg in ogbl_ppa mouse_gene ogbl_ppa_rmat; do
  for round in 1 2 3; do
    TAG="new" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
    HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib/libhisparse_hip_before.so TAG="before" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
  done
done
for round in 1 2; do
    TAG="new" python tools/probe_cfg.py ogbn_products 2>&1 | grep "kernel us"
    HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib/libhisparse_hip_before.so TAG="before" python tools/probe_cfg.py ogbn_products 2>&1 | grep "kernel us"
done
timeout 600 python tools/rowblock_timeline.py ogbl_ppa 2>&1 | tail -16
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
} > gpurun_out/r03/prologue_ab.log 2>&1
cat gpurun_out/r03/prologue_ab.log
