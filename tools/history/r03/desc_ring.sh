#!/bin/bash
# round 3: LDS descriptor ring (no scalar loads in the unit loop) against the build before it, same box
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
BEFORE=$PWD/hisparse_amd/lib/libhisparse_hip_before.so
{
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
for cfg in mouse_gene_slab8 pokec ogbl_ppa mouse_gene; do
  for round in 1 2; do
    TAG="new" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
    HISPARSE_HIP_LIB=$BEFORE TAG="before" python tools/probe_cfg.py $cfg 2>&1 | grep "kernel us"
  done
done
TAG="new" python tools/probe_cfg.py ogbn_products 2>&1 | grep "kernel us"
HISPARSE_HIP_LIB=$BEFORE TAG="before" python tools/probe_cfg.py ogbn_products 2>&1 | grep "kernel us"
TAG="new" python tools/probe_cfg.py ogbn_products 2>&1 | grep "kernel us"
timeout 600 python tests/gpu_fuzz_soak.py 150 21 2>&1 | tail -4
FUZZ_PROFILE=large timeout 900 python tests/gpu_fuzz_soak.py 40 22 2>&1 | tail -4
} > gpurun_out/r03/desc_ring.log 2>&1
cat gpurun_out/r03/desc_ring.log
