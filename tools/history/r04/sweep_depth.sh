#!/bin/bash
# round 4: SWEEP with 12 / 16 chunks and gathers in flight per wavefront instead of 8 (libraries built with -DHS_SWEEP_DEPTH=12|16), alternating on one box
mkdir -p gpurun_out
(for round in 1 2; do
  for c in "pokec fixed" "pokec float_stall"; do set -- $c
    TAG="depth 8" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"
    HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib_d12/libhisparse_hip.so TAG="depth 12" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"
    HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib_d16/libhisparse_hip.so TAG="depth 16" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"
  done
done
for lib in lib lib_d12 lib_d16; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so RUNS=300 timeout 300 python tools/slab_probe.py ogbn_products 8 "sweep:" 2>&1 | grep "slab 0:"; done
HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib_d16/libhisparse_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py -k sweep -x -q 2>&1 | tail -2
) > gpurun_out/r04_sweep_ring_depth.txt 2>&1
cat gpurun_out/r04_sweep_ring_depth.txt | cut -c1-175
