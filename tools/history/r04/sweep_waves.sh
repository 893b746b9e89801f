#!/bin/bash
# round 4: SWEEP with 512-thread workgroups (8 wavefronts per block, still one block per CU) against 1024: does a shorter launch ramp pay?
mkdir -p gpurun_out
(for round in 1 2; do
  for c in "pokec fixed" "pokec float_stall"; do set -- $c
    for lib in lib lib_w8; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so TAG="$lib" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"; done
  done
done
for lib in lib lib_w8; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so RUNS=300 timeout 300 python tools/slab_probe.py ogbn_products 8 "sweep:" 2>&1 | grep "slab 0:"; done
for lib in lib lib_w8; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so RUNS=300 timeout 300 python tools/slab_probe.py pokec 8 "sweep:" 2>&1 | grep "slab 0:"; done
) > gpurun_out/r04_sweep_waves.txt 2>&1
cat gpurun_out/r04_sweep_waves.txt | cut -c1-150
