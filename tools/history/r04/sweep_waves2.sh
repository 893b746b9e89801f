#!/bin/bash
# ... 4 wavefronts per block; 8 wavefronts with 16 in flight each
mkdir -p gpurun_out
(for round in 1 2; do
  for c in "pokec fixed" "pokec float_stall"; do set -- $c
    for lib in lib lib_w8 lib_w4 lib_w8d16; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so TAG="$lib" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"; done
  done
done
for lib in lib lib_w8 lib_w4 lib_w8d16; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so RUNS=300 timeout 300 python tools/slab_probe.py ogbn_products 8 "sweep:" 2>&1 | grep "slab 0:"; done
for lib in lib lib_w8 lib_w4; do HISPARSE_SWEEP=1 HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so TAG="$lib" timeout 300 python tools/probe_cfg.py ogbn_products float_stall 2>&1 | grep "step us"; done
) > gpurun_out/r04_sweep_waves2.txt 2>&1
cat gpurun_out/r04_sweep_waves2.txt | cut -c1-150
