#!/bin/bash
# ... and with 4 / 6 in flight
mkdir -p gpurun_out
(for round in 1 2; do
  for c in "pokec fixed" "pokec float_stall"; do set -- $c
    for lib in lib_d4 lib_d6 lib lib_d16; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so TAG="$lib" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"; done
  done
done
for lib in lib_d4 lib_d6 lib; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so RUNS=300 timeout 300 python tools/slab_probe.py ogbn_products 8 "sweep:" 2>&1 | grep "slab 0:"; done
HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib_d6/libhisparse_hip.so timeout 600 python -m pytest tests/test_gpu_parity.py -k sweep -x -q 2>&1 | tail -2
) > gpurun_out/r04_sweep_ring_depth2.txt 2>&1
cat gpurun_out/r04_sweep_ring_depth2.txt | cut -c1-150
