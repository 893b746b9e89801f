#!/bin/bash
# round 4: SWEEP as it ships now (8 wavefronts per block, the fixed-point carry looked at one step late) against the same source with 16 wavefronts
mkdir -p gpurun_out
(for round in 1 2 3; do
  for c in "pokec fixed" "pokec float_stall" "pokec float_pob"; do set -- $c
    for lib in lib lib_w16; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so TAG="$lib" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"; done
  done
done
for lib in lib lib_w16; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so RUNS=300 timeout 300 python tools/slab_probe.py ogbn_products 8 "sweep:" 2>&1 | grep "slab 0:"; done
for lib in lib lib_w16; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so RUNS=300 timeout 300 python tools/slab_probe.py pokec 8 "sweep:" 2>&1 | grep "slab 0:"; done
) > gpurun_out/r04_sweep_waves3.txt 2>&1
cat gpurun_out/r04_sweep_waves3.txt | cut -c1-150
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_retile.py -k "sweep" -x -q 2>&1 | tail -3
