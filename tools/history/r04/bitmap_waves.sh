#!/bin/bash
# round 4: the BITMAP kernel with 8 / 12 wavefronts per workgroup instead of 16 (a 1024-thread workgroup's wavefronts start 2-4 us apart; SWEEP gained 4 % from 512 threads)
mkdir -p gpurun_out
(for round in 1 2; do
  for c in "transformer_50 float_pob" "transformer_50 fixed" "transformer_80 fixed" "transformer_70 float_pob"; do set -- $c
    for lib in lib lib_bw12 lib_bw8; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so TAG="$lib" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us"; done
  done
done
for lib in lib_bw12 lib_bw8; do HISPARSE_HIP_LIB=$PWD/hisparse_amd/$lib/libhisparse_hip.so timeout 900 python -m pytest tests/test_gpu_bitmap.py tests/test_spmm.py -x -q 2>&1 | tail -2; done
) > gpurun_out/r04_bitmap_waves.txt 2>&1
cat gpurun_out/r04_bitmap_waves.txt | cut -c1-150
