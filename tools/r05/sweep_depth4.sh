#!/bin/bash
# round 5: spmv_sweep_kernel with four chunks / gathers in flight per wavefront (the default from here on) against eight (libhisparse_hip_d8.so:
# make variant NAME=d8 DEFS=-DHS_SWEEP_DEPTH=8): parity (GPU sweep tests, a SWEEP-only soak), then kernel alone and whole step on the SWEEP users
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
out=gpurun_out/r05/sweep_depth4.txt; : > $out
timeout 500 python -m pytest tests/test_gpu_sweep.py tests/test_spmm.py -m gpu -q -x 2>&1 | tail -2 | tee -a $out
( HISPARSE_STREAM_FORMAT_ONLY=sweep timeout 600 python tests/gpu_fuzz_soak.py 250 5701 | tail -2
  HISPARSE_STREAM_FORMAT_ONLY=sweep FUZZ_PROFILE=large timeout 600 python tests/gpu_fuzz_soak.py 100 5702 | tail -2 ) 2>&1 | tee -a $out
D8=$PWD/hisparse_amd/lib/libhisparse_hip_d8.so
for i in fixed float_pob float_stall; do
  for lib in "" $D8 "" $D8; do
    echo -n "pokec $i depth $([ -z "$lib" ] && echo 4 || echo 8): " >> $out
    HISPARSE_HIP_LIB=${lib:-$PWD/hisparse_amd/lib/libhisparse_hip.so} timeout 200 python tools/probe_cfg.py pokec $i 2>&1 | grep "step us" | cut -c42-130 >> $out
  done
done
for lib in "" $D8; do
  echo "== ogbn_products 8-way slabs, depth $([ -z "$lib" ] && echo 4 || echo 8)" >> $out
  HISPARSE_HIP_LIB=${lib:-$PWD/hisparse_amd/lib/libhisparse_hip.so} timeout 400 python tools/slab_probe.py ogbn_products 8 "default:" 2>&1 | grep "way slab [036]" >> $out
done
cat $out
