#!/bin/bash
# the BITMAP builder on the device: byte comparison with the host builder, parity, load times
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1200 python -m pytest tests/test_gpu_retile.py -x -q -m gpu -k "bitmap or host_opt_out" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_load_csr.py tests/test_spmm.py -x -q -m gpu 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bitmap" 2>&1 | tail -3
for w in gpu host; do
  echo "== transformer_50 float_pob, BITMAP build on the $w"
  HISPARSE_BITMAP_BUILD=$w timeout 600 python tools/load_time.py transformer_50 2>&1 | tail -4
  echo "== transformer_50 fixed, BITMAP build on the $w"
  IMPL=fixed HISPARSE_BITMAP_BUILD=$w timeout 600 python tools/load_time.py transformer_50 2>&1 | tail -4
done
HISPARSE_PLAN_DEBUG=1 timeout 600 python tools/load_time.py transformer_50 2>&1 | grep -i "bitmap\|load:" | tail -12
} > gpurun_out/r03/bitmap_gpu_build.log 2>&1
cat gpurun_out/r03/bitmap_gpu_build.log
