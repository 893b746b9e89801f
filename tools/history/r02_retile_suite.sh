#!/bin/bash
# Round 2, after the GPU re-tile became the default: the whole GPU suite, the fuzz soak (which now also compares the device-built
# image with the host builder's), and the load time of every configuration (cold first load, warm reloads).
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02_gpu_suite.log
cat gpurun_out/r02_gpu_suite.log
timeout 600 python tests/gpu_fuzz_soak.py 150 7 2>&1 | tail -8 > gpurun_out/r02_fuzz.log
FUZZ_PROFILE=large timeout 600 python tests/gpu_fuzz_soak.py 40 8 2>&1 | tail -8 >> gpurun_out/r02_fuzz.log
FUZZ_PROFILE=dense timeout 600 python tests/gpu_fuzz_soak.py 60 9 2>&1 | tail -8 >> gpurun_out/r02_fuzz.log
cat gpurun_out/r02_fuzz.log
: > gpurun_out/r02_load_times.txt
for cfg in ogbl_ppa ogbl_ppa_rmat mouse_gene ogbn_products transformer_50; do
  timeout 300 python tools/load_time.py $cfg 2>/dev/null >> gpurun_out/r02_load_times.txt
  HISPARSE_RETILE=host timeout 300 python tools/load_time.py $cfg 2>/dev/null | head -2 >> gpurun_out/r02_load_times.txt
done
cat gpurun_out/r02_load_times.txt
