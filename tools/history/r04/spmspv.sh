#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_spmspv.py tests/test_gpu_memory.py -x -q 2>&1 | tail -8
for cfg in ogbl_ppa mouse_gene pokec; do python tools/spmspv_probe.py $cfg; done 2>&1 | tee gpurun_out/r04_spmspv.txt
for c in "transformer_80 float_stall" "transformer_80 float_pob" "pokec float_pob" "pokec float_stall" "ogbn_products float_pob" "transformer_50 float_stall"; do python tools/probe_cfg.py $c 2>&1 | head -1; done | tee gpurun_out/r04_float_modes_after_planner_fixes.txt
python -m pytest tests/test_gpu_retile.py tests/test_gpu_bitmap.py -x -q 2>&1 | tail -3
