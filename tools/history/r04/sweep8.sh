#!/bin/bash
# round 4: the planner's model-based choice between SWEEP and OWNER24: the slabs of ogbn-products, the synthetic squares, the named matrices
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
(RUNS=300 timeout 600 python tools/slab_probe.py ogbn_products 8 "default:" "owner24:HISPARSE_SWEEP=0" "sweep:HISPARSE_SWEEP=1" 2>&1 | grep -E "slab [0-7]:"
 RUNS=300 timeout 600 python tools/slab_probe.py ogbn_products 4 "default:" "owner24:HISPARSE_SWEEP=0" "sweep:HISPARSE_SWEEP=1" 2>&1 | grep -E "slab [0-1]:"
 RUNS=300 timeout 600 python tools/slab_probe.py ogbn_products 2 "default:" "owner24:HISPARSE_SWEEP=0" "sweep:HISPARSE_SWEEP=1" 2>&1 | grep -E "slab [0-1]:"
 RUNS=300 timeout 600 python tools/slab_probe.py pokec 8 "default:" "owner24:HISPARSE_SWEEP=0" "sweep:HISPARSE_SWEEP=1" 2>&1 | grep -E "slab [0-1]:"
) > gpurun_out/r04_sweep_choice_on_slabs.txt 2>&1
cat gpurun_out/r04_sweep_choice_on_slabs.txt | cut -c1-190
