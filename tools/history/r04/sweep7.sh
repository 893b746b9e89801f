#!/bin/bash
# round 4: would SWEEP in ONE slice (one launch, no combine pass) beat the sliced row-block plans on the strong-scaling slabs?
mkdir -p gpurun_out
(RUNS=300 timeout 600 python tools/slab_probe.py mouse_gene 8 "default:" "sweep-1-slice:HISPARSE_SWEEP=1,HISPARSE_COL_SLICES=1" "sweep-auto:HISPARSE_SWEEP=1" 2>&1 | grep -E "slab [0-2]:"
 RUNS=300 timeout 600 python tools/slab_probe.py mouse_gene 4 "default:" "sweep-1-slice:HISPARSE_SWEEP=1,HISPARSE_COL_SLICES=1" 2>&1 | grep -E "slab [0-1]:"
 RUNS=300 timeout 600 python tools/slab_probe.py hollywood 8 "default:" "sweep-1-slice:HISPARSE_SWEEP=1,HISPARSE_COL_SLICES=1" "sweep-auto:HISPARSE_SWEEP=1" 2>&1 | grep -E "slab [0-1]:"
 RUNS=300 timeout 600 python tools/slab_probe.py ogbn_products 8 "default:" "sweep-1-slice:HISPARSE_SWEEP=1,HISPARSE_COL_SLICES=1" "sweep-auto:HISPARSE_SWEEP=1" 2>&1 | grep -E "slab [0-1]:"
) > gpurun_out/r04_sweep_on_slabs.txt 2>&1
cat gpurun_out/r04_sweep_on_slabs.txt | cut -c1-200
