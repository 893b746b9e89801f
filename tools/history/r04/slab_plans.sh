#!/bin/bash
# round 4: single-launch plans for the strong-scaling slabs of mouse_gene (one slice = no combine pass) in DELTA with per-lane sums
mkdir -p gpurun_out
(for n in 8 4; do RUNS=300 timeout 600 python tools/slab_probe.py mouse_gene $n "default:" "delta-1:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=1" "delta-1-sums:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=1,HISPARSE_ROW_RUNS=1" "delta-2:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=2" "delta-3:HISPARSE_STREAM_FORMAT=delta,HISPARSE_COL_SLICES=3" "pairs-1:HISPARSE_STREAM_FORMAT=pairs,HISPARSE_COL_SLICES=1" 2>&1 | grep -E "slab [0-1]:"; done) > gpurun_out/r04_slab_single_launch_plans.txt 2>&1
cat gpurun_out/r04_slab_single_launch_plans.txt | cut -c1-170
