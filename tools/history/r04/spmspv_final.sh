#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_spmspv.py -x -q 2>&1 | tail -2
(for c in ogbl_ppa mouse_gene pokec; do FRACS=0.0005,0.001,0.005,0.01,0.02,0.05,0.1 timeout 300 python tools/spmspv_probe.py $c; done) > gpurun_out/r04_spmspv_binned_columns.txt 2>&1
cat gpurun_out/r04_spmspv_binned_columns.txt | cut -c1-150
