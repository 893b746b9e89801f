#!/bin/bash
# round 4: how many x entries should a workgroup of the SpMSpV expand kernel take?  (workgroups = entries / columns; today: >= 512 workgroups before they grow)
# (HISPARSE_SPMSPV_DIVISOR was read by an experimental build of launch_spmspv only -- columns = entries / divisor, capped at 64; the rule that came out of this run is in spmspv.hip)
mkdir -p gpurun_out
(for d in 512 256 128 64 32 16; do echo "== at least $d workgroups before the columns per workgroup grow"; for c in ogbl_ppa mouse_gene pokec; do HISPARSE_SPMSPV_DIVISOR=$d FRACS=0.0005,0.001,0.005,0.01,0.05 timeout 200 python tools/spmspv_probe.py $c 2>&1 | grep "%" | cut -c1-75; done; done) > gpurun_out/r04_spmspv_expand_columns.txt 2>&1
cat gpurun_out/r04_spmspv_expand_columns.txt
