#!/bin/bash
# round 5: hs_spmm over element-stream (graph) images -- four columns per pass over a SWEEP image planned for it (spmm_vectors = 4) against k SpMVs
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_spmm.py -x -q 2>&1 | tail -6 | cut -c1-300
for cfg in ogbl_ppa pokec mouse_gene; do
  for k in 16 8; do
    timeout 300 python tools/spmm_probe.py $cfg $k 2>&1 | tail -1
    HISPARSE_SPMM_VECTORS=4 timeout 300 python tools/spmm_probe.py $cfg $k 2>&1 | tail -1 | sed 's/$/   [spmm_vectors=4]/'
  done
done 2>&1 | tee gpurun_out/r05/spmm_sweep.txt
