#!/bin/bash
# round 4, VERDICT item 2: the SWEEP format (x gathered from L2, column-ordered blocks) on the real matrices, against the planner's choice
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -k sweep -x -q 2>&1 | tail -6
(for c in "pokec fixed" "pokec float_pob" "ogbn_products float_stall" "ogbn_products fixed" "ogbl_ppa fixed" "hollywood fixed" "gplus fixed"; do
  set -- $c
  echo "== $1 $2"
  timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | head -1
  HISPARSE_STREAM_FORMAT=sweep HISPARSE_PLAN_DEBUG=1 timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep -E "sweep|step us|load" | head -24
done) > gpurun_out/r04_sweep.txt 2>&1
cat gpurun_out/r04_sweep.txt
