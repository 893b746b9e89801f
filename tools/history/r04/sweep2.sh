#!/bin/bash
# round 4: the whole -m gpu suite with SWEEP in the parity / CSR / soak variants and as the planner's choice for pokec; then pokec's numbers
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -8
(for c in "pokec fixed" "pokec float_pob" "pokec float_stall" "ogbn_products float_stall"; do set -- $c; HISPARSE_PLAN_DEBUG=1 timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep -E "step us|load |sweep:"; done) > gpurun_out/r04_sweep_auto.txt 2>&1
cat gpurun_out/r04_sweep_auto.txt
