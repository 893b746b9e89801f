#!/bin/bash
# round 4: SWEEP with 4-byte fixed-point accumulators (wrapping sum + carry bit): parity, pokec / ogbn-products, and the crossover in fixed point
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -k sweep -x -q 2>&1 | tail -3
(for c in "pokec fixed" "pokec float_pob" "ogbn_products fixed"; do set -- $c; timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep -E "step us|load "; done
 HISPARSE_SWEEP=1 timeout 300 python tools/probe_cfg.py ogbn_products fixed 2>&1 | grep -E "step us|load "
 HISPARSE_SWEEP=1 timeout 300 python tools/probe_cfg.py ogbn_products float_stall 2>&1 | grep -E "step us|load "
 GAPS=15000,25000,35000,50000,70000 SIZES=1000000,1600000,2400000 timeout 900 python tools/probe_sweep.py fixed) > gpurun_out/r04_sweep_narrow.txt 2>&1
cat gpurun_out/r04_sweep_narrow.txt | cut -c1-250
