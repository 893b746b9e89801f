#!/bin/bash
# round 4: SWEEP images built on the device: byte for byte against the host builder, the fallbacks, load times
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_retile.py tests/test_gpu_sweep.py tests/test_gpu_load_csr.py -x -q 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py -k sweep -x -q 2>&1 | tail -3
(for c in "pokec fixed" "pokec float_stall"; do set -- $c
  HISPARSE_PLAN_DEBUG=1 timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep -E "step us|load |sweep|gpu:"
  HISPARSE_RETILE=host timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep -E "load "
done
HISPARSE_SWEEP=1 timeout 300 python tools/probe_cfg.py ogbn_products float_stall 2>&1 | grep -E "step us|load ") > gpurun_out/r04_sweep_gpu_builder.txt 2>&1
cat gpurun_out/r04_sweep_gpu_builder.txt | cut -c1-220
