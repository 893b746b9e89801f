#!/bin/bash
# GPU re-tile bring-up: byte comparison with the host builder, load time on the headline matrix.
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_retile.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/retile_test.log
cat gpurun_out/retile_test.log
for cfg in ogbl_ppa mouse_gene ogbn_products; do
  HISPARSE_PLAN_DEBUG=1 timeout 300 python tools/probe_cfg.py $cfg 2>&1 | tail -30 > gpurun_out/retile_probe_$cfg.log
  HISPARSE_RETILE=host HISPARSE_PLAN_DEBUG=1 timeout 300 python tools/probe_cfg.py $cfg 2>&1 | tail -30 > gpurun_out/retile_probe_${cfg}_host.log
done
grep -h "load\|phase\|retile" gpurun_out/retile_probe_*.log | head -80
