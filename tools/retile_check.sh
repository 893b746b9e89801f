#!/bin/bash
# GPU re-tile bring-up: byte comparison with the host builder, load time per configuration.
mkdir -p gpurun_out
if [ "${1:-}" != "quick" ]; then
  timeout 900 python -m pytest tests/test_gpu_retile.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/retile_test.log
  cat gpurun_out/retile_test.log
fi
for cfg in ogbl_ppa mouse_gene ogbn_products transformer_50; do
  HISPARSE_PLAN_DEBUG=1 RUNS=5 timeout 300 python tools/probe_cfg.py $cfg 2>&1 | grep -v "^plan" | tail -30 > gpurun_out/retile_probe_$cfg.log
  cat gpurun_out/retile_probe_$cfg.log
done
