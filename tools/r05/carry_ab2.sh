#!/bin/bash
# carry_combine A/B after the carried work moved behind the ring priming (same box for both settings)
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_carry.py -x -q 2>&1 | tail -3 | cut -c1-200
for cfg in ogbl_ppa hollywood ogbl_ppa_rmat gplus ogbn_products; do
  for j in 0 1 0 1; do
    HISPARSE_CARRY_COMBINE=$j timeout 300 python bench.py --config $cfg --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg carry=$j step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'sync_us', round(d['ms_per_step_synchronous']*1e3,2))"
  done
done 2>&1 | tee gpurun_out/r05/carry_ab2.txt
