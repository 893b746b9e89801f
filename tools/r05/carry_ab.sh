#!/bin/bash
# round 5: the combine pass carried into the next step's kernel (carry_combine) -- parity, then A/B per sliced configuration
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_carry.py -x -q 2>&1 | tail -8 | cut -c1-300 > gpurun_out/r05/carry_tests.txt
for cfg in ogbl_ppa gplus pokec ogbn_products ogbl_ppa_rmat hollywood; do
  for j in 0 1 0 1; do
    HISPARSE_CARRY_COMBINE=$j timeout 300 python bench.py --config $cfg --steps 500 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg carry=$j step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'frac_whole_step', d['roofline']['frac_whole_step'], 'frac', d['roofline']['frac'], 'sync_us', round(d['ms_per_step_synchronous']*1e3,2), 'graph_us', round((d.get('ms_per_step_graph_replay') or 0)*1e3,2))"
  done
done > gpurun_out/r05/carry_ab.txt 2>&1
cat gpurun_out/r05/carry_tests.txt gpurun_out/r05/carry_ab.txt
