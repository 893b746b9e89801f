#!/bin/bash
# carry_combine A/B at the driver's short run (--steps 20 --warmup 5) and at a long one, same box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
for cfg in ogbl_ppa hollywood ogbl_ppa_rmat; do
  for st in "20 5" "500 50"; do
    set -- $st
    for j in 0 1 0 1; do
      HISPARSE_CARRY_COMBINE=$j timeout 300 python bench.py --config $cfg --steps $1 --warmup $2 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg steps=$1 carry=$j step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'graph_us', round((d.get('ms_per_step_graph_replay') or 0)*1e3,2), 'sync_us', round(d['ms_per_step_synchronous']*1e3,2))"
    done
  done
done 2>&1 | tee gpurun_out/r05/carry_ab_steps.txt
