#!/bin/bash
# round 5: the pruned-NN layers as a sliced DELTA plan (a slice per x sub-tile, combine carried: ONE launch) against the planner's BITMAP / LIGHT
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
for cfg in transformer_95 transformer_90 transformer_80 transformer_70 transformer_60 transformer_50; do
  for impl in fixed float_pob; do
    for spec in "auto:" "delta:HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5" "delta:HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5,HISPARSE_CARRY_COMBINE=0" "delta:HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=3"; do
      fmt=${spec%%:*}; envs=${spec#*:}
      ( [ "$fmt" != auto ] && export HISPARSE_STREAM_FORMAT=$fmt; IFS=,; for kv in $envs; do export "$kv"; done
        timeout 300 python bench.py --config $cfg --impl $impl --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
          python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg/$impl $spec ->', d['config']['stream_format'], d['config']['col_slices'], 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'frac_step', d['roofline']['frac_whole_step'], d['parity_vs_oracle'][:12])" )
    done
  done
done 2>&1 | tee gpurun_out/r05/t_delta5.txt
