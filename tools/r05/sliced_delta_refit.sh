#!/bin/bash
# round 5, after the lane-major dealing of DELTA runs: the pruned-NN layers as sliced DELTA plans (one slice per x sub-tile) against BITMAP / LIGHT,
# fixed point and float_pob -- the numbers behind the planner's rule in stream_tiles.cpp
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
for cfg in transformer_95 transformer_90 transformer_80 transformer_70 transformer_60 transformer_50; do
  for impl in fixed float_pob; do
    for spec in "bitmap:" "pairs:HISPARSE_LIGHT=1" "delta:HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5"; do
      fmt=${spec%%:*}; envs=${spec#*:}
      ( export HISPARSE_STREAM_FORMAT=$fmt; IFS=,; for kv in $envs; do export "$kv"; done
        timeout 300 python bench.py --config $cfg --impl $impl --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
          python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg/$impl $spec ->', d['config']['stream_format'], d['config']['col_slices'], 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), d['parity_vs_oracle'][:12])" )
    done
  done
done 2>&1 | tee gpurun_out/r05/sliced_delta_refit.txt
