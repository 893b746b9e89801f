#!/bin/bash
# round 5: transformer-70 / -80 (30 % / 20 % dense pruned-NN layers) under every element-stream format and slice count, against the planner's BITMAP
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
for cfg in transformer_80 transformer_70; do
  for spec in "bitmap:" "delta:HISPARSE_LIGHT=0" "delta:HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5" "pairs:HISPARSE_LIGHT=0" "pairs:HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5" "delta:HISPARSE_LIGHT=0,HISPARSE_ROW_RUNS=1"; do
    fmt=${spec%%:*}; envs=${spec#*:}
    ( export HISPARSE_STREAM_FORMAT=$fmt; IFS=,; for kv in $envs; do export "$kv"; done
      timeout 300 python bench.py --config $cfg --impl fixed --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $spec ->', d['config']['stream_format'], d['config']['col_slices'], 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), d['parity_vs_oracle'][:12])" )
  done
done 2>&1 | tee gpurun_out/r05/t80_formats.txt
