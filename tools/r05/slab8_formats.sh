#!/bin/bash
# round 5, after the lane-major dealing of DELTA runs: one rank's slab of mouse_gene split 8 ways under DELTA plans against the planner's PAIRS plan
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
for cfg in mouse_gene_slab8; do
  for spec in "auto:" "delta:HISPARSE_COL_SLICES=1" "delta:HISPARSE_COL_SLICES=3" "delta:HISPARSE_COL_SLICES=6" "pairs:HISPARSE_COL_SLICES=6" "delta:HISPARSE_COL_SLICES=6,HISPARSE_ROW_RUNS=1" "delta:HISPARSE_COL_SLICES=6,HISPARSE_ROW_RUNS=0"; do
    fmt=${spec%%:*}; envs=${spec#*:}
    ( [ $fmt != auto ] && export HISPARSE_STREAM_FORMAT=$fmt; IFS=,; for kv in $envs; do export "$kv"; done
      timeout 300 python bench.py --config $cfg --impl fixed --steps 500 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg $spec ->', d['config']['stream_format'], d['config']['col_slices'], 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), d['parity_vs_oracle'][:12])" )
  done
done 2>&1 | tee gpurun_out/r05/slab8_formats.txt
