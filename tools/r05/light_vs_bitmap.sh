#!/bin/bash
# round 5: the LIGHT plan (PAIRS image, 8 B per non-zero, one launch) forced onto the pruned-NN layers the planner gives to BITMAP
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
for cfg in transformer_60 transformer_70 transformer_80; do
  for impl in fixed float_pob; do
    for light in "" 1; do
      HISPARSE_LIGHT=$light HISPARSE_STREAM_FORMAT=${light:+pairs} timeout 300 python bench.py --config $cfg --impl $impl --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg/$impl light=${light:-auto}', d['config']['stream_format'], 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'frac_whole_step', d['roofline']['frac_whole_step'], d['parity_vs_oracle'][:20])"
    done
  done
done 2>&1 | tee gpurun_out/r05/light_vs_bitmap.txt
