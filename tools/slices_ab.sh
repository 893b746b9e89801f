#!/bin/bash
# whole job per step (bench.py: back-to-back steps, one sync) with forced column slices, alternating on one box:
# bash tools/slices_ab.sh <config> <slices A> <slices B>
cfg=$1; a=$2; b=$3
for round in 1 2 3; do
  for cs in $a $b; do
    HISPARSE_COL_SLICES=$cs timeout 300 python bench.py --config $cfg --steps 500 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
      python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%-14s slices=$cs  ms_per_step %.5f  whole-job frac %.4f  kernel_ms %.5f' % ('$cfg', r['ms_per_step'], r['hbm_roofline_fraction_whole_job'], r['roofline']['kernel_ms']))"
  done
done
