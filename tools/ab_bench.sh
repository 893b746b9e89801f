#!/bin/bash
# whole job per step (bench.py --quick) of two builds of libhisparse_hip.so alternating on one box: bash tools/ab_bench.sh <other.so> <config> ...
other=$1; shift
for cfg in "$@"; do
  for round in 1 2 3; do
    for lib in "" "$other"; do
      HISPARSE_HIP_LIB=$lib timeout 300 python bench.py --config $cfg --steps 500 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
        python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%-14s %-8s ms_per_step %.5f  whole-job frac %.4f  kernel_ms %.5f  %s' % ('$cfg', 'other' if '$lib' else 'current', r['ms_per_step'], r['hbm_roofline_fraction_whole_job'], r['roofline']['kernel_ms'], r['parity_vs_oracle'][:12]))"
    done
  done
done
