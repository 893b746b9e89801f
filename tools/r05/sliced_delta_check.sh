#!/bin/bash
# round 5: the planner's own choice for the pruned-NN layers after the sliced-DELTA rule (fixed point), all numeric modes, + the tests that look at plans
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_light.py tests/test_gpu_bitmap.py tests/test_perf_model.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-300
for cfg in transformer_95 transformer_90 transformer_80 transformer_70 transformer_60 transformer_50 mouse_gene_slab8 ppa_small; do
  for impl in fixed float_pob; do
    timeout 300 python bench.py --config $cfg --impl $impl --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg/$impl ->', d['config']['stream_format'], d['config']['col_slices'], 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'frac_step', d['roofline']['frac_whole_step'], 'sync_us', round(d['ms_per_step_synchronous']*1e3,2), d['parity_vs_oracle'][:12])"
  done
done 2>&1 | tee gpurun_out/r05/sliced_delta_check.txt
