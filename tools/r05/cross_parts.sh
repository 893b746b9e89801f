#!/bin/bash
# round 5: row blocks that cross the reference's row-partition borders (tiles_common.h: Layout::cross_parts) -- the GPU tests that run
# partitions one at a time, then A/B on the float_pob configurations (13 / 19 partitions) and the 2- / 3-partition ones
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_light.py tests/test_gpu_bitmap.py tests/test_gpu_retile.py tests/test_benchmark_cli.py -m gpu -x -q -k "partition or multi or retile or reference_literal or from_csr" 2>&1 | tail -6 | cut -c1-300 > gpurun_out/r05/cross_tests.txt
for spec in "ogbn_products float_pob" "pokec float_pob" "ogbn_products float_stall" "pokec fixed" "hollywood fixed" "ogbn_products fixed" "mouse_gene float_pob"; do
  set -- $spec
  for j in 0 1 0 1; do
    HISPARSE_CROSS_PARTITIONS=$j timeout 300 python bench.py --config $1 --impl $2 --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1/$2 cross=$j step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'frac_whole_step', d['roofline']['frac_whole_step'], d['config']['partitions'], d['config']['stream_format'], d['config']['col_slices'], d['parity_vs_oracle'][:40])"
  done
done > gpurun_out/r05/cross_ab.txt 2>&1
cat gpurun_out/r05/cross_tests.txt gpurun_out/r05/cross_ab.txt
