#!/bin/bash
# round 5, first GPU call: the slice join (spmv_device.h) -- parity, then A/B against the combine launch on the sliced configurations
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 1500 python -m pytest tests/test_gpu_slice_join.py -x -q 2>&1 | tail -8 > gpurun_out/r05/join_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "column_slices or repeated" 2>&1 | tail -5 >> gpurun_out/r05/join_tests.txt
for cfg in ogbl_ppa gplus pokec ogbn_products ogbl_ppa_rmat hollywood; do
  for j in 0 1 0 1; do
    HISPARSE_SLICE_JOIN=$j timeout 300 python bench.py --config $cfg --steps 500 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg join=$j', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['frac_whole_step'])"
  done
done > gpurun_out/r05/join_ab.txt 2>&1
cat gpurun_out/r05/join_tests.txt gpurun_out/r05/join_ab.txt
