#!/bin/bash
# round 5: hs_run_batch -- plain C loop against hipGraph replay: tests, the headline, a small matrix, and every slab of the 2/4/8-way split of mouse_gene
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_options.py tests/test_benchmark_cli.py -m gpu -x -q 2>&1 | tail -4 | cut -c1-300
for cfg in ogbl_ppa transformer_95 gplus csim_1k; do
  timeout 300 python bench.py --config $cfg --steps 500 --warmup 50 --no-cpu-baseline --quick 2>&1 | grep "graph replay" | tail -1
done
timeout 900 python bench.py --predict-scaling --config mouse_gene 2>&1 | grep -v "^\[bench\] mouse_gene/" | tail -8 | cut -c1-900
