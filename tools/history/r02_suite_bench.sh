#!/bin/bash
# whole GPU suite, then bench.py (the driver's command line), outputs under gpurun_out/
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r02_gpu_suite.log
cat gpurun_out/r02_gpu_suite.log
timeout 900 python bench.py > gpurun_out/r02_bench_n1.log 2>gpurun_out/r02_bench_n1.err
tail -1 gpurun_out/r02_bench_n1.log > gpurun_out/r02_bench_n1.json
tail -3 gpurun_out/r02_bench_n1.err
cat gpurun_out/r02_bench_n1.json
