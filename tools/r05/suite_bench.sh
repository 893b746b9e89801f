#!/bin/bash
# round 5: the whole GPU suite, then the driver's bench command line; outputs under gpurun_out/r05/
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05/gpu_suite.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05/bench_n1.out 2> gpurun_out/r05/bench_n1.err
cp bench_details.json gpurun_out/r05/bench_details.json 2>/dev/null
tail -3 gpurun_out/r05/gpu_suite.txt; tail -c 1500 gpurun_out/r05/bench_n1.out; tail -45 gpurun_out/r05/bench_n1.err
