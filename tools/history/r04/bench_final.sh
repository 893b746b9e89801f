#!/bin/bash
mkdir -p gpurun_out
SECONDS=0; python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.log; echo "bench.py default run (--steps 20 --warmup 5): $SECONDS s, rc $?"
grep "^\[bench\]" gpurun_out/r04_bench_n1.log | grep -v "oracle\|generate" | tail -50
SECONDS=0; python bench.py > gpurun_out/r04_bench_n1_default.json 2> gpurun_out/r04_bench_n1_default.log; echo "bench.py default run (no flags): $SECONDS s"
