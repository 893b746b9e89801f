#!/bin/bash
# round 4: float sweep parity at full size, then the default bench line (timed)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "light" 2>&1 | tail -3
SECONDS=0; python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_n1.json 2> gpurun_out/r04_bench_n1.log; echo "bench.py default run: $SECONDS s"

grep "^\[bench\]" gpurun_out/r04_bench_n1.log | tail -60
