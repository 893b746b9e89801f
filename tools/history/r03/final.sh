#!/bin/bash
# round 3 closing run: the whole GPU suite, the default bench line (all configurations, bm list, MALL-cold legs, scaling prediction)
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r03/gpu_suite.log 2>&1
tail -5 gpurun_out/r03/gpu_suite.log
( time python bench.py --steps 200 --warmup 20 ) > gpurun_out/r03/bench_n1.json 2> gpurun_out/r03/bench_n1.log
grep "\[bench\]" gpurun_out/r03/bench_n1.log | tail -60
python bench.py --force-dist --steps 100 --warmup 10 > gpurun_out/r03/bench_force_dist_n1.json 2> gpurun_out/r03/bench_force_dist_n1.log
tail -c 600 gpurun_out/r03/bench_force_dist_n1.json
