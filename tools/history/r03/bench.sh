#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.log
tail -45 gpurun_out/r03/bench_default.log
tail -c 3000 gpurun_out/r03/bench_default.json
