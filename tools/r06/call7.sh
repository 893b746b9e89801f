#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
bash tools/r06/sweep_depth.sh
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_n1.out 2> gpurun_out/r06/bench_n1.err
cp bench_details.json gpurun_out/r06/bench_details.json 2>/dev/null
tail -c 1500 gpurun_out/r06/bench_n1.out
