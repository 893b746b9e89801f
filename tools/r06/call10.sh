#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
out=gpurun_out/r06/sweep_depth_streamed.txt; : > $out
L=$PWD/hisparse_amd/lib
for lib in "" _sw6 _sw8 _sw12; do
  HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 900 python tools/r06/sweep_depth_streamed.py >> $out 2>&1
done
cat $out | cut -c1-200
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_n1.out 2> gpurun_out/r06/bench_n1.err
cp bench_details.json gpurun_out/r06/bench_details.json 2>/dev/null
grep -E "^\[bench\] (mouse_gene/fixed|ogbl_ppa/fixed)  " gpurun_out/r06/bench_n1.err | cut -c1-150
tail -c 600 gpurun_out/r06/bench_n1.out
