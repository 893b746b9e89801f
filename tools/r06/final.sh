#!/bin/bash
# round 6, on the FINAL sources: the driver's bench command line, then rocprofv3 --kernel-trace --stats + the two counter passes for every measured
# configuration (tools/profile_cfg.sh), then the bench command itself under rocprofv3 --stats
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_n1.out 2> gpurun_out/r06/bench_n1.err
cp bench_details.json gpurun_out/r06/bench_details.json 2>/dev/null
tail -c 2600 gpurun_out/r06/bench_n1.out; echo; tail -42 gpurun_out/r06/bench_n1.err | cut -c1-200
for cfg in ogbl_ppa transformer_50 ogbn_products mouse_gene ogbl_ppa_rmat pokec hollywood gplus; do
  timeout 900 bash tools/profile_cfg.sh $cfg 200 > gpurun_out/prof_$cfg.log 2>&1
  grep -E "consistency|roofline_frac_rocprof|kernel_avg_us" gpurun_out/prof_$cfg/summary.txt | head -4 | cut -c1-260
done
for cfg in transformer_80 transformer_95; do
  PROFILE_IMPL=fixed timeout 900 bash tools/profile_cfg.sh $cfg 200 > gpurun_out/prof_$cfg.log 2>&1
  grep -E "consistency|roofline_frac_rocprof|kernel_avg_us" gpurun_out/prof_$cfg/summary.txt | head -4 | cut -c1-260
done
timeout 900 bash tools/r05/profile_bench.sh > gpurun_out/r06/profile_bench.log 2>&1
tail -8 gpurun_out/r06/profile_bench.log | cut -c1-300
