#!/bin/bash
# round 6, GPU call 3: (a) the new option / policy tests; (b) the plan-time stream policy as shipped, on the matrices call 1 / 2 measured by hand; (c) the
# planner's out-of-sample check (tools/planner_check.py)
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/test_gpu_options.py tests/test_gpu_sweep.py tests/test_gpu_carry.py -m gpu -x -q > gpurun_out/r06/call3_tests.log 2>&1
tail -3 gpurun_out/r06/call3_tests.log
out=gpurun_out/r06/stream_policy_shipped.txt; : > $out
for spec in "ogbl_ppa fixed" "mouse_gene fixed" "gplus fixed" "transformer_80 fixed" "hollywood fixed" "ogbl_ppa_rmat fixed"; do
  set -- $spec
  for res in "" 0 1; do
    echo -n "$1/$2 stream_resident=${res:-plan}: " >> $out
    HISPARSE_STREAM_RESIDENT=$res timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us" | cut -c42-150 >> $out
  done
done
for spec in "mouse_gene 8" "mouse_gene 2" "ogbl_ppa 8"; do
  set -- $spec
  echo "== $1 $2-way slabs" >> $out
  timeout 400 python tools/slab_probe.py $1 $2 "plan:" "nt:HISPARSE_STREAM_RESIDENT=0" "sc1:HISPARSE_STREAM_RESIDENT=1" 2>&1 | grep "way slab [03]" >> $out
done
cat $out
timeout 2400 python tools/planner_check.py --json gpurun_out/r06/planner_check.json > gpurun_out/r06/planner_check.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check.txt | tail -40
