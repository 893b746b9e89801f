#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 2400 python tools/planner_check.py --json gpurun_out/r06/planner_check_after4.json > gpurun_out/r06/planner_check_after4.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check_after4.txt | tail -28 | cut -c1-240
timeout 2400 python tools/planner_check.py --second --json gpurun_out/r06/planner_check_second2.json > gpurun_out/r06/planner_check_second2.txt 2>&1
grep -v "^    " gpurun_out/r06/planner_check_second2.txt | tail -15 | cut -c1-240
bash tools/r06/suite_soak.sh
bash tools/r06/final.sh > gpurun_out/r06/final.log 2>&1
tail -5 gpurun_out/r06/final.log | cut -c1-200
