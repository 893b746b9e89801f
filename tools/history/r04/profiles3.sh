#!/bin/bash
# the three single-kernel configurations whose profiled kernel average exceeds the plain step: what the trace says about overlap
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 900 bash tools/profile_cfg.sh mouse_gene 200 > gpurun_out/prof_mouse_gene.log 2>&1; grep -E "^trace|consistency" gpurun_out/prof_mouse_gene/summary.txt
for cfg in transformer_80 transformer_95; do PROFILE_IMPL=fixed timeout 900 bash tools/profile_cfg.sh $cfg 200 > gpurun_out/prof_$cfg.log 2>&1; grep -E "^trace|consistency" gpurun_out/prof_$cfg/summary.txt; done
timeout 900 bash tools/profile_cfg.sh pokec 200 > gpurun_out/prof_pokec.log 2>&1; grep -E "^trace|consistency" gpurun_out/prof_pokec/summary.txt
