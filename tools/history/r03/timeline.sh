#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 600 python tools/rowblock_timeline.py ogbn_products 2>&1 | tail -40
timeout 600 python tools/rowblock_timeline.py ogbl_ppa 2>&1 | tail -25
timeout 600 python tools/rowblock_timeline.py mouse_gene 2>&1 | tail -25
} > gpurun_out/r03/timeline.log 2>&1
cat gpurun_out/r03/timeline.log
