#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 600 python tools/bitmap_timeline.py transformer_50 2>&1 | tail -12
timeout 600 python tools/bitmap_timeline.py transformer_80 2>&1 | tail -12
} > gpurun_out/r03/bitmap_tl.log 2>&1
cat gpurun_out/r03/bitmap_tl.log
