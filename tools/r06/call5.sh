#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 1500 python tools/r06/fmt_ab.py > gpurun_out/r06/fmt_ab.txt 2>&1
cat gpurun_out/r06/fmt_ab.txt
