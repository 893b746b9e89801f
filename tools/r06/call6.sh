#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_bench_launcher.py -m gpu -q -k "default_series" > gpurun_out/r06/call6_dry.log 2>&1
tail -30 gpurun_out/r06/call6_dry.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "dense_rows_pick_bitmap" 2>&1 | tail -3
