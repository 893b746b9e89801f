#!/bin/bash
# Profiles of the bench command on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats           -> per-kernel durations
#   2. rocprofv3 --pmc FETCH_SIZE   (own pass)    -> HBM read bytes per launch
#   3. rocprofv3 --pmc WRITE_SIZE   (own pass)    -> HBM write bytes per launch
# Counter passes use --kernel-trace only (never sys/hip/hsa traces together with --pmc).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python bench.py --steps 50 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
find "$OUT" -name "*.csv" | head -50
python tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
