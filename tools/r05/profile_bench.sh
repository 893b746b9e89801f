#!/bin/bash
# round 5: rocprofv3 --kernel-trace --stats of the bench command itself (the headline configuration only: --quick, so that the run does not start
# its own counter passes under the profiler), summarised like the per-configuration profiles
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out/prof_bench
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python bench.py --quick --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline"
$CMD > "$OUT/plain.out" 2> "$OUT/plain.err"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
python tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
echo "== the same command without the profiler, same box ==" >> "$OUT/summary.txt"
tail -1 "$OUT/plain.out" | cut -c1-1800 >> "$OUT/summary.txt"
echo "== under the profiler ==" >> "$OUT/summary.txt"
grep -E '^\{"metric"' "$OUT/stats.log" | tail -1 | cut -c1-1800 >> "$OUT/summary.txt"
rm -rf "$OUT/stats"
cat "$OUT/summary.txt" | cut -c1-260
