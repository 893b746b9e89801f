#!/bin/bash
# rocprofv3 passes for ONE named config (tools/probe_cfg.py <config>), run through gpurun from the repo root:
#   bash tools/profile_cfg.sh <config> [runs]
#   0. plain run (step time)   1. --kernel-trace --stats   2. --pmc FETCH_SIZE   3. --pmc WRITE_SIZE   (counter passes use --kernel-trace only)
# Output: gpurun_out/prof_<config>/summary.txt (+ hbm_traffic.json)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
CFG=${1:?config name}
export RUNS=${2:-20}
OUT=gpurun_out/prof_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/probe_cfg.py $CFG ${PROFILE_IMPL:-}"      # PROFILE_IMPL=fixed|float_pob|float_stall: another numeric mode than the config's own
# 0. the same command WITHOUT the profiler, on this box: the step time the kernel averages below must fit inside
#    (under rocprofv3 the wall-clock step carries the tracing, and the counter passes serialise the launches)
PROBE_JSON="$OUT/probe.json" $CMD > "$OUT/plain.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- $CMD > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
python tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
head -1 "$OUT/plain.log" >> "$OUT/summary.txt"
tail -3 "$OUT/stats.log" >> "$OUT/summary.txt"
# the rocpd databases are tens of MB each and gpurun merges at most 64 MiB back: keep the summaries only
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
cat "$OUT/summary.txt"
