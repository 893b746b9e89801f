#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_spmspv -o s -- env FRACS=${FRACS:-0.01} python $GRAFT_REPO_ROOT/tools/spmspv_probe.py ${CFG:-ogbl_ppa} > /tmp/prof_spmspv.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r04_spmspv_rocprof.txt
import glob, sqlite3
db = sqlite3.connect(glob.glob('/tmp/prof_spmspv/**/*.db', recursive=True)[0])
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
    print(f"{name[:90]:90s} calls {calls:6d} avg_us {avg:10.2f} pct {pct:6.2f}")
PY
tail -3 /tmp/prof_spmspv.log
