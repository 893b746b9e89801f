#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 900 python -m pytest tests/test_spmm.py -x -q 2>&1 | tail -6
for k in 16 32; do
  timeout 300 python tools/spmm_probe.py transformer_50 $k 2>&1 | tail -1
  HISPARSE_SPMM_MFMA=0 timeout 300 python tools/spmm_probe.py transformer_50 $k 2>&1 | tail -1
done
HISPARSE_SPMM_FUSED=0 timeout 300 python tools/spmm_probe.py transformer_50 16 2>&1 | tail -1
timeout 300 python tools/spmm_probe.py transformer_80 16 2>&1 | tail -1
timeout 300 python tools/spmm_probe.py transformer_50 1 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/mfma_prof -o mf -- python $GRAFT_REPO_ROOT/tools/spmm_probe.py transformer_50 16 > /tmp/mf.log 2>&1; cd $GRAFT_REPO_ROOT
python - <<'PY'
import glob, sqlite3
hits = glob.glob('/tmp/mfma_prof/**/*.db', recursive=True)
if hits:
    d = sqlite3.connect(hits[0])
    for name, calls, total, avg, pct in d.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 8"):
        print(f"{name[:90]:90s} calls {calls:5d} avg_us {avg:9.2f} pct {pct:5.1f}")
PY
} > gpurun_out/r03/mfma.log 2>&1
cat gpurun_out/r03/mfma.log
