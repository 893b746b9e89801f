#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
ROUNDS=3 timeout 900 python tools/probe_variants.py gplus "default:" "1sl:HISPARSE_COL_SLICES=1" "7sl:HISPARSE_COL_SLICES=7" "5sl:HISPARSE_COL_SLICES=5" 2>&1 | tail -4
python bench.py --config bm --steps 100 --warmup 20 2> gpurun_out/r03/bm.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for e in d['bm_list']: print(e['matrix'], e['stream_format'], e['col_slices'], e['ms_per_step'], e['hbm_roofline_fraction_whole_job'], e['parity_vs_oracle'])
"
} > gpurun_out/r03/unit_cost_plans2.log 2>&1
cat gpurun_out/r03/unit_cost_plans2.log
