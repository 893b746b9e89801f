#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
timeout 1200 python -m pytest tests/test_gpu_retile.py tests/test_gpu_load_csr.py tests/test_spmm.py tests/test_gpu_soak.py -x -q -m gpu 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "bitmap or transformer or dense" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_perf_model.py -x -q -m gpu -k "transformer" -s 2>&1 | grep -i "transformer\|passed\|failed" | tail -4
FUZZ_PROFILE=dense timeout 1500 python tests/gpu_fuzz_soak.py 300 403 2>&1 | tail -2
python bench.py --config transformer_50 --steps 200 --warmup 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('transformer_50 float_pob', d['ms_per_step'], d['hbm_roofline_fraction_whole_job'], d['roofline']['frac'], d['roofline'].get('frac_mall_cold'))"
} > gpurun_out/r03/bitmap_final2.log 2>&1
cat gpurun_out/r03/bitmap_final2.log
