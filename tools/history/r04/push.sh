#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_peer_gather.py -x -q 2>&1 | tail -8
python bench.py --force-dist --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_bench_force_dist_n1.json 2> gpurun_out/r04_bench_force_dist_n1.log; tail -3 gpurun_out/r04_bench_force_dist_n1.log | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_force_dist_n1.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("n_gpus","ms_per_step","compute_only","exchange","exchange_push","same_workload_on_one_gpu")})
PY
for c in ogbl_ppa pokec; do FRACS=0.001,0.005,0.01,0.02 python tools/spmspv_probe.py $c; done 2>&1 | tee gpurun_out/r04_spmspv_auto.txt
