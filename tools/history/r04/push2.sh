#!/bin/bash
# the N-rank path of bench.py on one GPU (--force-dist: world size 1 through the distributed code) after the default exchange became the final gather
mkdir -p gpurun_out
python bench.py --force-dist --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r04_bench_force_dist_n1.json 2> gpurun_out/r04_bench_force_dist_n1.log; tail -3 gpurun_out/r04_bench_force_dist_n1.log | cut -c1-300
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_force_dist_n1.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ("n_gpus","value","ms_per_step","compute_only","exchange","exchange_every_step","exchange_push","same_workload_on_one_gpu")})
PY
python bench.py --gpus 2 2>&1 | tail -2 | cut -c1-300; echo "rc of --gpus 2 on a 1-GPU box: $?"
