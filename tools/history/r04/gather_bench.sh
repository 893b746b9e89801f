#!/bin/bash
# round 4: price the no-LDS-x row-owner block before building it (tools/gather_bench.hip); gather cache policies
mkdir -p gpurun_out
for v in plain sc1 sc0sc1 nt; do echo "== gather policy: $v"; timeout 300 tools/gather_bench_$v.bin 30; done 2>&1 | tee gpurun_out/r04_gather_bench.txt
