#!/bin/bash
# the 8-rank dry run, repeatedly, keeping the full output of a failing attempt
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
for i in 1 2 3 4 5 6; do
  timeout 600 python bench.py --gpus 8 --backend gloo --share-gpu --config mouse_gene --steps 20 --warmup 5 > gpurun_out/r05/dry8_$i.out 2> gpurun_out/r05/dry8_$i.err
  rc=$?
  echo "attempt $i rc=$rc"
  if [ $rc -ne 0 ]; then grep -v "^\[W\|Gloo\|^W0\|^\*\*\*" gpurun_out/r05/dry8_$i.err | grep -v "^$" | head -60; cat gpurun_out/r05/dry8_$i.out | tail -3; break; fi
done
