#!/bin/bash
# round 4: SWEEP blocks to XCDs by column slice (an XCD's L2 then holds the x its workgroups gather) against the spread assignment
mkdir -p gpurun_out
(for c in "pokec fixed" "pokec float_stall" "pokec float_pob"; do set -- $c
  for a in 1 0; do HISPARSE_XCD_AFFINITY=$a TAG="xcd_affinity=$a" timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep -E "step us"; done
done
for a in 1 0; do HISPARSE_SWEEP=1 HISPARSE_XCD_AFFINITY=$a TAG="xcd_affinity=$a" timeout 300 python tools/probe_cfg.py ogbn_products float_stall 2>&1 | grep -E "step us"; done
) > gpurun_out/r04_sweep_xcd_affinity.txt 2>&1
cat gpurun_out/r04_sweep_xcd_affinity.txt | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_parity.py -k sweep -x -q 2>&1 | tail -3
