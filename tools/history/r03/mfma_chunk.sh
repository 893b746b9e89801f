#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r03
{
for c in 6 9 11 17 6 9; do
  if [ $c = 0 ]; then unset HISPARSE_MFMA_CHUNK; else export HISPARSE_MFMA_CHUNK=$c; fi
  echo "== chunk ${HISPARSE_MFMA_CHUNK:-default}"
  timeout 600 python tools/spmm_probe.py transformer_50 16 2>&1 | tail -1
done
} > gpurun_out/r03/mfma_chunk.log 2>&1
cat gpurun_out/r03/mfma_chunk.log
