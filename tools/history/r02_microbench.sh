#!/bin/bash
# Round-2 measurement pass: the micro-benchmarks the roofline argument cites + per-config kernel times.
# usage (from the repo root, on the GPU box): bash tools/r02_microbench.sh [outdir]
out=${1:-gpurun_out/r02}
mkdir -p $out
for b in lds_accum_bench lds_atomic_bench hbm_read_bench record_stream_bench; do
  [ -x tools/$b.bin ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result -o tools/$b.bin tools/$b.hip
  timeout 120 tools/$b.bin > $out/$b.txt 2>&1
done
