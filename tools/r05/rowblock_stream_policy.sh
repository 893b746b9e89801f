#!/bin/bash
# NEXT (not run in round 5: no GPU minutes left): the row-block kernels' stream loads without the non-temporal hint on images that fit the 256 MiB
# Infinity Cache -- what gave the SWEEP kernel 2-19 % (profiles/r05_sweep_stream_policy.txt).  Whole step, alternating, one box.
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
make -j16 variant NAME=rb_sc1 DEFS='-DHS_ROWBLOCK_STREAM_POLICY="\"sc1\""' > /dev/null 2>&1 || exit 1
out=gpurun_out/r05/rowblock_stream_policy.txt; : > $out
for cfg in mouse_gene gplus transformer_80 ogbl_ppa hollywood; do
  for lib in "" _rb_sc1 "" _rb_sc1; do
    echo -n "$cfg ${lib:-nt}: " >> $out
    HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib/libhisparse_hip$lib.so timeout 200 python tools/probe_cfg.py $cfg 2>&1 | grep "step us" | cut -c42-130 >> $out
  done
done
for m in mouse_gene hollywood; do
  for lib in "" _rb_sc1; do
    echo "== $m 8-way slabs ${lib:-nt}" >> $out
    HISPARSE_HIP_LIB=$PWD/hisparse_amd/lib/libhisparse_hip$lib.so timeout 300 python tools/slab_probe.py $m 8 "default:" 2>&1 | grep "way slab [03]" >> $out
  done
done
cat $out
