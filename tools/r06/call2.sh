#!/bin/bash
# round 6, GPU call 2: row-block stream policy on the LARGE images (above the Infinity Cache) and the remaining formats: does `nt` still win anywhere?
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
out=gpurun_out/r06/rowblock_stream_policy_large.txt; : > $out
L=$PWD/hisparse_amd/lib
for spec in "hollywood fixed" "ogbn_products float_stall" "ogbn_products fixed" "ogbl_ppa_rmat fixed" "mouse_gene float_stall" "csim_1k fixed"; do
  set -- $spec
  for round in 1 2; do
    for lib in "" _rb_sc1 _rb_plain; do
      echo -n "$1/$2 ${lib:-nt}: " >> $out
      HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us" | cut -c42-150 >> $out
    done
  done
done
for spec in "mouse_gene 2" "hollywood 8" "ogbl_ppa 8" "ogbl_ppa 2"; do
  set -- $spec
  for lib in "" _rb_sc1; do
    echo "== $1 $2-way slabs ${lib:-nt}" >> $out
    HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 400 python tools/slab_probe.py $1 $2 "default:" 2>&1 | grep "way slab [03]" >> $out
  done
done
cat $out
