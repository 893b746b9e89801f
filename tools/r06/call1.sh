#!/bin/bash
# round 6, GPU call 1: (a) the ADVICE regression tests + the csim dataset cases; (b) row-block stream policy A/B (nt | sc1 | plain), whole step, alternating
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_carry.py tests/test_spmspv.py tests/test_benchmark_cli.py -m gpu -x -q > gpurun_out/r06/call1_tests.log 2>&1
tail -5 gpurun_out/r06/call1_tests.log
out=gpurun_out/r06/rowblock_stream_policy.txt; : > $out
L=$PWD/hisparse_amd/lib
for spec in "mouse_gene fixed" "gplus fixed" "transformer_80 fixed" "transformer_90 fixed" "transformer_70 fixed" "mouse_gene float_pob" "ogbl_ppa fixed"; do
  set -- $spec
  for round in 1 2; do
    for lib in "" _rb_sc1 _rb_plain; do
      echo -n "$1/$2 ${lib:-nt}: " >> $out
      HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 200 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us" | cut -c42-150 >> $out
    done
  done
done
for spec in "mouse_gene 8" "hollywood 4" "mouse_gene 4"; do
  set -- $spec
  for lib in "" _rb_sc1 _rb_plain; do
    echo "== $1 $2-way slabs ${lib:-nt}" >> $out
    HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 300 python tools/slab_probe.py $1 $2 "default:" 2>&1 | grep "way slab [03]" >> $out
  done
done
cat $out
