#!/bin/bash
# round 6: the SWEEP ring depth measured again now that resident images are streamed with `sc1` (round 5's sweep -- profiles/r05_sweep_ring_depth.txt -- was taken
# with `nt` streams: deeper rings lost, presumably because streamed lines pushed the gathered lines of x out of the L1)
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
out=gpurun_out/r06/sweep_depth_sc1.txt; : > $out
L=$PWD/hisparse_amd/lib
for spec in "pokec fixed" "pokec float_pob"; do
  set -- $spec
  for lib in _sw3 "" _sw6 _sw8 _sw16 _sw16d6; do
    echo -n "$1/$2 ${lib:-d4}: " >> $out
    HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 300 python tools/probe_cfg.py $1 $2 2>&1 | grep "step us" | cut -c42-150 >> $out
  done
done
for lib in _sw3 "" _sw6 _sw8 _sw16 _sw16d6; do
  echo "== ogbn_products 8-way slabs / hollywood 8-way slabs ${lib:-d4}" >> $out
  HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 400 python tools/slab_probe.py ogbn_products 8 "default:" 2>&1 | grep "way slab [03]" >> $out
  HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 400 python tools/slab_probe.py hollywood 8 "default:" 2>&1 | grep "way slab [0]" >> $out
  HISPARSE_HIP_LIB=$L/libhisparse_hip$lib.so timeout 300 python tools/planner_check.py --only rmat20_sym_50_20_20,er_1500k_8,bipartite_100k_x_4m_60 2>&1 | grep -E "^(rmat|er_|bip)" | cut -c1-110 >> $out
done
cat $out
