#!/bin/bash
# round 6: the carried combine on mid-size images (48-160 MiB), where round 5 had no measurement between "8.6 vs 10.1 us" and "a wash"
cd "${GRAFT_REPO_ROOT:-.}" && mkdir -p gpurun_out/r06
out=gpurun_out/r06/carry_mid_size.txt; : > $out
for cfg in gplus mouse_gene_slab2 mouse_gene_slab4 transformer_70; do
  for round in 1 2; do
    for c in 0 1; do
      echo -n "$cfg carry_combine=$c: " >> $out
      HISPARSE_CARRY_COMBINE=$c timeout 300 python tools/probe_cfg.py $cfg fixed 2>&1 | grep "step us" | cut -c42-150 >> $out
    done
  done
done
for spec in "ogbl_ppa 8" "ogbl_ppa 4" "ogbl_ppa 2" "hollywood 8"; do
  set -- $spec
  echo "== $1 $2-way slabs" >> $out
  timeout 400 python tools/slab_probe.py $1 $2 "plan:" "carry0:HISPARSE_CARRY_COMBINE=0" "carry1:HISPARSE_CARRY_COMBINE=1" 2>&1 | grep "way slab [03]" >> $out
done
cat $out
