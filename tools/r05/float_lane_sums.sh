#!/bin/bash
# round 5: fp32 lane sums in the float DELTA dense-row path (new) against double lane sums (hisparse_amd/lib_dold/libhisparse_hip.so), same box
cd "$(dirname "$0")/../.." && mkdir -p gpurun_out/r05
OLD=$PWD/hisparse_amd/lib_dold/libhisparse_hip.so
for spec in "mouse_gene float_pob" "mouse_gene float_stall" "transformer_90 float_pob:HISPARSE_STREAM_FORMAT=delta,HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5" "transformer_80 float_pob:HISPARSE_STREAM_FORMAT=delta,HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5" "transformer_70 float_pob:HISPARSE_STREAM_FORMAT=delta,HISPARSE_LIGHT=0,HISPARSE_COL_SLICES=5" "ogbl_ppa float_pob" "hollywood float_pob"; do
  main=${spec%%:*}; envs=${spec#*:}; [ "$envs" = "$spec" ] && envs=""
  set -- $main
  for lib in old new old new; do
    ( [ $lib = old ] && export HISPARSE_HIP_LIB=$OLD; IFS=,; for kv in $envs; do export "$kv"; done
      timeout 300 python bench.py --config $1 --impl $2 --steps 300 --warmup 50 --no-cpu-baseline --quick 2>/dev/null | tail -1 | \
        python -c "import sys,json; d=json.loads(sys.stdin.read()); fe=d.get('float_error',{}); print('$1/$2 $lib ->', d['config']['stream_format'], d['config']['col_slices'], 'step_us', round(d['ms_per_step']*1e3,2), 'kernel_us', round(d['roofline']['kernel_ms']*1e3,2), 'frac_step', d['roofline']['frac_whole_step'], 'err_vs_csim', fe.get('max_abs_err_vs_csim'), 'rows_over', fe.get('rows_over_csim_absolute_1e-4'))" )
  done
done 2>&1 | tee gpurun_out/r05/float_lane_sums.txt
