#!/bin/bash
# kernel time of every BASELINE config (best of 3 x 50): bash tools/probe_all.sh [configs...]
for c in ${@:-transformer_50 mouse_gene ogbl_ppa ogbn_products}; do python tools/probe_cfg.py $c; done
