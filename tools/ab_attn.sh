#!/bin/bash
# Same-box A/B of the attention kernels: the library in the tree against a twin built from another attention.hip
# (fbk_fairseq_st_amd/libs2t_hip_base.so).  Usage: bash tools/ab_attn.sh [rounds]
cd "$(dirname "$0")/.."
for i in $(seq 1 ${1:-3}); do
  echo "== base"; S2T_HIP_LIB=$PWD/fbk_fairseq_st_amd/libs2t_hip_base.so python tools/attn_time.py
  echo "== new";  python tools/attn_time.py
done
