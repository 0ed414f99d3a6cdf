#!/bin/bash
# Same-box A/B of the library in the tree against a twin (fbk_fairseq_st_amd/libs2t_hip_base.so, linked from the tree's objects with one
# object built from another revision's source): bash tools/ab_lib.sh <rounds> <python tool and its arguments>
cd "$(dirname "$0")/.."
n=$1; shift
for i in $(seq 1 $n); do
  echo "== base"; S2T_HIP_LIB=$PWD/fbk_fairseq_st_amd/libs2t_hip_base.so python "$@"
  echo "== new";  python "$@"
done
