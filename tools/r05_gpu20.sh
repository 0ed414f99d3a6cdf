#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu20.txt
{
echo "== bench, non-temporal stores by site mask (S2T_NT: 1 bn_apply, 2 conv1 fwd, 4 conv2 dgrad, 8 conv2 fwd, 16 bn_bwd_apply, 32 attention fwd, 64 attention bwd), alternating"
for v in tree nt31 nt96 nt127 nt127 nt96 nt31 tree tree nt31 nt96 nt127; do
  lib=$L/libs2t_hip.so; [ $v != tree ] && lib=$L/libs2t_hip_$v.so
  echo "-- $v $(S2T_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300
