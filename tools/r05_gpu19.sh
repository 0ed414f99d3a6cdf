#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu19c.txt
{
echo "== bench, LayerNorm non-temporal stores (tree = backward stores; lnnt0 = none; lnnt10 = backward + forward stores), alternating"
for v in tree lnnt0 lnnt10 lnnt10 lnnt0 tree tree lnnt10 lnnt0; do
  lib=$L/libs2t_hip.so; [ $v != tree ] && lib=$L/libs2t_hip_$v.so
  echo "-- $v $(S2T_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300
