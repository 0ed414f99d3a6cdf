#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu18.txt
{
echo "== bench A/B, alternating order (adam0 = cached one-group Adam loop; tree = two groups per step, non-temporal)"
for v in tree adam0 adam0 tree tree adam0 adam0 tree; do
  lib=$L/libs2t_hip.so; [ $v = adam0 ] && lib=$L/libs2t_hip_adam0.so
  echo "-- $v $(S2T_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -30
