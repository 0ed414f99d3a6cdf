#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu16.txt
{
echo "== bench A/B, alternating order (prev = priorities 1/0; tree = gemm256 0/1, wgrad 0/0)"
for v in tree prev prev tree tree prev prev tree; do
  lib=$L/libs2t_hip.so; [ $v = prev ] && lib=$L/libs2t_hip_prev.so
  echo "-- $v $(S2T_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -30
