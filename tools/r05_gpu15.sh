#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu15.txt
{
echo "== kernel tests"; timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q 2>&1 | tail -4
echo "== bench A/B (prev = priorities 1/0 in gemm256 and wgrad_group; tree = 0/1 and 0/0)"
for i in 1 2 3; do
  echo "-- prev"; S2T_HIP_LIB=$L/libs2t_hip_prev.so python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | cut -c1-230
  echo "-- tree"; S2T_HIP_LIB=$L/libs2t_hip.so python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | cut -c1-230
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -30
