#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu7.txt
{
echo "== route test"; timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -s -k "gemm256_route" 2>&1 | grep -v amdgpu.ids | tail -40
echo "== gemm256 NN: builtin tr reads (g256tr0 twin) vs asm (tree)"
for i in 1 2 3; do
  echo "-- g256tr0"; S2T_HIP_LIB=$L/libs2t_hip_g256tr0.so python tools/gemm_x_time.py 0
  echo "-- tree"; python tools/gemm_x_time.py 0
done
echo "== gemm tests"; timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_big or gemm256 or relu_one_bit or gemm_nn or odd_vocab" 2>&1 | tail -4
echo "== attention tests"; timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "attention or alignment" 2>&1 | tail -4
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-1500 | tail -80
