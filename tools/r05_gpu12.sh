#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu12.txt
{
echo "== gemm tests (tree: 8 MFMAs behind the barrier)"; timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_big or gemm256 or relu_one_bit or gemm_nn or odd_vocab or gemm_nt_epi" 2>&1 | tail -4
for i in 1 2 3; do
  for v in tail0 tail2; do echo "-- $v"; S2T_HIP_LIB=$L/libs2t_hip_$v.so python tools/gemm_x_time.py 0; done
  echo "-- tree (tail1)"; python tools/gemm_x_time.py 0
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -40
