#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu5.txt
{
echo "== tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "softmax_rows or normalized_probs" 2>&1 | tail -8
for i in 1 2 3; do
  for v in base sc1 nt sc01; do echo "== $v"; S2T_HIP_LIB=$L/libs2t_hip_$v.so python tools/gemm_x_time.py 0; done
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -70
