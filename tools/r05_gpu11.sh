#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu11.txt
{
for v in 0 1 2; do
  echo "== epilogue stamps, S2T_EPX=$v (0 as shipped, 1 no global stores, 2 no LDS turn), NT 24000 x 2048 x 512 with bias"
  S2T_HIP_LIB=$L/libs2t_hip_labepx$v.so python tools/gemm_timeline.py 2048 512 bias | grep "epilogue\|period between"
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-260
