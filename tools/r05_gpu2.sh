#!/bin/bash
# round-5 call 2: deeper lane-turn pipeline -- correctness, same-box A/B, stamped timeline
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu2.txt
{
echo "== tests"; timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_big or gemm256 or relu_one_bit or gemm_nt_epi or gemm_nn or odd_vocab" 2>&1 | tail -5
for i in 1 2 3; do
  for v in td2 td3 td4e0; do echo "== $v"; S2T_HIP_LIB=$L/libs2t_hip_$v.so python tools/gemm_x_time.py 0; done
  echo "== tree"; python tools/gemm_x_time.py 0
done
for sh in "2048 512" "1536 512"; do
  echo "== timeline dbg1 $sh"; S2T_HIP_LIB=$L/libs2t_hip_dbg1.so python tools/gemm_timeline.py $sh
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -60
