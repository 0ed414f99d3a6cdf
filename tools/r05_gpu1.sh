#!/bin/bash
# round-5 call 1: gemm256 early restage -- correctness, same-box A/B, stamped timelines
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu1.txt
{
echo "== tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_big or gemm256_epilogues or relu_one_bit or deterministic_under_load or gemm_nt_epi or gemm_nn" 2>&1 | tail -5
for i in 1 2 3; do
  echo "== e0";   S2T_HIP_LIB=$L/libs2t_hip_e0.so python tools/gemm_x_time.py 0
  echo "== tree"; python tools/gemm_x_time.py 0
done
for v in dbg0 dbg1; do
  for sh in "2048 512" "512 512" "1536 512"; do
    echo "== timeline $v $sh"; S2T_HIP_LIB=$L/libs2t_hip_$v.so python tools/gemm_timeline.py $sh
  done
done
} > $O 2>&1
tail -40 $O
