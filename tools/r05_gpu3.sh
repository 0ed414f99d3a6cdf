#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu3.txt
{
echo "== oob probe, td2 twin (round-4 addressing)"; S2T_HIP_LIB=$L/libs2t_hip_td2.so python tools/oob_probe.py
echo "== oob probe, tree"; python tools/oob_probe.py
echo "== tests (tree)"; timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_big or gemm256 or relu_one_bit or gemm_nt_epi or gemm_nn or odd_vocab" 2>&1 | tail -5
for sh in "2048 512" "2048 512 bias" "512 512 bias"; do
  echo "== timeline dbg1 $sh"; S2T_HIP_LIB=$L/libs2t_hip_dbg1.so python tools/gemm_timeline.py $sh | grep -v "reads:\|dma:\|mem start\|MEM segment"
done
for i in 1 2; do
  for v in td2; do echo "== $v"; S2T_HIP_LIB=$L/libs2t_hip_$v.so python tools/gemm_x_time.py 0; done
  echo "== tree"; python tools/gemm_x_time.py 0
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -70
