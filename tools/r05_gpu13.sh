#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu13.txt
{
for i in 1 2 3; do
  for v in prio10 prio00 prio01 prio12; do echo "-- $v (cluster prio / memory-segment prio)"; S2T_HIP_LIB=$L/libs2t_hip_$v.so python tools/gemm_x_time.py 0; done
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -40
