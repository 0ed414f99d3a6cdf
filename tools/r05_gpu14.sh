#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu14.txt
{
for i in 1 2 3; do
  echo "-- tree (cluster 1 / memory 0)"; python tools/wgrad_bench.py | cut -c1-110; LAYERS=12 python tools/wgrad_bench.py | cut -c1-110
  for v in wg00 wg01; do echo "-- $v"; S2T_HIP_LIB=$L/libs2t_hip_$v.so python tools/wgrad_bench.py | cut -c1-110; S2T_HIP_LIB=$L/libs2t_hip_$v.so LAYERS=12 python tools/wgrad_bench.py | cut -c1-110; done
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -40
