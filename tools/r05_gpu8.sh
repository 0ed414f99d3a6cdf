#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu8.txt
{
echo "== ln_bwd: one row ahead vs queue (in-process A/B)"; python tools/ln_time.py
echo "== attention, 12-slot hash twin vs tree (10 slots)"
for i in 1 2; do
  echo "-- hash12"; S2T_HIP_LIB=$L/libs2t_hip_hash12.so python tools/attn_time.py
  echo "-- tree";   S2T_HIP_LIB=$L/libs2t_hip.so python tools/attn_time.py
done
echo "== tests"; timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -k "layernorm or dropout or attention or ln_ or engine" 2>&1 | tail -5
echo "== bench A/B"
for i in 1 2; do
  echo "-- hash12 + ln one-row"; S2T_HIP_LIB=$L/libs2t_hip_hash12.so python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | cut -c1-200
  echo "-- tree"; S2T_HIP_LIB=$L/libs2t_hip.so python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | cut -c1-200
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-400 | tail -90
