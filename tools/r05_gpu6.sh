#!/bin/bash
cd "$(dirname "$0")/.."
L=$PWD/fbk_fairseq_st_amd
O=gpurun_out/r05_gpu6.txt
{
echo "== wgrad A/B (builtin tr reads = wgtr0 twin, asm = tree)"
for i in 1 2 3; do
  echo "-- wgtr0"; S2T_HIP_LIB=$L/libs2t_hip_wgtr0.so python tools/wgrad_bench.py; S2T_HIP_LIB=$L/libs2t_hip_wgtr0.so LAYERS=12 python tools/wgrad_bench.py
  echo "-- tree";  python tools/wgrad_bench.py; LAYERS=12 python tools/wgrad_bench.py
done
echo "== wgrad tests"; timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "wgrad" 2>&1 | tail -4
echo "== new tests"; timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py tests/test_configs_gpu.py -x -q -s -k "attention or alignment or normalized_probs or softmax_rows or gemm256_route or soak" 2>&1 | grep -v amdgpu.ids | tail -25
echo "== full gpu suite"; ( time timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 ) 2>&1
echo "== bench"; python bench.py --no-cpu-baseline --no-extra 2>&1 | tail -1
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-2500 | tail -70
