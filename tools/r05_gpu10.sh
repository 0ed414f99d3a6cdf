#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_gpu10.txt
{
echo "== full gpu suite"; ( time timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 2>&1 | tail -45 ) 2>&1
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300 | tail -60
