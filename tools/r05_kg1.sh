#!/bin/bash
# in-workgroup k-groups of the small-product kernels: GPU-side durations by option setting
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for o in "gemm_kgroups=-1" "gemm_kgroups=2" "gemm_kgroups=4" "gemm_kgroups=2,gemm_small_nn=200" "gemm_kgroups=4,gemm_small_nn=200" "gemm_kgroups=0"; do
  echo "== $o"
  rm -rf $R/gpurun_out/sgt
  S2T_OPTS=$o rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sgt -- python3 $R/tools/small_gemm_trace.py run 2560 320 2>&1 | grep "worst"
  python3 $R/tools/small_gemm_trace.py parse $R/gpurun_out/sgt 2560 320 | cut -c1-95
done > $R/gpurun_out/r05_kg1.txt 2>&1
rm -rf $R/gpurun_out/sgt
cat $R/gpurun_out/r05_kg1.txt
