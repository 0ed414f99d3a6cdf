#!/bin/bash
# PMC counters of gemm256 on the encoder's product shapes: bash tools/gemm_pmc.sh [case ...]   (default: all eight)
# Three separate --pmc passes (8 SQ slots each) + one --kernel-trace --stats pass per case, no other trace flags
# (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Output: gpurun_out/gemm_pmc/<case>/<pass>/; summary: tools/gemm_pmc_summary.py
R="$(cd "$(dirname "$0")/.." && pwd)"
CASES=${@:-qkv out fc1 fc2 dqkv dout dfc1 dfc2}
cd /tmp && export TMPDIR=/tmp
for c in $CASES; do
  O=$R/gpurun_out/gemm_pmc/$c
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/a -- python3 $R/tools/gemm_shape_run.py $c > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/b -- python3 $R/tools/gemm_shape_run.py $c > /dev/null 2>&1
  rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_WAVES --output-format csv -d $O/c -- python3 $R/tools/gemm_shape_run.py $c > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/gemm_shape_run.py $c > /dev/null 2>&1
done
python3 $R/tools/gemm_pmc_summary.py $R/gpurun_out/gemm_pmc $CASES
