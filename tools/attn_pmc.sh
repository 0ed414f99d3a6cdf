#!/bin/bash
# PMC counters of the encoder-shaped attention kernels (tree library and, if present, the base twin): bash tools/attn_pmc.sh [p_drop]
# Separate --pmc passes, no trace flags (MI355X_MICROARCH.md, profiling section).  Output: gpurun_out/attn_pmc/<lib>/<pass>/
R="$(cd "$(dirname "$0")/.." && pwd)"; P=${1:-0.1}
cd /tmp && export TMPDIR=/tmp
for lib in new base; do
  if [ $lib = base ]; then export S2T_HIP_LIB=$R/fbk_fairseq_st_amd/libs2t_hip_base.so; [ -f $S2T_HIP_LIB ] || continue; else unset S2T_HIP_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/attn_pmc/$lib/a -- python3 $R/tools/attn_only.py $P > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $R/gpurun_out/attn_pmc/$lib/b -- python3 $R/tools/attn_only.py $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/attn_pmc/$lib/t -- python3 $R/tools/attn_only.py $P > /dev/null 2>&1
done
find $R/gpurun_out/attn_pmc -name "*.csv" | head -30
