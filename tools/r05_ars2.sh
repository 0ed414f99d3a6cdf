#!/bin/bash
# activation-stationary GEMM probe: the two MFMA shapes side by side (stamped build)
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ars3.txt
{
for a in "24000 2048 5" "24000 1536 5"; do
  for sh in 32 16 16 32; do
    if [ $sh = 16 ]; then ARS_SHAPE16=1 timeout 60 tools/_bin/ars_probe_st $a 30 | grep -v "^  row" | grep "check\|wg    0\|wg  104\|gemm_ars"
    else timeout 60 tools/_bin/ars_probe_st $a 30 | grep -v "^  row" | grep "check\|wg    0\|wg  104\|gemm_ars"; fi
  done
done
} > $O 2>&1
cat $O
