#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ars2.txt
{
for a in "24000 2048 5" "24000 1536 5"; do
  timeout 60 tools/_bin/ars_probe_st $a 30 | grep -v "^  row"
done
} > $O 2>&1
cat $O
