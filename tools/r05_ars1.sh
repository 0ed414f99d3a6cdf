#!/bin/bash
# activation-stationary GEMM probe (tools/lab/ars_probe.hip): correctness + time on the K = 512 shapes
cd "$(dirname "$0")/.."
O=gpurun_out/r05_ars1.txt
{
for a in "24000 1536 5" "24000 1536 3" "24000 1536 8" "24000 2048 5" "24000 2048 8" "24000 6144 5" "24000 1024 5" "2560 1536 5" "23872 1536 5"; do
  timeout 60 tools/_bin/ars_probe $a 30
done
timeout 60 tools/_bin/ars_probe_st 24000 1536 5 30
timeout 60 tools/_bin/ars_probe_st 24000 2048 5 30
} > $O 2>&1
cat $O
