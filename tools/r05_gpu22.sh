#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_gpu22.txt
{
echo "== bench, the decoder's six encoder-attention K|V projections (and their data gradients) as one product each (one) against six (six), alternating"
for v in one six six one one six six one; do
  e=""; [ $v = six ] && e=1
  echo "-- $v $(S2T_AB_NO_XKV=$e python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("loss"))')"
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-400
