#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r05_gpu21.txt
{
python -m pytest tests -m gpu -x -q -k "ctc or criterion or multi_loss or trainer" 2>&1 | tail -3
echo "== bench, CTC gradient kernel on the side stream under the decoder's backward (side) against the main stream (main), alternating"
for v in side main main side side main main side; do
  e=""; [ $v = main ] && e=1
  echo "-- $v $(S2T_AB_CTC_MAIN=$e python bench.py --no-cpu-baseline --no-extra --no-roofline --steps 40 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("loss"))')"
done
} > $O 2>&1
grep -v amdgpu.ids $O | cut -c1-300
