#!/bin/bash
# Regenerates the files of profiles/ on a GPU box (run from the repo root through gpurun); results land in gpurun_out/ref/.
# Order matters: the PMC passes come first so that bench.py's `roofline.traffic` reads the traffic file of THIS build.
set -x
R=$PWD
mkdir -p $R/gpurun_out/ref
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-dry-run > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pf/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra --no-dry-run > /dev/null 2>&1
python3 $R/tools/hbm_traffic.py /tmp/pf/fetch /tmp/pf/write $R/gpurun_out/ref/hbm_traffic.json
cp $R/gpurun_out/ref/hbm_traffic.json $R/profiles/r06_hbm_traffic.json
cd $R
python bench.py > $R/gpurun_out/ref/bench.json 2> $R/gpurun_out/ref/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $R/bench.py --no-cpu-baseline --no-extra --no-dry-run > $R/gpurun_out/ref/bench_under_rocprof.json 2>/tmp/pk.err
cp $(find /tmp/pk -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ref/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pa -- python3 $R/bench.py --no-cpu-baseline --no-roofline --no-extra --attn-2d > $R/gpurun_out/ref/bench_attn2d_under_rocprof.json 2>/tmp/pa.err
cp $(find /tmp/pa -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ref/kernel_stats_attn2d.csv
cd $R
python bench.py --no-cpu-baseline --no-roofline --no-extra --attn-2d > $R/gpurun_out/ref/bench_attn2d.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb8 -- python3 $R/bench.py --batch 8 --no-cpu-baseline --no-roofline --no-extra --no-dry-run > $R/gpurun_out/ref/bench_batch8_under_rocprof.json 2>/tmp/pb8.err
cp $(find /tmp/pb8 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ref/kernel_stats_batch8.csv
cd /tmp
for cfg in cfg5_beam5 cfg2_s_fp32 cfg5_kd_train; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$cfg -- python3 $R/bench.py --only $cfg --no-cpu-baseline > $R/gpurun_out/ref/${cfg}_under_rocprof.json 2>/tmp/p_$cfg.err
  cp $(find /tmp/p_$cfg -name "*kernel_stats.csv" | head -1) $R/gpurun_out/ref/kernel_stats_$cfg.csv
done
cd $R
python tools/decode_stamps.py bf16 200 100 > $R/gpurun_out/ref/decode_stamps.txt 2>&1
python tools/gemm_soak.py 2000 > $R/gpurun_out/ref/gemm_soak.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $R/gpurun_out/ref/smoke.txt 2>&1
tail -2 $R/gpurun_out/ref/smoke.txt
