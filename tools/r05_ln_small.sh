#!/bin/bash
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lnp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lnp -- python3 $R/tools/ln_time.py > /tmp/ln_time.out 2>&1
grep "M=2560" /tmp/ln_time.out
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/lnp/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'ln_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# group consecutive launches with the same (kernel, grid)
out = []
for r in rows:
    key = (r['Kernel_Name'][:40], r['Grid_Size_X'], r['Workgroup_Size_X'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if out and out[-1][0] == key: out[-1][1].append(d)
    else: out.append((key, [d]))
for key, ds in out:
    ds = sorted(ds)
    print("%-42s grid %8s wg %5s  n=%3d  median %6.1f us" % (key[0], key[1], key[2], len(ds), ds[len(ds)//2]))
PY
