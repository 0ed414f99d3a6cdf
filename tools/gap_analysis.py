"""Idle time between kernels from a rocprofv3 --kernel-trace CSV of bench.py: python tools/gap_analysis.py <kernel_trace.csv> [steps_to_skip]
Prints, for the traced interval after the skipped head, wall time, the union of kernel busy time, and which (previous kernel -> next kernel)
boundaries the idle time sits at."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0][:48]
# steps are delimited by the optimizer kernel
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lo, hi = adam[skip] + 1, adam[-1] + 1
seg = rows[lo:hi]; nsteps = len(adam) - 1 - skip
t0 = int(seg[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in seg)
busy = 0; cur_end = t0; gaps = collections.Counter(); ngaps = collections.Counter(); prev = None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        if prev is not None:
            gaps[(name(prev), name(r))] += s - cur_end; ngaps[(name(prev), name(r))] += 1
        busy += e - s; cur_end = e; prev = r
    elif e > cur_end:
        busy += e - cur_end; cur_end = e; prev = r
print("steps %d: wall %.3f ms/step, busy (union) %.3f ms/step, idle %.3f ms/step, kernels/step %.0f" %
      (nsteps, (t1 - t0) / 1e6 / nsteps, busy / 1e6 / nsteps, (t1 - t0 - busy) / 1e6 / nsteps, len(seg) / nsteps))
tot = sum(gaps.values())
for k, v in gaps.most_common(25):
    print("  %7.1f us/step  (%5.1f per step x %5.2f us)  %s -> %s" % (v / 1e3 / nsteps, ngaps[k] / nsteps, v / 1e3 / ngaps[k], k[0], k[1]))
hist = collections.Counter()
for k, v in gaps.items():
    hist[min(int(v / ngaps[k] / 1e3), 20)] += v
print("idle by average gap length (us -> us/step):", {k: round(v / 1e3 / nsteps, 1) for k, v in sorted(hist.items())})
