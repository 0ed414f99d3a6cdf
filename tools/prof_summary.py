"""Per-step summary of a rocprofv3 --kernel-trace --stats kernel_stats.csv: python tools/prof_summary.py <csv> <steps>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(int(r['TotalDurationNs']) for r in rows)
print("kernel time per step: %.3f ms" % (tot / steps / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-100s %6.1f calls/step %8.3f ms/step %9.1f us avg" % (r['Name'][:100], int(r['Calls']) / steps,
                                                                   int(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3))
