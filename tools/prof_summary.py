"""Prints per-kernel averages from a rocprofv3 kernel_stats CSV:  python tools/prof_summary.py FILE [N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:n]:
    print("%-70s calls %5s  avg %10.1f us  %6s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
