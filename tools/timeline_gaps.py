"""Where one step of a workload spends its time on the GPU, from a rocprofv3 kernel trace.

    python tools/timeline_gaps.py <kernel_trace.csv> [first_kernel_of_a_step] [--list]

Takes the last complete step in the trace (from one launch of `first_kernel_of_a_step` to the next; default: the
row-wise gate check, which opens a proof), and prints the kernels in launch order with the idle time in front of each
one, then the totals per kernel name and the largest gaps."""
import csv
import sys
from collections import defaultdict

listing = "--list" in sys.argv      # also print every launch of the step: gap before it, duration, name
argv = [a for a in sys.argv if a != "--list"]
path = argv[1]
first = argv[2] if len(argv) > 2 else "gate_check_kernel"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[2]]
if len(starts) < 2:
    raise SystemExit("fewer than two launches of %s in the trace" % first)
lo, hi = starts[-2], starts[-1]
step = rows[lo:hi]
span = rows[hi][0] - step[0][0]
busy = 0
per = defaultdict(lambda: [0, 0])
gaps = []
prev_end = step[0][0]
prev_short = "-"
for pos, (s, e, name) in enumerate(step):
    short = name.split("(")[0].split("<")[0].replace("zk::", "").replace("void ", "")
    gap = s - prev_end
    if gap > 0:
        gaps.append((gap, "%s  (launch %d, %.2f ms into the step, after %s)" % (short, pos, (s - step[0][0]) / 1e6, prev_short)))
    prev_short = short
    busy += e - max(s, prev_end) if e > prev_end else 0
    prev_end = max(prev_end, e)
    per[short][0] += e - s
    per[short][1] += 1
print("step: %.3f ms from first launch to the next step's first launch, %d kernels, GPU busy %.3f ms, idle %.3f ms"
      % (span / 1e6, len(step), busy / 1e6, (span - busy) / 1e6))
print("\nper kernel (sum of durations):")
for name, (ns, cnt) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("  %-40s %4d launches %9.3f ms" % (name[:40], cnt, ns / 1e6))
print("\nlargest idle gaps (before kernel):")
for gap, name in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f us  before %s" % (gap / 1e3, name))
if listing:
    print("\nlaunches in order (gap before, duration, kernel):")
    prev_end = step[0][0]
    for s, e, name in step:
        short = name.split("(")[0].split("<")[0].replace("zk::", "").replace("void ", "")
        print("  %7.1f us %8.1f us  %s" % (max(s - prev_end, 0) / 1e3, (e - s) / 1e3, short))
        prev_end = max(prev_end, e)
print("\nsum of gaps above 20 us: %.3f ms; of gaps below: %.3f ms"
      % (sum(g for g, _ in gaps if g > 20000) / 1e6, sum(g for g, _ in gaps if g <= 20000) / 1e6))
