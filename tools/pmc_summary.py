"""Per-kernel hardware-counter summary from the rocprofv3 --pmc passes written by tools/pmc.sh.
   python tools/pmc_summary.py gpurun_out/pmc_<tag>   -> CSV on stdout: kernel, dispatches, counter averages per dispatch.
FETCH_SIZE is doubled for gfx950 (the counter tallies 128-byte read requests as 64 bytes) and both sizes are
converted from KiB-units to bytes."""
import csv
import glob
import sys
from collections import defaultdict

prefix = sys.argv[1]
data = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for d in sorted(glob.glob(prefix + "_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            c = r["Counter_Name"]
            data[k][c] += float(r["Counter_Value"])
            calls[k][c] += 1
counters = sorted({c for k in data for c in data[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + [c + "_per_dispatch" for c in counters] + ["hbm_read_bytes_corrected", "hbm_write_bytes"])
for k in sorted(data, key=lambda k: -data[k].get("SQ_BUSY_CYCLES", 0)):
    n = max(calls[k].values())
    row = [k, n]
    for c in counters:
        row.append(round(data[k][c] / calls[k][c], 1) if calls[k][c] else "")
    fetch = data[k].get("FETCH_SIZE", 0) / max(calls[k].get("FETCH_SIZE", 1), 1)
    write = data[k].get("WRITE_SIZE", 0) / max(calls[k].get("WRITE_SIZE", 1), 1)
    row += [int(fetch * 1024 * 2), int(write * 1024)]
    w.writerow(row)
