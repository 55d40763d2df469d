"""Split one kernel's launches in a rocprofv3 kernel trace into back-to-back runs of itself (bench.py's isolated timing loop)
and launches inside frames (neighbours are other kernels; two streams overlap there).  usage: split_kernel_trace.py <trace.csv> <name substring>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
key = sys.argv[2]
iso, fr = [], []
for i, r in enumerate(rows):
    if key not in r["Kernel_Name"]:
        continue
    prev = rows[i - 1]["Kernel_Name"] if i > 0 else ""
    nxt = rows[i + 1]["Kernel_Name"] if i + 1 < len(rows) else ""
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    (iso if (key in prev and key in nxt) else fr).append(d)
# overlap: share of a launch's interval during which another launch of ANY kernel was running
print("kernel: %s" % key)
for name, v in (("back-to-back (isolated loop)", iso), ("inside frames (two streams)", fr)):
    if v:
        v.sort()
        print("  %-30s %6d launches  avg %7.1f us  median %7.1f us  min %7.1f us" % (name, len(v), sum(v) / len(v), v[len(v) // 2], v[0]))
