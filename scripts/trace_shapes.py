"""Per-launch-shape table from a rocprofv3 --kernel-trace CSV: groups launches by (kernel name, grid, workgroup, LDS) -- i.e. by
layer shape -- and prints calls / total / average per step, plus the time between the first and last launch and the idle gaps.
    python scripts/trace_shapes.py <..._kernel_trace.csv> [steps] [name filter]"""
import csv
import sys
from collections import defaultdict


def main(path, steps=1, flt=None):
    rows = list(csv.DictReader(open(path)))
    g = defaultdict(lambda: [0, 0])
    for r in rows:
        name = r["Kernel_Name"]
        if flt and flt not in name:
            continue
        key = (name[:90], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""),
               r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        g[key][0] += 1
        g[key][1] += d
    tot = sum(v[1] for v in g.values())
    print("%7s %10s %9s  grid(x,y,z)/wg/lds  kernel" % ("calls/st", "ms/step", "avg us"))
    for key, (n, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:70]:
        print("%7.1f %10.3f %9.1f  %s,%s,%s/%s/%s  %s" % (n / steps, t / 1e6 / steps, t / 1e3 / n, key[1], key[2], key[3], key[4], key[5], key[0]))
    print("kernel time per step %.2f ms over %d steps" % (tot / 1e6 / steps, steps))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1, sys.argv[3] if len(sys.argv) > 3 else None)
