"""Summarise rocprofv3 --pmc CSV output (counter_collection + kernel_trace) per kernel."""
import collections, csv, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
rows = list(csv.DictReader(open(d + "_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
seen = set()
for r in rows:
    if pat in r["Kernel_Name"]:
        k = r["Kernel_Name"][:90]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    ds = dur[k][len(dur[k]) // 2:]
    print(k, "| launches", len(dur[k]), "| avg us (2nd half) %.1f" % (sum(ds) / len(ds)))
    for c, vals in sorted(v.items()):
        vals = vals[len(vals) // 2:]
        print("   %-32s %.5g" % (c, sum(vals) / len(vals)))
