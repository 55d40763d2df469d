#!/bin/bash
# frame_prof.py under rocprofv3 --kernel-trace --stats; per-frame kernel table.  Usage: prof_frames.sh <tag> [frame_prof args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
tag=$1; shift
d=gpurun_out/prof_frames_$tag
rm -rf $d; mkdir -p $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python scripts/frame_prof.py "$@" > $d/log.txt 2>&1
grep FRAMES $d/log.txt
python - "$d" <<'PY'
import csv, glob, sys, re
d = sys.argv[1]
nf = float(re.search(r"FRAMES (\d+)", open(d + "/log.txt").read()).group(1))
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
out = []
for r in rows[:22]:
    out.append("%6.1f/frame %8.3f ms/frame %5.1f%% avg %7.1f us  %s" % (int(r["Calls"]) / nf, int(r["TotalDurationNs"]) / 1e6 / nf,
               100 * int(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3, r["Name"][:120]))
out.append("kernel time %.3f ms/frame, %.1f launches/frame" % (tot / 1e6 / nf, sum(int(r["Calls"]) for r in rows) / nf))
open(d + "/summary.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
