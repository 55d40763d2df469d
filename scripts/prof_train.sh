#!/bin/bash
# train_bench.py under rocprofv3 --kernel-trace --stats; prints the per-kernel table per step
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o train -- python scripts/train_bench.py --iters 2 "$@" > gpurun_out/prof_train/log.txt 2>&1
tail -1 gpurun_out/prof_train/log.txt
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print("%6s %9.2f ms %5.1f%%  avg %8.1f us  %s" % (r["Calls"], int(r["TotalDurationNs"]) / 1e6 / 4, 100 * int(r["TotalDurationNs"]) / tot,
                                                   float(r["AverageNs"]) / 1e3, r["Name"][:100]))
print("kernel time per step (4 steps incl. warm-up) %.1f ms" % (tot / 1e6 / 4))
PY
