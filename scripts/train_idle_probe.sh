#!/bin/bash
# How much of a train step is the GPU idle?  Kernel trace of train_bench.py (single stream: T2V_WGRAD_STREAM=0 so that intervals do
# not overlap), union of the kernel intervals against the wall time of the last step, and the gaps by size.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
d=gpurun_out/prof_idle; rm -rf $d; mkdir -p $d
T2V_WGRAD_STREAM=0 T2V_PACK_PREFETCH=0 rocprofv3 --kernel-trace --output-format csv -d $d -o t -- python scripts/train_bench.py --iters 3 > $d/log.txt 2>&1
grep "ms/step" $d/log.txt | head -2
python - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# last step = the interval between the last two adam_multi launches of the generator (largest ones)
adam = [i for i, (s, e, n) in enumerate(iv) if "adam_multi" in n]
marks = [iv[i][1] for i in adam]
# steps end with the D optimiser's adam: take the last two "ends" that are > 50 ms apart
ends = [marks[-1]]
for m in reversed(marks[:-1]):
    if ends[-1] - m > 50e6:
        ends.append(m)
    if len(ends) == 3:
        break
t1, t0 = ends[0], ends[1]
sel = [(s, e, n) for (s, e, n) in iv if s >= t0 and e <= t1]
busy, cur_s, cur_e, gaps, big, last_n = 0, None, None, [], [], ""
for s, e, n in sel:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        if s - cur_e > 60000:
            big.append(((s - cur_e) / 1e3, (cur_e - t0) / 1e6, last_n[:60], n[:60]))
        cur_s, cur_e = s, e
    last_n = n
busy += cur_e - cur_s
wall = t1 - t0
print("last step: wall %.2f ms, kernels %d, busy %.2f ms (%.1f %%), idle %.2f ms" % (wall / 1e6, len(sel), busy / 1e6, 100 * busy / wall, (wall - busy) / 1e6))
import collections
b = collections.Counter()
tot = collections.Counter()
for g in gaps:
    k = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"
    b[k] += 1; tot[k] += g
for k in ("<2us", "2-5us", "5-20us", "20-100us", ">100us"):
    print("   gaps %-9s %5d  %.2f ms" % (k, b[k], tot[k] / 1e6))
for g, at, a, c in big:
    print("   gap %7.1f us at %6.2f ms: after %s | before %s" % (g, at, a, c))
PY
rm -rf $d
