"""Concurrency picture of the LAST train step in a rocprofv3 --kernel-trace CSV (two streams = two queues):
per queue the summed kernel time, the union of busy time, the time both queues have a kernel running, the idle time inside the
step, and -- for the window in which the side queue is active -- a per-kernel-family table of durations alone vs overlapped.
    python scripts/trace_overlap.py <..._kernel_trace.csv> [steps in the trace]"""
import csv
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void t2v::", "").replace("t2v::", "")
    return n.split("(")[0][:60]


def main(path, steps=4, window=None):
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), short(r["Kernel_Name"])) for r in rows))
    # the last step: a step ends with the optimisers' two adam_multi_kernel launches (the generator's: > 1 ms, then the
    # discriminators'); boundary = the last Adam launch that follows each long one
    big = [i for i, e in enumerate(ev) if "adam_multi" in e[3] and e[1] - e[0] > 1000000]
    ends = []
    for n, b in enumerate(big):
        stop = min(b + 200, big[n + 1] if n + 1 < len(big) else len(ev))
        later = [i for i, e in enumerate(ev[b:stop], b) if "adam_multi" in e[3]]
        ends.append(later[-1])
    if len(ends) < 2:
        print("fewer than two steps in the trace")
        return
    step = ev[ends[-2] + 1:ends[-1] + 1]
    t0, t1 = step[0][0], max(e[1] for e in step)
    queues = sorted({e[2] for e in step}, key=lambda q: -sum(e[1] - e[0] for e in step if e[2] == q))
    print("last step: %.2f ms wall, %d launches, queues %s" % ((t1 - t0) / 1e6, len(step), queues))
    for q in queues:
        es = [e for e in step if e[2] == q]
        print("  queue %s: %5d launches, %.2f ms of kernel time" % (q, len(es), sum(e[1] - e[0] for e in es) / 1e6))
    # sweep: time with 0 / 1 / >= 2 kernels running
    pts = []
    for s, e, q, n in step:
        pts.append((s, 1))
        pts.append((e, -1))
    pts.sort()
    depth, last, hist = 0, t0, defaultdict(int)
    for t, d in pts:
        hist[min(depth, 2)] += t - last
        last = t
        depth += d
    print("  idle %.2f ms, one kernel running %.2f ms, two or more %.2f ms" % (hist[0] / 1e6, hist[1] / 1e6, hist[2] / 1e6))
    # idle gaps: no kernel running; by size class, and the largest ones with the kernels either side
    busy_end, gaps = step[0][1], []
    for i, e in enumerate(step[1:], 1):
        if e[0] > busy_end:
            prev = max((p for p in step[:i]), key=lambda p: p[1])
            gaps.append((e[0] - busy_end, (busy_end - t0) / 1e6, prev[3], e[3], e[2]))
        busy_end = max(busy_end, e[1])
    for lo, hi in ((0, 2), (2, 5), (5, 10), (10, 30), (30, 1e9)):
        g = [x for x in gaps if lo * 1e3 <= x[0] < hi * 1e3]
        print("  gaps of %3g-%-5g us: %4d, %.2f ms" % (lo, hi, len(g), sum(x[0] for x in g) / 1e6))
    for d, at, a, b, q in sorted(gaps, reverse=True)[:25]:
        print("    %7.1f us at %6.2f ms: %-45s -> %-45s (queue %s)" % (d / 1e3, at, a[:45], b[:45], q))
    # phase marks (ms from the step's start): the loss terms close the forward passes, the first / last Winograd-domain weight
    # gradient bracket the second backward pass of the generator, the generator's Adam launch closes the backward passes
    def first(name, last=False):
        es = [e for e in step if name in e[3]]
        return None if not es else ((es[-1][1] if last else es[0][0]) - t0) / 1e6
    marks = [("loss_terms", first("loss_terms_kernel")), ("first wino_wgrad_sk", first("wino_wgrad_sk")),
             ("last wino_wgrad_sk ends", first("wino_wgrad_sk", True)), ("first adam_multi", first("adam_multi")), ("end", (t1 - t0) / 1e6)]
    print("  marks: " + ", ".join("%s %.2f" % (k, v) for k, v in marks if v is not None))
    if window:      # timeline dump: every launch that starts inside [a, b) ms of the step
        a, b = (float(v) for v in window.split("-"))
        for s_, e_, q, n in step:
            if a * 1e6 <= s_ - t0 < b * 1e6:
                print("    %s %9.1f -> %9.1f us (%7.1f)  %s" % ("M" if q == queues[0] else "      S", (s_ - t0) / 1e3, (e_ - t0) / 1e3,
                                                                (e_ - s_) / 1e3, n[:70]))
    if len(queues) < 2:
        return
    side = [e for e in step if e[2] != queues[0]]
    w0, w1 = min(e[0] for e in side), max(e[1] for e in side)
    print("  side queue active from %.2f to %.2f ms of the step (%.2f ms)" % ((w0 - t0) / 1e6, (w1 - t0) / 1e6, (w1 - w0) / 1e6))
    main_in = [e for e in step if e[2] == queues[0] and e[0] >= w0 and e[1] <= w1]
    print("  in that window: main queue %.2f ms of kernels, side queue %.2f ms" % (sum(e[1] - e[0] for e in main_in) / 1e6,
                                                                                  sum(e[1] - e[0] for e in side) / 1e6))
    # per family: average duration when no kernel of the other queue overlaps / when one does
    fam = defaultdict(lambda: [0, 0, 0, 0])
    other = {queues[0]: side, }
    for e in step:
        if e[0] < w0 or e[1] > w1:
            continue
        oth = side if e[2] == queues[0] else main_in
        ov = sum(max(0, min(e[1], o[1]) - max(e[0], o[0])) for o in oth)
        k = fam[(e[2] == queues[0], e[3])]
        if ov * 2 > (e[1] - e[0]):
            k[2] += 1
            k[3] += e[1] - e[0]
        else:
            k[0] += 1
            k[1] += e[1] - e[0]
    print("  %-5s %-60s %6s %9s %6s %9s" % ("queue", "kernel", "alone", "avg us", "ovlp", "avg us"))
    for (is_main, n), (a, ta, o, to) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][3]))[:24]:
        print("  %-5s %-60s %6d %9.1f %6d %9.1f" % ("main" if is_main else "side", n, a, ta / 1e3 / max(a, 1), o, to / 1e3 / max(o, 1)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4, sys.argv[3] if len(sys.argv) > 3 else None)
