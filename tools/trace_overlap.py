#!/usr/bin/env python3
"""How the device is used during a bench step, from a rocprofv3 kernel + memory-copy trace (run on the GPU box):
   tools/trace_overlap.py TRACE_DIR   -> concurrency of each kernel kind and of the copies over the busiest 60 % of the trace,
   per-queue busy fractions, and the gaps between consecutive operations of a queue."""
import csv, glob, sys, collections, re

d = sys.argv[1]
def short(n):
    m = re.search(r"(\w+_kernel)<([^>]*)>", n)
    if m:
        return m.group(1) + "<" + m.group(2).replace(" ", "") + ">"
    m = re.search(r"(\w+_kernel)\b", n)
    return m.group(1) if m else n[:40]
ops = []   # (start, end, kind, queue)
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
for fn in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy_" + r.get("Direction", "?"), "copy"))
ops.sort()
t0, t1 = ops[0][0], max(o[1] for o in ops)
starts = sorted(o[0] for o in ops if not o[2].startswith("copy"))
lo, hi = starts[int(0.25 * len(starts))], starts[int(0.9 * len(starts))]   # well inside the timed steps (the start is warm-up)
span = hi - lo
kinds = collections.defaultdict(lambda: [0, 0.0, 0.0])   # count, summed duration inside the window, summed full duration
for s, e, k, q in ops:
    if e <= lo or s >= hi:
        continue
    kinds[k][0] += 1
    kinds[k][1] += min(e, hi) - max(s, lo)
    kinds[k][2] += e - s
print("window %.1f ms of a %.1f ms trace" % (span / 1e6, (t1 - t0) / 1e6))
print("kind | ops in window | mean duration us | mean concurrency")
for k, (c, inside, full) in sorted(kinds.items(), key=lambda kv: -kv[1][1]):
    print("%s | %d | %.1f | %.2f" % (k, c, full / c / 1e3, inside / span))
# any-kernel busy fraction and the distribution of the number of kernels in flight
ev = []
for s, e, k, q in ops:
    if k.startswith("copy") or e <= lo or s >= hi:
        continue
    ev.append((max(s, lo), 1)); ev.append((min(e, hi), -1))
ev.sort()
cur, last, hist = 0, lo, collections.Counter()
for t, dlt in ev:
    hist[cur] += t - last
    last = t
    cur += dlt
hist[cur] += hi - last
print("kernels in flight: " + ", ".join("%d: %.1f%%" % (n, 100.0 * v / span) for n, v in sorted(hist.items())))
# per stream (queue): gap between the end of an operation and the start of the next one on the same queue
byq = collections.defaultdict(list)
for s, e, k, q in ops:
    if not k.startswith("copy") and s >= lo and e <= hi:
        byq[q].append((s, e, k))
gaps = collections.defaultdict(list)
for q, lst in byq.items():
    lst.sort()
    for (s0, e0, k0), (s1, e1, k1) in zip(lst, lst[1:]):
        gaps[k0 + " -> " + k1].append(s1 - e0)
print("gap between consecutive kernels of a queue (us): pair | n | median | mean")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:8]:
    v.sort()
    print("%s | %d | %.1f | %.1f" % (k, len(v), v[len(v) // 2] / 1e3, sum(v) / len(v) / 1e3))
print("queues with kernels: %d" % len(byq))
