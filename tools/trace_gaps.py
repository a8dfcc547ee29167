#!/usr/bin/env python3
"""GPU idle time inside the timed steps of a rocprofv3 --kernel-trace run of bench.py: union of the kernel intervals against the wall
time of the last `steps` steps (a step = from one adam_kernel to the next).   usage: trace_gaps.py <kernel_trace.csv> [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_kernel" in e[2]]
if len(adam) < steps + 1:
    raise SystemExit("only %d optimizer launches in the trace" % len(adam))
lo, hi = adam[-steps - 1], adam[-1]
seg = ev[lo + 1:hi + 1]
t0, t1 = ev[lo][1], ev[hi][1]
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n in seg:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
wall = t1 - t0
print("%d steps: wall %.3f ms per step, some kernel running %.3f ms per step, idle %.1f us per step in %d gaps per step" %
      (steps, wall / steps / 1e6, busy / steps / 1e6, (wall - busy) / steps / 1e3, len(gaps) / steps))
total = sum(e - s for s, e, n in seg)
print("sum of kernel durations %.3f ms per step (overlap of the two streams: %.1f us per step)" % (total / steps / 1e6, (total - busy) / steps / 1e3))
agg = {}
for g, n in gaps:
    k = n.split("(")[0][-60:]
    a = agg.setdefault(k, [0, 0])
    a[0] += g
    a[1] += 1
print("idle time in front of (us per step, count per step):")
for k, (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %8.1f %5.1f  %s" % (g / steps / 1e3, c / steps, k))
