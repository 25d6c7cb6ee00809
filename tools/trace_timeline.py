#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into a picture of the pipeline: per hardware queue the kernels it carried and its
busy time, per kernel the launches and mean / max duration, and for a window in the steady state the order of starts.
   python3 tools/trace_timeline.py <dir with *_kernel_trace.csv> [t0_ms t1_ms]"""
import csv, glob, os, re, sys
from collections import defaultdict

d = sys.argv[1]
fn = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(fn)))


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n


ev = []
for r in rows:
    n = short(r["Kernel_Name"])
    if not n.startswith(("pm_", "pcs_", "pc_")):
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
t0 = ev[0][0]
print("kernels: %d, span %.1f ms" % (len(ev), (ev[-1][1] - t0) / 1e6))
perq = defaultdict(list)
perk = defaultdict(list)
for s, e, n, q, st in ev:
    perq[(q, st)].append((s, e, n))
    perk[n].append((e - s) / 1e6)
print("\nper kernel: launches, mean ms, max ms, total ms")
for n, v in sorted(perk.items(), key=lambda kv: -sum(kv[1])):
    print("  %-48s %6d %8.3f %8.3f %9.1f" % (n[:48], len(v), sum(v) / len(v), max(v), sum(v)))
print("\nper (queue, stream): kernels, busy ms (union), names")
for q, v in sorted(perq.items()):
    v.sort()
    busy = 0
    cs, ce = v[0][0], v[0][1]
    for s, e, n in v[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    names = defaultdict(int)
    for s, e, n in v:
        names[n.split("<")[0]] += 1
    print("  q=%s st=%s  n=%d busy=%.1f ms  %s" % (q[0], q[1], len(v), busy / 1e6, dict(names)))
if len(sys.argv) > 3:
    a, b = float(sys.argv[2]) * 1e6 + t0, float(sys.argv[3]) * 1e6 + t0
    print("\nwindow %.1f..%.1f ms: start, end, dur, queue, kernel" % (float(sys.argv[2]), float(sys.argv[3])))
    for s, e, n, q, st in ev:
        if s >= a and s <= b:
            print("  %9.3f %9.3f %7.3f  q%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, n.split("<")[0]))
