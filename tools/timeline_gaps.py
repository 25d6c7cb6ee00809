#!/usr/bin/env python3
"""Where the device waits: from a rocprofv3 --kernel-trace CSV, the idle gaps of every hardware queue and of the device as a whole inside
the timed region of a bench.py run (the seam steps come first, the resident steps after them).
   python3 tools/timeline_gaps.py <kernel_trace.csv> [min_gap_us=150]"""
import csv, sys, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"^void ", "", r["Kernel_Name"]); n = re.sub(r"[<(].*$", "", n)
    if n.startswith("pm_"):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
rows.sort()
min_gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 150e3
# phases: separated by device-wide gaps of more than 20 ms (index build / synth / sync between the seam and the resident run)
phases, cur, end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - end > 20e6:
        phases.append(cur); cur = []
    cur.append(r); end = max(end, r[1])
phases.append(cur)
for pi, ph in enumerate(phases):
    t0, t1 = ph[0][0], max(r[1] for r in ph)
    n_seed = sum(1 for r in ph if r[2].startswith("pm_seed4"))
    if n_seed < 8:
        continue
    print("== phase %d: %.1f ms, %d kernels, %d pm_seed4 launches" % (pi, (t1 - t0) / 1e6, len(ph), n_seed))
    # device-wide idle
    idle, e = 0, ph[0][1]
    for r in ph[1:]:
        if r[0] > e:
            idle += r[0] - e
        e = max(e, r[1])
    print("   device idle (no pm_ kernel running): %.2f ms" % (idle / 1e6))
    # the seed kernel's own stream: gaps between consecutive pm_seed4 launches
    seeds = [r for r in ph if r[2].startswith("pm_seed4")]
    busy = sum(r[1] - r[0] for r in seeds)
    gaps = [(b[0] - a[1], a[1] - t0) for a, b in zip(seeds, seeds[1:])]
    print("   pm_seed4: busy %.2f ms (avg %.3f), sum of gaps between launches %.2f ms, largest gaps (ms at ms): %s" % (
        busy / 1e6, busy / len(seeds) / 1e6, sum(g for g, _ in gaps) / 1e6,
        ", ".join("%.2f@%.1f" % (g / 1e6, at / 1e6) for g, at in sorted(gaps, reverse=True)[:8])))
    byq = {}
    for r in ph:
        byq.setdefault(r[3], []).append(r)
    for q, rs in sorted(byq.items()):
        b = sum(r[1] - r[0] for r in rs)
        names = {}
        for r in rs:
            names[r[2]] = names.get(r[2], 0) + (r[1] - r[0])
        top = sorted(names.items(), key=lambda kv: -kv[1])[:4]
        print("   queue %s: %d kernels, busy %.2f ms: %s" % (q, len(rs), b / 1e6, ", ".join("%s %.1f" % (k, v / 1e6) for k, v in top)))
