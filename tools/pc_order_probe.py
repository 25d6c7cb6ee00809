#!/usr/bin/env python3
"""does the caller's rate depend on what the process did before the object was made?  Two objects one after the other: resident
run of 2 M columns x 3 and the dense seam rate of each.  With every stream at the default priority (PECALL_FLAT_PRIORITIES=1) the
first object of the process has the fast resident runs and a slow seam, the second the other way round (which streams share a
hardware queue); with the copies' streams at high and the early beam search's at low priority (the default) both have both.
python3 tools/pc_order_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pecaller_amd.pecall import PecallDev
n = 2000000
reads, dom = bench.pecall_columns(n, 64)
for small_first in (True, False):
    pc = PecallDev(0)
    if small_first:
        pc.call_sites(reads[:20000], dom[:20000])
    out = pc.out_arrays(n, 64)
    import time
    for arr in (reads, dom) + out:
        pc.pin_host(arr)
    pc.call_sites(reads, dom, out=out)
    t0 = time.perf_counter(); pc.call_sites(reads, dom, out=out); seam = time.perf_counter() - t0
    for arr in (reads, dom) + out:
        pc.unpin_host(arr)
    pc.sites_stage(reads, dom)
    print("priorities %-8s object %s: resident ms %s seam M columns/s %.1f" % (("flat" if os.environ.get("PECALL_FLAT_PRIORITIES") else "default"), "1st of the process (small call first)" if small_first else "2nd (large call first)", [round(pc.sites_run(), 2) for _ in range(3)], n / seam / 1e6), flush=True)
    pc.close()
