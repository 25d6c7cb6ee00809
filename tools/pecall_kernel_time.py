#!/usr/bin/env python3
"""the per-site caller's kernel alone on resident columns (bench.py's PECaller leg without the rest of the bench)
   python3 tools/pecall_kernel_time.py [sites]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pecaller_amd.pecall import PecallDev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
reads, dom = bench.pecall_columns(n, 64)
pc = PecallDev(0)
pc.call_sites(reads[:20000], dom[:20000])
pc.sites_stage(reads, dom)
ms = [pc.sites_run() for _ in range(5)]
npass = pc.sites_collect()[4]
print("kernel ms", [round(x, 2) for x in ms], "M columns/s", round(n / (min(ms) * 1e-3) / 1e6, 2), "passes", np.bincount(npass).tolist())
