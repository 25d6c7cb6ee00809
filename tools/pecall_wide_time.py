#!/usr/bin/env python3
"""the per-site caller on resident columns of more than 64 samples: python3 tools/pecall_wide_time.py [samples] [sites]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pecaller_amd.pecall import PecallDev
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
reads, dom = bench.pecall_columns(n, S)
pc = PecallDev(0)
pc.call_sites(reads[:5000], dom[:5000])
pc.sites_stage(reads, dom)
ms = [pc.sites_run() for _ in range(3)]
npass = pc.sites_collect()[4]
print("samples", S, "columns", n, "kernel ms", [round(x, 2) for x in ms], "M columns/s", round(n / (min(ms) * 1e-3) / 1e6, 3), "passes", np.bincount(npass).tolist())
