#!/usr/bin/env python3
"""PCIe-inclusive rate of the drop-in call: pemap_dev_map_batch with host buffers in and m1 / m2 / mapping_type out, 1 M pairs
per call, against the resident-reads rate bench.py reports.  Never `value`; recorded in DESIGN.md section 5."""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pecaller_amd import PemapDev   # noqa: E402

B = 1000000
dev = PemapDev(0)
gptr, cl = dev.synth_genome(20240601, 3100000000, 25, 0.5)
dev.build_index_resident(gptr, int(cl.sum()), cl)
dev.free(gptr)
dev.set_params(paired=True, min_dist=0, max_dist=500, min_align=0.85)
dev.synth_reads(12345, 3 * B, 150)
r1, l1, r2, l2 = dev.staged_reads()
for k in range(2):
    sl = slice(k * B, (k + 1) * B)
    dev.map_batch(r1[sl], l1[sl], r2[sl], l2[sl])          # warm-up: buffers allocated
t0 = time.time()
n = 0
for rep in range(2):
    for k in range(3):
        sl = slice(k * B, (k + 1) * B)
        dev.map_batch(r1[sl], l1[sl], r2[sl], l2[sl])
        n += B
dt = time.time() - t0
print({"pairs_per_call": B, "stride": int(r1.shape[1]), "host_bytes_in_per_pair": int(2 * r1.shape[1] + 8),
       "M_reads_per_s_host_buffers": round(2 * n / dt / 1e6, 2), "ms_per_call": round(dt / (n / B) * 1e3, 1)})
dev.close()
